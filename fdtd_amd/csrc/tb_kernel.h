// tb_kernel.h -- two fused time steps of a radius-1 sweep (fdtd1.cu / jacobi.cu) in one launch.
//
// Counterpart of jacobi.cu's temporal_blocking kernel (jacobi.cu:115-400: 3 fused steps, 32x32
// blocks, one LDS plane per time level).  Fusing steps is what lowers HBM traffic below one
// read + one write per cell-update; this kernel moves 8 bytes per cell per TWO updates.
//
// Organisation (different from the single-step kernels on purpose: everything that is reused
// stays in registers, LDS only carries what crosses a wave):
//   * one WAVE owns RY consecutive rows over 64*RX quads (the whole x extent up to X = 256*RX;
//     wider rows are cut into x tiles overlapping by one quad): lane l holds quads l, l+64, ...
//     (RX per row), so a wave-load is 1 KiB contiguous and every x neighbour lives in the same
//     wave: it is fetched with a DPP wave rotate, no LDS.
//   * y neighbours between the rows of one wave are registers of the same lane; only the first
//     and last row of each wave are published to LDS for the wave above/below (1 row each way,
//     per time level).
//   * z neighbours are register queues of three planes per time level; the z loop is unrolled
//     three times so that the queues rotate by renaming.
//   * a workgroup of NW waves covers NW*RY rows; tiles overlap by 2 rows on each side (the rows
//     whose level-1 / level-2 values would need data outside the tile are recomputed by the
//     neighbouring tile), so NW*RY-4 rows per tile are stored.
//   * iteration p consumes input plane p, produces level-1 plane p-1 and stores level-2 plane p-2.
//
// Fixed shell.  The reference never writes the outermost cell layer of either cube, and its two
// cubes start with DIFFERENT shells (fdtd1.cu:28-29).  After one step the shell a stencil sees
// is that of the other cube, so the fused kernel takes level-1 shell cells from `mid` (the
// cube the skipped intermediate step would have written) and leaves the shell of `out`
// untouched; `out` must already carry the shell of `in` (the caller copies it once).
// With that the fused launch is bit-identical to two single steps.
#pragma once
#include "sweep_common.h"

namespace fdtdx {

struct TbParams {
    const float *in;   // level 0: full cube
    const float *mid;  // shell source for level 1 (interior values are ignored)
    float *out;        // level 2: interior written, shell untouched
    long long plane;
    int px, pxq, nx, ny;
    int nz;            // planes in the local cube (ghost planes included)
    int zg0, nzg;      // global z of local plane 0, planes of the global grid: global planes 0 and nzg-1 are shell
    int z_lo, z_hi;    // local output planes [z_lo, z_hi); planes z_lo-2 .. z_hi+1 must exist or be global shell
    int zchunk;
    int gx, gy, gz;    // x tiles, y tiles, z chunks (1-D launch of gx*gy*gz workgroups)
    int qstep, qbase;  // x tile bx loads quads [qbase + bx*qstep, +64*RX); one tile: qbase 0, else -1 (see kernel)
    float c[7];
};

__device__ __forceinline__ float dpp_from_prev_lane(float v)  // lane l receives lane l-1 (lane 0: lane 63)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x13C, 0xF, 0xF, false));
}
__device__ __forceinline__ float dpp_from_next_lane(float v)  // lane l receives lane l+1 (lane 63: lane 0)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x134, 0xF, 0xF, false));
}

// one cell: the reference's summation order with every "+ c*v" fused (device order)
template <int ORDER>
__device__ __forceinline__ float cell7(const float (&c)[7], float C, float xp, float xm, float yp, float ym, float zp, float zm)
{
    if constexpr (ORDER == ORD_STAR_CHAIN) {  // fdtd1.cu:63
        float a = c[0] * C;
        a = __builtin_fmaf(c[1], xp, a);
        a = __builtin_fmaf(c[2], xm, a);
        a = __builtin_fmaf(c[3], yp, a);
        a = __builtin_fmaf(c[4], ym, a);
        a = __builtin_fmaf(c[5], zp, a);
        return __builtin_fmaf(c[6], zm, a);
    } else {  // jacobi.cu:31-38: ce*E + cw*W + cs*S + cn*N + ct*T + cb*B + cc*C
        float a = c[1] * xp;
        a = __builtin_fmaf(c[2], xm, a);
        a = __builtin_fmaf(c[3], yp, a);
        a = __builtin_fmaf(c[4], ym, a);
        a = __builtin_fmaf(c[5], zp, a);
        a = __builtin_fmaf(c[6], zm, a);
        return __builtin_fmaf(c[0], C, a);
    }
}

// One time level of one plane for all rows of the wave: out[j][i] = stencil(centre plane `cen`,
// z neighbours `zp`/`zm`), y neighbours across waves from `up`/`dn`.
template <int RY, int RX, int ORDER>
__device__ __forceinline__ void level_step(float4 (&res)[RY][RX], const float4 (&zm)[RY][RX], const float4 (&cen)[RY][RX],
                                           const float4 (&zp)[RY][RX], const float4 (&up)[RX], const float4 (&dn)[RX],
                                           const float (&c)[7], int lane)
{
#pragma unroll
    for (int j = 0; j < RY; j++) {
        float lw[RX], rx[RX];
#pragma unroll
        for (int i = 0; i < RX; i++) {
            lw[i] = dpp_from_prev_lane(cen[j][i].w);
            rx[i] = dpp_from_next_lane(cen[j][i].x);
        }
#pragma unroll
        for (int i = 0; i < RX; i++) {
            // lane 0's left neighbour is lane 63 of the previous quad column, lane 63's right
            // neighbour is lane 0 of the next one (the outermost two are shell cells: unused)
            const float L = (i > 0 && lane == 0) ? lw[i - 1] : lw[i];
            const float R = (i < RX - 1 && lane == 63) ? rx[i + 1] : rx[i];
            const float4 C = cen[j][i];
            const float4 Yp = (j < RY - 1) ? cen[j + 1][i] : dn[i];
            const float4 Ym = (j > 0) ? cen[j - 1][i] : up[i];
            const float4 Zp = zp[j][i], Zm = zm[j][i];
            res[j][i].x = cell7<ORDER>(c, C.x, C.y, L, Yp.x, Ym.x, Zp.x, Zm.x);
            res[j][i].y = cell7<ORDER>(c, C.y, C.z, C.x, Yp.y, Ym.y, Zp.y, Zm.y);
            res[j][i].z = cell7<ORDER>(c, C.z, C.w, C.y, Yp.z, Ym.z, Zp.z, Zm.z);
            res[j][i].w = cell7<ORDER>(c, C.w, R, C.z, Yp.w, Ym.w, Zp.w, Zm.w);
        }
    }
}

template <int NW, int RY, int RX, int ORDER>
__global__ __launch_bounds__(NW * 64) void tb2_r1_kernel(const TbParams p)
{
    constexpr int KT = 2;
    constexpr int ROWS = NW * RY;       // rows a tile loads
    constexpr int CYO = ROWS - 2 * KT;  // rows a tile stores
    // boundary rows exchanged through LDS: [parity][level][wave][side][quad column][lane]
    __shared__ float4 ex[2][KT][NW][2][RX][64];

    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    int bx, by, bz;
    {
        SweepParams g;  // only the grid fields are read by logical_tile
        g.gx = p.gx, g.gy = p.gy, g.gz = p.gz;
        logical_tile(g, bx, by, bz);
    }
    // Rows wider than one wave's 64*RX quads are cut into x tiles that overlap by one quad on each
    // side: the outermost quad of a tile only feeds its neighbours' level-1 values (a level-2
    // cell reaches two cells, i.e. never beyond the adjacent quad) and is stored by the next tile.
    const int tq0 = p.qbase + bx * p.qstep;
    const bool xtiled = p.gx > 1;
    const int zs = p.z_lo + bz * p.zchunk;
    const int ze = min(zs + p.zchunk, p.z_hi);
    if (zs >= ze) return;
    const int yt0 = 1 - KT + by * CYO;  // first row the tile loads (may be negative)
    const int yown0 = yt0 + KT, yown1 = yown0 + CYO;

    // ---- per-lane constants ---------------------------------------------------------------
    unsigned goff[RY][RX];   // quad inside the plane (rows/quads outside are out of range)
    unsigned moff[RY][RX];   // ... only where the quad contains x/y shell cells (level-1 shell source)
    unsigned soff[RY][RX];   // ... only where the quad is stored
    unsigned imask[RY][RX];  // bit k: component k is an interior cell in x and y
#pragma unroll
    for (int j = 0; j < RY; j++) {
        const int y = yt0 + w * RY + j;
        const bool ydom = (y >= 0) && (y < p.ny);
        const bool yin = (y >= 1) && (y < p.ny - 1);
        const bool yown = (y >= yown0) && (y < yown1);
#pragma unroll
        for (int i = 0; i < RX; i++) {
            const int lq = lane + 64 * i;
            const int qx = tq0 + lq;
            const int x0 = 4 * qx;
            const bool xown = !xtiled || (lq >= 1 && lq < 64 * RX - 1);
            unsigned m = 0;
#pragma unroll
            for (int k = 0; k < 4; k++)
                if (yin && x0 + k >= 1 && x0 + k < p.nx - 1) m |= 1u << k;
            const bool qdom = ydom && (qx >= 0) && (qx < p.pxq);
            const unsigned off = (unsigned)(y * p.px + x0) * 4u;
            imask[j][i] = m;
            goff[j][i] = qdom ? off : OOB;
            moff[j][i] = (qdom && x0 < p.nx && m != 0xFu) ? off : OOB;
            soff[j][i] = (qdom && yown && xown && m != 0u) ? off : OOB;
        }
    }
    float c[7];
#pragma unroll
    for (int k = 0; k < 7; k++) c[k] = p.c[k];

    const unsigned plane_bytes = (unsigned)(p.plane * 4);
    auto in_rsrc = [&](int z) { return make_rsrc(p.in + (long long)z * p.plane, (z >= 0 && z < p.nz) ? plane_bytes : 0u); };
    auto mid_rsrc = [&](int z) { return make_rsrc(p.mid + (long long)z * p.plane, (z >= 0 && z < p.nz) ? plane_bytes : 0u); };

    // ---- queues: q0 = input planes, q1 = level-1 planes; slot (k + phase) % 3 holds the
    //      k-th oldest plane -------------------------------------------------------------------
    float4 q0[3][RY][RX], q1[3][RY][RX], mv[RY][RX];
    {
        // first iteration is p = zs: q0 = planes zs-2, zs-1, zs at phase 0
#pragma unroll
        for (int k = 0; k < 3; k++) {
            const __amdgpu_buffer_rsrc_t r = in_rsrc(zs - 2 + k);
#pragma unroll
            for (int j = 0; j < RY; j++)
#pragma unroll
                for (int i = 0; i < RX; i++) {
                    q0[k][j][i] = buf_ld4(r, goff[j][i]);
                    q1[k][j][i] = make_float4(0.f, 0.f, 0.f, 0.f);
                }
        }
        // level-1 shell values for plane zs-1
        const bool zsh = (p.zg0 + zs - 1 <= 0) || (p.zg0 + zs - 1 >= p.nzg - 1);
        const __amdgpu_buffer_rsrc_t r = mid_rsrc(zs - 1);
#pragma unroll
        for (int j = 0; j < RY; j++)
#pragma unroll
            for (int i = 0; i < RX; i++) mv[j][i] = buf_ld4(r, zsh ? goff[j][i] : moff[j][i]);
    }

    const int wu = (w > 0) ? w - 1 : 0, wd = (w < NW - 1) ? w + 1 : NW - 1;
    int par = 0;
    int pz = zs;  // input plane consumed by the current iteration
    const int pend = ze + 2;

    auto iteration = [&](auto ph) {
        constexpr int PH = decltype(ph)::value;
        constexpr int OLD = PH % 3, MID = (PH + 1) % 3, NEW = (PH + 2) % 3;
        // publish this wave's first/last rows of the two centre planes (input plane pz-1, level-1 plane pz-2)
#pragma unroll
        for (int i = 0; i < RX; i++) {
            ex[par][0][w][0][i][lane] = q0[MID][0][i];
            ex[par][0][w][1][i][lane] = q0[MID][RY - 1][i];
            ex[par][1][w][0][i][lane] = q1[NEW][0][i];  // q1 slots: MID = plane pz-3, NEW = pz-2, OLD = (dead) pz-4
            ex[par][1][w][1][i][lane] = q1[NEW][RY - 1][i];
        }
        __syncthreads();
        float4 up0[RX], dn0[RX], up1[RX], dn1[RX];
#pragma unroll
        for (int i = 0; i < RX; i++) {
            up0[i] = ex[par][0][wu][1][i][lane];
            dn0[i] = ex[par][0][wd][0][i][lane];
            up1[i] = ex[par][1][wu][1][i][lane];
            dn1[i] = ex[par][1][wd][0][i][lane];
        }
        // ---- level 1 at plane pz-1 from input planes pz-2, pz-1, pz --------------------------
        {
            float4 res[RY][RX];
            level_step<RY, RX, ORDER>(res, q0[OLD], q0[MID], q0[NEW], up0, dn0, c, lane);
            const bool zsh = (p.zg0 + pz - 1 <= 0) || (p.zg0 + pz - 1 >= p.nzg - 1);  // whole plane is shell
            // q1 rotates one step behind q0's naming: its OLD slot (plane pz-4) is dead, level-1
            // plane pz-1 goes there.  With q1 planes (pz-3, pz-2, pz-1) = (MID, NEW, OLD) slots.
#pragma unroll
            for (int j = 0; j < RY; j++)
#pragma unroll
                for (int i = 0; i < RX; i++) {
                    const unsigned m = zsh ? 0u : imask[j][i];
                    q1[OLD][j][i].x = (m & 1u) ? res[j][i].x : mv[j][i].x;
                    q1[OLD][j][i].y = (m & 2u) ? res[j][i].y : mv[j][i].y;
                    q1[OLD][j][i].z = (m & 4u) ? res[j][i].z : mv[j][i].z;
                    q1[OLD][j][i].w = (m & 8u) ? res[j][i].w : mv[j][i].w;
                }
        }
        // ---- level 2 at plane pz-2 from level-1 planes pz-3 (MID), pz-2 (NEW), pz-1 (OLD) ------
        {
            float4 res[RY][RX];
            level_step<RY, RX, ORDER>(res, q1[MID], q1[NEW], q1[OLD], up1, dn1, c, lane);
            const bool st = (pz - 2 >= zs) && (pz - 2 < ze);
            const __amdgpu_buffer_rsrc_t ro = make_rsrc(p.out + (long long)(pz - 2) * p.plane, st ? plane_bytes : 0u);
            // shell components of a stored quad keep the value `out` holds, which equals the input's
            // (plane pz-2 is the oldest slot of q0)
#pragma unroll
            for (int j = 0; j < RY; j++)
#pragma unroll
                for (int i = 0; i < RX; i++) {
                    const unsigned m = imask[j][i];
                    buf_st4(ro, soff[j][i], (m & 1u) ? res[j][i].x : q0[OLD][j][i].x, (m & 2u) ? res[j][i].y : q0[OLD][j][i].y,
                            (m & 4u) ? res[j][i].z : q0[OLD][j][i].z, (m & 8u) ? res[j][i].w : q0[OLD][j][i].w);
                }
        }
        // ---- loads for the next iteration: input plane pz+1 into the dead slot, shell source of plane pz
        {
            const __amdgpu_buffer_rsrc_t rn = in_rsrc((pz + 1 < pend) ? pz + 1 : -1);
            const bool zshn = (p.zg0 + pz <= 0) || (p.zg0 + pz >= p.nzg - 1);
            const __amdgpu_buffer_rsrc_t rm = mid_rsrc((pz + 1 < pend) ? pz : -1);
#pragma unroll
            for (int j = 0; j < RY; j++)
#pragma unroll
                for (int i = 0; i < RX; i++) {
                    q0[OLD][j][i] = buf_ld4(rn, goff[j][i]);
                    mv[j][i] = buf_ld4(rm, zshn ? goff[j][i] : moff[j][i]);
                }
        }
        par ^= 1;
        pz++;
    };

    while (pz < pend) {
        unroll_phases<0, 3>([&](auto ph) {
            if (pz < pend) iteration(ph);
        });
    }
}

struct TbVariantInfo {
    int order;  // ORD_STAR_CHAIN | ORD_JACOBI_CHAIN
    int nw, ry, rx, threads, lds_bytes;
    void (*launch)(const TbParams &, dim3 grid, hipStream_t);
};
int num_tb_variants();
const TbVariantInfo *tb_variant(int i);
void register_tb_variant(const TbVariantInfo &v);

template <int NW, int RY, int RX, int ORDER>
void tb2_launch(const TbParams &p, dim3 grid, hipStream_t s)
{
    hipLaunchKernelGGL((tb2_r1_kernel<NW, RY, RX, ORDER>), grid, dim3(NW * 64), 0, s, p);
}

template <int NW, int RY, int RX, int ORDER>
struct TbRegistrar {
    TbRegistrar()
    {
        TbVariantInfo v;
        v.order = ORDER;
        v.nw = NW;
        v.ry = RY;
        v.rx = RX;
        v.threads = NW * 64;
        v.lds_bytes = 2 * 2 * NW * 2 * RX * 64 * 16;
        v.launch = &tb2_launch<NW, RY, RX, ORDER>;
        register_tb_variant(v);
    }
};
#define FDTDX_TB_VARIANT(NW, RY, RX, ORDER) static ::fdtdx::TbRegistrar<NW, RY, RX, ORDER> tbreg_##NW##_##RY##_##RX##_##ORDER;

}  // namespace fdtdx
