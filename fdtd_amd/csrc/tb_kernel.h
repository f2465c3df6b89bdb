// tb_kernel.h -- two or three fused time steps of a radius-1 sweep (fdtd1.cu / jacobi.cu) in one launch.
//
// Counterpart of jacobi.cu's temporal_blocking kernel (jacobi.cu:115-400: 3 fused steps, 32x32
// blocks, one LDS plane per time level).  Fusing steps is what lowers HBM traffic below one
// read + one write per cell-update; this kernel moves 8 bytes per cell per KT = 2 or 3 updates
// (plus the rows neighbouring tiles both load).
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
//   * a workgroup of NW waves covers NW*RY rows; tiles overlap by KT rows on each side (the rows
//     whose higher-level values would need data outside the tile are recomputed by the
//     neighbouring tile), so NW*RY-2*KT rows per tile are stored.
//   * the iteration that consumes input plane p produces level-l plane p-l for l = 1..KT-1 and
//     stores level-KT plane p-KT.
//
// Fixed shell.  The reference never writes the outermost cell layer of either cube, and its two
// cubes start with DIFFERENT shells (fdtd1.cu:28-29).  After one step the shell a stencil sees
// is that of the other cube, so levels of equal parity see equal shells: the kernel takes the
// shell cells of odd levels from `mid` (the cube a single step from `in` would write) and those
// of even levels from `in`, and leaves the shell of `out` untouched.  `out` must already carry the
// shell of its level: for KT = 2 that of `in` (the caller copies it once into a third cube), for
// KT = 3 that of `mid`, so a three-step launch can run straight from one of the reference's two
// cubes into the other (mid == out).  With that a fused launch is bit-identical to KT single steps.
#pragma once
#include "sweep_common.h"

namespace fdtdx {

struct TbParams {
    const float *in;   // level 0: full cube
    const float *mid;  // shell source for level 1 (interior values are ignored)
    const float *even; // shell source for level 0 where it is not `in` itself (kernels with TbVariantInfo::shells only)
    float *out;        // level 2: interior written, shell untouched
    long long plane;
    int px, pxq, nx, ny;
    int nz;            // planes in the local cube (ghost planes included)
    int zg0, nzg;      // global z of local plane 0, planes of the global grid: global planes 0 and nzg-1 are shell
    int z_lo, z_hi;    // local output planes [z_lo, z_hi); planes z_lo-2 .. z_hi+1 must exist or be global shell
    int zchunk;
    int gx, gy, gz;    // x tiles, y tiles, z chunks (1-D launch of gx*gy*gz workgroups)
    int gz1, z_lo2, z_hi2;  // z chunks gz1 .. gz-1 belong to a second plane range [z_lo2, z_hi2) (gz1 == gz: there is none)
    int qstep, qbase;  // x tile bx loads quads [qbase + bx*qstep, +64*RX); one tile: qbase 0, else -1 (see kernel)
    int xt;            // rows are cut into x tiles (possibly launched separately): a tile does not store its two outermost quads
    float c[13];
};

__device__ __forceinline__ float dpp_from_prev_lane(float v)  // lane l receives lane l-1 (lane 0: lane 63)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x13C, 0xF, 0xF, true));
}
__device__ __forceinline__ float dpp_from_next_lane(float v)  // lane l receives lane l+1 (lane 63: lane 0)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x134, 0xF, 0xF, true));
}

// one cell, in one of the reference's four summation orders
template <int ORDER>
__device__ __forceinline__ float cell7(const float (&c)[7], float C, float xp, float xm, float yp, float ym, float zp, float zm)
{
    if constexpr (ORDER == ORD_STAR_CHAIN) {  // fdtd1.cu:63, every "+ c*v" one FMA
        float a = c[0] * C;
        a = __builtin_fmaf(c[1], xp, a);
        a = __builtin_fmaf(c[2], xm, a);
        a = __builtin_fmaf(c[3], yp, a);
        a = __builtin_fmaf(c[4], ym, a);
        a = __builtin_fmaf(c[5], zp, a);
        return __builtin_fmaf(c[6], zm, a);
    } else if constexpr (ORDER == ORD_STAR_GOLDEN) {  // fdtd1.cu:46-49: centre + parenthesised pairs, separate multiplies and adds
        float a = c[0] * C;
        a = a + (c[1] * xp + c[2] * xm);
        a = a + (c[3] * yp + c[4] * ym);
        return a + (c[5] * zp + c[6] * zm);
    } else if constexpr (ORDER == ORD_JACOBI_CHAIN) {  // jacobi.cu:31-38: ce*E + cw*W + cs*S + cn*N + ct*T + cb*B + cc*C, fused
        float a = c[1] * xp;
        a = __builtin_fmaf(c[2], xm, a);
        a = __builtin_fmaf(c[3], yp, a);
        a = __builtin_fmaf(c[4], ym, a);
        a = __builtin_fmaf(c[5], zp, a);
        a = __builtin_fmaf(c[6], zm, a);
        return __builtin_fmaf(c[0], C, a);
    } else {  // jacobi.cu:478-485: the same sum with separate multiplies and adds
        static_assert(ORDER == ORD_JACOBI_GOLDEN, "unknown order");
        return c[1] * xp + c[2] * xm + c[3] * yp + c[4] * ym + c[5] * zp + c[6] * zm + c[0] * C;
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

// SEG > 1: a "strip" shape for the few quads a row has left after its last full x tile.  Nothing in the march couples one lane with
// another except the x-neighbour moves between adjacent lanes and the rows of other waves in the SAME lane, so a wave row of 64
// lanes can just as well be SEG independent segments of 64/SEG quads: segment s of a workgroup works on y tile by*SEG + s of the
// same narrow x range.  Only the per-lane constants know; each segment is an x tile with its one quad of overlap on either side
// (what the lane moves carry across a segment's ends lands in those quads), 64/SEG - 2 quads stored.
template <int KT, int NW, int RY, int RX, int ORDER, int SEG = 1>
__global__ __launch_bounds__(NW * 64) void tb_r1_kernel(const TbParams p)
{
    static_assert(KT == 2 || KT == 3, "two or three fused steps");
    static_assert(SEG == 1 || (RX == 1 && 64 % SEG == 0 && 64 / SEG >= 4), "strips are cut out of single-quad wave rows");
    constexpr int WQ = 64 * RX / SEG;   // quads of a tile row
    constexpr int ROWS = NW * RY;       // rows a tile loads
    constexpr int CYO = ROWS - 2 * KT;  // rows a tile stores
    static_assert(CYO > 0, "tile too small for its overlap");
    // boundary rows exchanged through LDS: [parity][level][wave][side][quad column][lane].
    // Two fused steps double-buffer the exchange (one barrier per iteration); three would not fit
    // in 160 KB that way, so they share one buffer and close every iteration with a second barrier.
    constexpr int PAR = (KT == 2) ? 2 : 1;
    constexpr int NS = (RY == 1) ? 1 : 2;  // a wave of one row publishes it once, for both neighbours
    __shared__ float4 ex[PAR][KT][NW][NS][RX][64];

    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    int bx, by, bz;
    {
        SweepParams g;  // only the grid fields are read by logical_tile
        g.gx = p.gx, g.gy = p.gy, g.gz = p.gz;
        logical_tile(g, bx, by, bz);
    }
    // Rows wider than one wave's 64*RX quads are cut into x tiles that overlap by one quad on each
    // side: the outermost quad of a tile only feeds its neighbours' lower-level values (a level-KT
    // cell reaches KT <= 4 cells, i.e. never beyond the adjacent quad) and is stored by the next tile.
    const int tq0 = p.qbase + bx * p.qstep;
    const bool xtiled = p.xt != 0;
    int zs, ze, zrange_end;
    chunk_planes(p, bz, zs, ze, zrange_end);
    (void)zrange_end;
    if (zs >= ze) return;
    const int seg = (SEG == 1) ? 0 : lane / WQ;    // per lane in a strip shape
    const int yt0 = 1 - KT + (by * SEG + seg) * CYO;  // first row the tile loads (may be negative)
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
            const int lq = (SEG == 1) ? lane + 64 * i : lane % WQ;
            const int qx = tq0 + lq;
            const int x0 = 4 * qx;
            const bool xown = !xtiled || (lq >= 1 && lq < WQ - 1);
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
    auto zshell = [&](int z) { return (p.zg0 + z <= 0) || (p.zg0 + z >= p.nzg - 1); };  // the whole plane is shell

    // ---- queues: q[l] = three planes of time level l (l = 0: input).  An iteration that consumes
    //      input plane pz finds, in the slots named (OLD, MID, NEW) for its phase,
    //          level 0:      planes pz-2, pz-1, pz
    //          level l >= 1: a dead plane, planes pz-l-2, pz-l-1
    //      and every level writes its new plane (level l: plane pz-l; level 0: the load of plane
    //      pz+1 at the end) into slot OLD, so all queues rotate together by renaming. -------------
    float4 q[KT][3][RY][RX], mv[RY][RX];
    const int pz0 = zs - KT + 2;  // first iteration: the first level-1 plane a stored cell depends on is zs-KT+1
    {
#pragma unroll
        for (int k = 0; k < 3; k++) {
            const __amdgpu_buffer_rsrc_t r = in_rsrc(pz0 - 2 + k);
#pragma unroll
            for (int j = 0; j < RY; j++)
#pragma unroll
                for (int i = 0; i < RX; i++) {
                    q[0][k][j][i] = buf_ld4(r, goff[j][i]);
#pragma unroll
                    for (int l = 1; l < KT; l++) q[l][k][j][i] = make_float4(0.f, 0.f, 0.f, 0.f);
                }
        }
        // level-1 shell values for plane pz0-1
        const bool zsh = zshell(pz0 - 1);
        const __amdgpu_buffer_rsrc_t r = mid_rsrc(pz0 - 1);
#pragma unroll
        for (int j = 0; j < RY; j++)
#pragma unroll
            for (int i = 0; i < RX; i++) mv[j][i] = buf_ld4(r, zsh ? goff[j][i] : moff[j][i]);
    }

    const int wu = (w > 0) ? w - 1 : 0, wd = (w < NW - 1) ? w + 1 : NW - 1;
    int par = 0;
    int pz = pz0;  // input plane consumed by the current iteration
    const int pend = ze + KT;

    auto iteration = [&](auto ph) {
        constexpr int PH = decltype(ph)::value;
        constexpr int OLD = PH % 3, MID = (PH + 1) % 3, NEW = (PH + 2) % 3;
        // publish this wave's first/last rows of every level's centre plane (level 0: input plane
        // pz-1 in MID; level l: plane pz-l-1 in NEW)
#pragma unroll
        for (int l = 0; l < KT; l++)
#pragma unroll
            for (int i = 0; i < RX; i++) {
                ex[par][l][w][0][i][lane] = q[l][l == 0 ? MID : NEW][0][i];
                if constexpr (NS == 2) ex[par][l][w][1][i][lane] = q[l][l == 0 ? MID : NEW][RY - 1][i];
            }
        __syncthreads();
        // ---- levels 1..KT: level l at plane pz-l from level l-1 planes pz-l-1, pz-l, pz-l+1 ------
#pragma unroll
        for (int l = 1; l <= KT; l++) {
            if (l == KT) {
                // Loads for the next iteration, as soon as the lower levels have used the oldest input plane (level 2 takes
                // its shell cells from it) and level 1 its shell source:
                // input plane pz+1 into the dead slot, shell source of level-1 plane pz.  With three levels they are issued
                // ahead of the last level's arithmetic and of the iteration's stores, so that the wait for them at the top
                // of the next iteration does not wait for the stores as well.
                const __amdgpu_buffer_rsrc_t rn = in_rsrc((pz + 1 < pend) ? pz + 1 : -1);
                const bool zshn = zshell(pz);
                const __amdgpu_buffer_rsrc_t rm = mid_rsrc((pz + 1 < pend) ? pz : -1);
#pragma unroll
                for (int j = 0; j < RY; j++)
#pragma unroll
                    for (int i = 0; i < RX; i++) {
                        if constexpr (KT == 3) q[0][OLD][j][i] = buf_ld4(rn, goff[j][i]);
                        mv[j][i] = buf_ld4(rm, zshn ? goff[j][i] : moff[j][i]);
                    }
            }
            // Rows closer than l to the tile's edge have no valid level-l value (the neighbouring tile
            // computes them): a wave all of whose rows are such rows skips the level.  What it leaves in
            // its queue and publishes is only ever read for rows that are invalid one level up.
            if (w * RY + RY - 1 < l || w * RY > ROWS - 1 - l) continue;  // uniform
            float4 up[RX], dn[RX], res[RY][RX];
#pragma unroll
            for (int i = 0; i < RX; i++) {
                up[i] = ex[par][l - 1][wu][NS - 1][i][lane];
                dn[i] = ex[par][l - 1][wd][0][i][lane];
            }
            if (l == 1)
                level_step<RY, RX, ORDER>(res, q[0][OLD], q[0][MID], q[0][NEW], up, dn, c, lane);
            else
                level_step<RY, RX, ORDER>(res, q[l - 1][MID], q[l - 1][NEW], q[l - 1][OLD], up, dn, c, lane);
            // Shell cells keep the value the cube written by that step holds there: level 1 takes
            // it from `mid`; level l >= 2 finds it in the queue of level l-2, whose plane pz-l has the
            // same shell (levels of equal parity live in cubes with equal shells).
            const float4(&sh)[RY][RX] = (l == 1) ? mv : (l == 2) ? q[0][OLD] : q[l >= 3 ? l - 2 : 0][MID];
            if (l < KT) {
                const bool zsh = zshell(pz - l);
#pragma unroll
                for (int j = 0; j < RY; j++)
#pragma unroll
                    for (int i = 0; i < RX; i++) {
                        const unsigned m = zsh ? 0u : imask[j][i];
                        q[l][OLD][j][i].x = (m & 1u) ? res[j][i].x : sh[j][i].x;
                        q[l][OLD][j][i].y = (m & 2u) ? res[j][i].y : sh[j][i].y;
                        q[l][OLD][j][i].z = (m & 4u) ? res[j][i].z : sh[j][i].z;
                        q[l][OLD][j][i].w = (m & 8u) ? res[j][i].w : sh[j][i].w;
                    }
            } else {
                const bool st = (pz - KT >= zs) && (pz - KT < ze);
                const __amdgpu_buffer_rsrc_t ro = make_rsrc(p.out + (long long)(pz - KT) * p.plane, st ? plane_bytes : 0u);
#pragma unroll
                for (int j = 0; j < RY; j++)
#pragma unroll
                    for (int i = 0; i < RX; i++) {
                        const unsigned m = imask[j][i];
                        buf_st4(ro, soff[j][i], (m & 1u) ? res[j][i].x : sh[j][i].x, (m & 2u) ? res[j][i].y : sh[j][i].y,
                                (m & 4u) ? res[j][i].z : sh[j][i].z, (m & 8u) ? res[j][i].w : sh[j][i].w);
                    }
            }
        }
        if constexpr (KT == 2) {  // two levels: the store above takes its shell cells from the slot the next input plane goes into
            const __amdgpu_buffer_rsrc_t rn = in_rsrc((pz + 1 < pend) ? pz + 1 : -1);
#pragma unroll
            for (int j = 0; j < RY; j++)
#pragma unroll
                for (int i = 0; i < RX; i++) q[0][OLD][j][i] = buf_ld4(rn, goff[j][i]);
        }
        if constexpr (PAR == 2)
            par ^= 1;
        else
            __syncthreads();  // everyone has read this iteration's rows before the next ones are published
        pz++;
    };

    while (pz < pend) {
        unroll_phases<0, 3>([&](auto ph) {
            if (pz < pend) iteration(ph);
        });
    }
}

struct TbVariantInfo {
    int order;  // ORD_STAR_* | ORD_JACOBI_* | ORD_SPC_*
    int kt;     // fused steps per launch
    int radius; // 1 (tb_r1_kernel) or 2..4 (tb_rm_kernel)
    int nw, ry, rx, threads, lds_bytes;
    int seg = 1;     // > 1: strip shape, a wave row is `seg` segments of 64/seg quads, each in another y tile
    int shells = 0;  // 1: honours TbParams::even and never writes a shell cell of `out` (in-place loops over two cubes)
    void (*launch)(const TbParams &, dim3 grid, hipStream_t);
};
int num_tb_variants();
const TbVariantInfo *tb_variant(int i);
void register_tb_variant(const TbVariantInfo &v);

template <int KT, int NW, int RY, int RX, int ORDER, int SEG>
void tb_launch(const TbParams &p, dim3 grid, hipStream_t s)
{
    hipLaunchKernelGGL((tb_r1_kernel<KT, NW, RY, RX, ORDER, SEG>), grid, dim3(NW * 64), 0, s, p);
}

template <int KT, int NW, int RY, int RX, int ORDER, int SEG = 1>
struct TbRegistrar {
    TbRegistrar()
    {
        TbVariantInfo v;
        v.order = ORDER;
        v.kt = KT;
        v.radius = 1;
        v.nw = NW;
        v.ry = RY;
        v.rx = RX;
        v.seg = SEG;
        v.threads = NW * 64;
        v.lds_bytes = (KT == 2 ? 2 : 1) * KT * NW * (RY == 1 ? 1 : 2) * RX * 64 * 16;
        v.launch = &tb_launch<KT, NW, RY, RX, ORDER, SEG>;
        register_tb_variant(v);
    }
};
#define FDTDX_TB_VARIANT(KT, NW, RY, RX, ORDER) static ::fdtdx::TbRegistrar<KT, NW, RY, RX, ORDER> tbreg_##KT##_##NW##_##RY##_##RX##_##ORDER;
#define FDTDX_TB_STRIP_VARIANT(KT, NW, RY, ORDER, SEG) static ::fdtdx::TbRegistrar<KT, NW, RY, 1, ORDER, SEG> tbstrip_##KT##_##NW##_##RY##_##ORDER##_##SEG;

}  // namespace fdtdx
