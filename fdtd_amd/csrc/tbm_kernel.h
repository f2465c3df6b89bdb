// tbm_kernel.h -- two fused time steps of a radius-M star sweep (fdtd2.cu ...) in one launch.
//
// The radius-1 fused kernel (tb_kernel.h) carried over to wider stencils: one WAVE owns RY rows over
// 64*RX quads, x neighbours come from the neighbouring lanes by DPP wave rotates (M cells = the
// adjacent quad for M <= 4), y neighbours are rows of the same wave or of the waves above/below, published once per
// time level and iteration to LDS (double-buffered: one barrier per iteration), z neighbours are
// register queues of 2M+1 planes per time level that rotate by renaming (z loop unrolled 2M+1 times).
// A tile of NW*RY rows recomputes 2M rows on each side and stores NW*RY-4M.  The iteration that consumes
// input plane p makes level-1 plane p-M and stores level-2 plane p-2M.
//
// Fixed shell (M cells wide, fdtd2.cu:43-45 leaves it untouched in both cubes): level-1 shell cells come
// from `mid` (the cube a single step from `in` would write), level-2 shell components of partially
// stored quads from `in`, whose shell `out` must already carry.  With that a launch is bit-identical
// to two single steps, in the device order (fdtd2.cu:66-69) or the host order (fdtd2.cu:46-52).
#pragma once
#include "tb_kernel.h"

namespace fdtdx {

// One time level of one plane for the RY rows of a wave.  q[(KC + r) % QN] = plane z+r, q[(KC - r + QN) % QN] =
// plane z-r (slots rotate by renaming).  Row j's y neighbours at distance r are rows of the same wave
// (registers) or rows of the waves above/below (LDS image `exl` of this level: [wave][row][quad column][lane]).
template <int M, int NW, int RY, int RX, int KC, int GOLDEN>
__device__ __forceinline__ void star_rows(float4 (&res)[RY][RX], const float4 (&q)[2 * M + 1][RY][RX], const float4 (*exl)[RY][RX][64],
                                          const float (&c)[7], int w, int lane)
{
    constexpr int QN = 2 * M + 1;
#pragma unroll
    for (int j = 0; j < RY; j++) {
        float lt[RX][M], rt[RX][M];  // last M components of the left lane's quad, first M of the right lane's
#pragma unroll
        for (int i = 0; i < RX; i++) {
            const float4 C = q[KC][j][i];
            const float cv[4] = {C.x, C.y, C.z, C.w};
#pragma unroll
            for (int k = 0; k < M; k++) {
                lt[i][k] = dpp_from_prev_lane(cv[4 - M + k]);
                rt[i][k] = dpp_from_next_lane(cv[k]);
            }
        }
#pragma unroll
        for (int i = 0; i < RX; i++) {
            const float4 C = q[KC][j][i];
            float xs[4 + 2 * M];
#pragma unroll
            for (int k = 0; k < M; k++) {
                // lane 0's left neighbour is lane 63 of the previous quad column, lane 63's right neighbour lane 0 of the next
                xs[k] = (i > 0 && lane == 0) ? lt[i - 1][k] : lt[i][k];
                xs[M + 4 + k] = (i < RX - 1 && lane == 63) ? rt[i + 1][k] : rt[i][k];
            }
            xs[M] = C.x, xs[M + 1] = C.y, xs[M + 2] = C.z, xs[M + 3] = C.w;
            // two cells per instruction (v_pk_mul/add/fma_f32), spelled out: cells 0,1 and 2,3 of the quad; per cell the same
            // operations in the same order as the scalar form
            // a pair that starts on an odd component of a quad is one v_pk_mov_b32 away from the two register pairs it straddles
            auto xpair = [&](auto kt) {
                constexpr int k = decltype(kt)::value;
                if constexpr (((k - M) & 1) == 0 || k - 1 < 0 || k + 2 >= 4 + 2 * M) {
                    return f32x2{xs[k], xs[k + 1]};
                } else {
                    const f32x2 lo = {xs[k - 1], xs[k]}, hi = {xs[k + 1], xs[k + 2]};
                    return __builtin_shufflevector(lo, hi, 1, 2);
                }
            };
#define XP(k) xpair(PhaseTag<(k)>{})
            f32x2 a01 = c[0] * XP(M), a23 = c[0] * XP(M + 2);
            unroll_phases<1, M + 1>([&](auto rt) {
                constexpr int r = decltype(rt)::value;
                // rows t+r and t-r of the tile (t = w*RY + j): wave offset and row inside that wave are constants
                constexpr int BIG = 64 * RY;
                const int jp = j + r, jm = j - r;
                const int dwp = jp / RY, rp = jp % RY;
                const int dwm = (jm + BIG) / RY - 64, rm = (jm + BIG) % RY;
                const float4 Yp = (dwp == 0) ? q[KC][rp][i] : exl[min(w + dwp, NW - 1)][rp][i][lane];
                const float4 Ym = (dwm == 0) ? q[KC][rm][i] : exl[max(w + dwm, 0)][rm][i][lane];
                const float4 Zp = q[(KC + r) % QN][j][i], Zm = q[(KC - r + QN) % QN][j][i];
                auto term = [&](f32x2 acc, float cp, f32x2 vp, float cm, f32x2 vm) {
                    if constexpr (GOLDEN) {  // fdtd2.cu:46-52: (+r, -r) pairs, separate multiplies and adds
                        return acc + (cp * vp + cm * vm);
                    } else {  // fdtd2.cu:66-69: x+r, x-r, y+r, y-r, z+r, z-r, every "+ c*v" fused
                        acc = __builtin_elementwise_fma(f32x2{cp, cp}, vp, acc);
                        return __builtin_elementwise_fma(f32x2{cm, cm}, vm, acc);
                    }
                };
                a01 = term(a01, c[1], XP(M + r), c[2], XP(M - r));
                a01 = term(a01, c[3], f32x2{Yp.x, Yp.y}, c[4], f32x2{Ym.x, Ym.y});
                a01 = term(a01, c[5], f32x2{Zp.x, Zp.y}, c[6], f32x2{Zm.x, Zm.y});
                a23 = term(a23, c[1], XP(M + 2 + r), c[2], XP(M + 2 - r));
                a23 = term(a23, c[3], f32x2{Yp.z, Yp.w}, c[4], f32x2{Ym.z, Ym.w});
                a23 = term(a23, c[5], f32x2{Zp.z, Zp.w}, c[6], f32x2{Zm.z, Zm.w});
            });
#undef XP
            const float a[4] = {a01.x, a01.y, a23.x, a23.y};
            res[j][i] = make_float4(a[0], a[1], a[2], a[3]);
        }
    }
}

// SEG > 1: the strip shape of tb_r1_kernel (tb_kernel.h): a wave row is SEG segments of 64/SEG quads, segment s in y tile by*SEG + s.
template <int M, int NW, int RY, int RX, int GOLDEN, int SEG = 1>
__global__ __launch_bounds__(NW * 64) void tb_rm_kernel(const TbParams p)
{
    static_assert(M >= 2 && M <= 4, "x neighbours must lie in the adjacent quad");
    static_assert(SEG == 1 || (RX == 1 && 64 % SEG == 0 && 64 / SEG >= 8), "strips are cut out of single-quad wave rows");
    constexpr int WQ = 64 * RX / SEG;  // quads of a tile row
    constexpr int KT = 2;
    constexpr int QN = 2 * M + 1;
    constexpr int ROWS = NW * RY;
    constexpr int CYO = ROWS - 2 * KT * M;  // rows a tile stores
    static_assert(CYO > 0, "tile too small for its overlap");
    __shared__ float4 ex[2][KT][NW][RY][RX][64];  // [parity][level][wave][row][quad column][lane]

    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    int bx, by, bz;
    {
        SweepParams g;
        g.gx = p.gx, g.gy = p.gy, g.gz = p.gz;
        logical_tile(g, bx, by, bz);
    }
    const int tq0 = p.qbase + bx * p.qstep;
    const bool xtiled = p.xt != 0;
    int zs, ze, zrange_end;
    chunk_planes(p, bz, zs, ze, zrange_end);
    (void)zrange_end;
    if (zs >= ze) return;
    const int seg = (SEG == 1) ? 0 : lane / WQ;            // per lane in a strip shape
    const int yt0 = M - KT * M + (by * SEG + seg) * CYO;  // first row the tile loads (may be negative)
    const int yown0 = yt0 + KT * M, yown1 = yown0 + CYO;

    unsigned goff[RY][RX], moff[RY][RX], soff[RY][RX], imask[RY][RX];
#pragma unroll
    for (int j = 0; j < RY; j++) {
        const int y = yt0 + w * RY + j;
        const bool ydom = (y >= 0) && (y < p.ny);
        const bool yin = (y >= M) && (y < p.ny - M);
        const bool yown = (y >= yown0) && (y < yown1);
#pragma unroll
        for (int i = 0; i < RX; i++) {
            const int lq = (SEG == 1) ? lane + 64 * i : lane % WQ;
            const int qx = tq0 + lq;
            const int x0 = 4 * qx;
            // x tiles overlap by the quads two levels reach into (one for radius 2, two for radius 3 and 4): a tile's
            // outermost OVQ quads on either side are a neighbouring tile's to store
            constexpr int OVQ = (KT * M + 3) / 4;
            const bool xown = !xtiled || (lq >= OVQ && lq < WQ - OVQ);
            unsigned m = 0;
#pragma unroll
            for (int k = 0; k < 4; k++)
                if (yin && x0 + k >= M && x0 + k < p.nx - M) m |= 1u << k;
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
    auto zshell = [&](int z) { return (p.zg0 + z < M) || (p.zg0 + z >= p.nzg - M); };  // the whole plane is shell

    // Queues.  The iteration that consumes input plane pz finds, counting slots from the oldest (k = 0),
    //     level 0: planes pz-2M .. pz in k = 0 .. 2M
    //     level 1: a dead plane in k = 0, planes pz-3M .. pz-M-1 in k = 1 .. 2M
    // makes level-1 plane pz-M into the dead slot, level-2 plane pz-2M from level-1 planes pz-3M .. pz-M,
    // and loads input plane pz+1 over plane pz-2M; slot k of phase PH is register slot (PH + k) % QN.
    float4 q0[QN][RY][RX], q1[QN][RY][RX], mv[RY][RX];
    const int pz0 = zs;
#pragma unroll
    for (int k = 0; k < QN; k++) {
        const __amdgpu_buffer_rsrc_t r = in_rsrc(pz0 - 2 * M + k);
#pragma unroll
        for (int j = 0; j < RY; j++)
#pragma unroll
            for (int i = 0; i < RX; i++) {
                q0[k][j][i] = buf_ld4(r, goff[j][i]);
                q1[k][j][i] = make_float4(0.f, 0.f, 0.f, 0.f);
            }
    }
    {
        const bool zsh = zshell(pz0 - M);
        const __amdgpu_buffer_rsrc_t r = mid_rsrc(pz0 - M);
#pragma unroll
        for (int j = 0; j < RY; j++)
#pragma unroll
            for (int i = 0; i < RX; i++) mv[j][i] = buf_ld4(r, zsh ? goff[j][i] : moff[j][i]);
    }

    // rows closer than l*M to the tile's edge have no valid level-l value: a wave all of whose rows are such rows skips the level
    const bool has1 = (w * RY + RY - 1 >= M) && (w * RY <= ROWS - 1 - M);
    const bool has2 = (w * RY + RY - 1 >= 2 * M) && (w * RY <= ROWS - 1 - 2 * M);
    int par = 0;
    int pz = pz0;
    const int pend = ze + KT * M;

    auto iteration = [&](auto ph) {
        constexpr int PH = decltype(ph)::value;
#define SLOT(k) ((PH + (k)) % QN)
        // publish the rows of both levels' centre planes: level 0 plane pz-M, level 1 plane pz-2M
#pragma unroll
        for (int j = 0; j < RY; j++)
#pragma unroll
            for (int i = 0; i < RX; i++) {
                ex[par][0][w][j][i][lane] = q0[SLOT(M)][j][i];
                ex[par][1][w][j][i][lane] = q1[SLOT(M + 1)][j][i];
            }
        __syncthreads();
        float4 res[RY][RX];
        if (has1) {  // uniform
            star_rows<M, NW, RY, RX, SLOT(M), GOLDEN>(res, q0, ex[par][0], c, w, lane);
            const bool zsh = zshell(pz - M);
#pragma unroll
            for (int j = 0; j < RY; j++)
#pragma unroll
                for (int i = 0; i < RX; i++) {
                    const unsigned m = zsh ? 0u : imask[j][i];
                    q1[SLOT(0)][j][i].x = (m & 1u) ? res[j][i].x : mv[j][i].x;
                    q1[SLOT(0)][j][i].y = (m & 2u) ? res[j][i].y : mv[j][i].y;
                    q1[SLOT(0)][j][i].z = (m & 4u) ? res[j][i].z : mv[j][i].z;
                    q1[SLOT(0)][j][i].w = (m & 8u) ? res[j][i].w : mv[j][i].w;
                }
        }
        if (has2) {
            star_rows<M, NW, RY, RX, SLOT(M + 1), GOLDEN>(res, q1, ex[par][1], c, w, lane);
            const bool st = (pz - 2 * M >= zs) && (pz - 2 * M < ze);
            const __amdgpu_buffer_rsrc_t ro = make_rsrc(p.out + (long long)(pz - 2 * M) * p.plane, st ? plane_bytes : 0u);
#pragma unroll
            for (int j = 0; j < RY; j++)
#pragma unroll
                for (int i = 0; i < RX; i++) {
                    const unsigned m = imask[j][i];
                    const float4 sh = q0[SLOT(0)][j][i];  // input plane pz-2M: the shell `out` keeps
                    buf_st4(ro, soff[j][i], (m & 1u) ? res[j][i].x : sh.x, (m & 2u) ? res[j][i].y : sh.y, (m & 4u) ? res[j][i].z : sh.z,
                            (m & 8u) ? res[j][i].w : sh.w);
                }
        }
        {
            const bool more = pz + 1 < pend;
            const __amdgpu_buffer_rsrc_t rn = in_rsrc(more ? pz + 1 : -1);
            const bool zshn = zshell(pz + 1 - M);
            const __amdgpu_buffer_rsrc_t rm = mid_rsrc(more ? pz + 1 - M : -1);
#pragma unroll
            for (int j = 0; j < RY; j++)
#pragma unroll
                for (int i = 0; i < RX; i++) {
                    q0[SLOT(0)][j][i] = buf_ld4(rn, goff[j][i]);
                    mv[j][i] = buf_ld4(rm, zshn ? goff[j][i] : moff[j][i]);
                }
        }
#undef SLOT
        par ^= 1;
        pz++;
    };

    while (pz < pend) {
        unroll_phases<0, QN>([&](auto ph) {
            if (pz < pend) iteration(ph);
        });
    }
}

template <int M, int NW, int RY, int RX, int GOLDEN, int SEG>
void tbm_launch(const TbParams &p, dim3 grid, hipStream_t s)
{
    hipLaunchKernelGGL((tb_rm_kernel<M, NW, RY, RX, GOLDEN, SEG>), grid, dim3(NW * 64), 0, s, p);
}

template <int M, int NW, int RY, int RX, int GOLDEN, int SEG = 1>
struct TbmRegistrar {
    TbmRegistrar()
    {
        TbVariantInfo v;
        v.order = GOLDEN ? ORD_STAR_GOLDEN : ORD_STAR_CHAIN;
        v.radius = M;
        v.kt = 2;
        v.nw = NW;
        v.ry = RY;
        v.rx = RX;
        v.seg = SEG;
        v.threads = NW * 64;
        v.lds_bytes = 2 * 2 * NW * RY * RX * 64 * 16;
        v.launch = &tbm_launch<M, NW, RY, RX, GOLDEN, SEG>;
        register_tb_variant(v);
    }
};
#define FDTDX_TBM_VARIANT(M, NW, RY, RX, GOLDEN) static ::fdtdx::TbmRegistrar<M, NW, RY, RX, GOLDEN> tbmreg_##M##_##NW##_##RY##_##RX##_##GOLDEN;
#define FDTDX_TBM_STRIP_VARIANT(M, NW, RY, GOLDEN, SEG) static ::fdtdx::TbmRegistrar<M, NW, RY, 1, GOLDEN, SEG> tbmstrip_##M##_##NW##_##RY##_##GOLDEN##_##SEG;

}  // namespace fdtdx
