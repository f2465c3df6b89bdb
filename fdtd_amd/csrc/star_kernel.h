// star_kernel.h -- radius-M star sweep for gfx950 (wave64), one time step per launch.
//
// Replaces prop_inner_kernel / prop_center_kernel / prop_halo_kernel / prop_small_kernel of the
// reference (stencil.h:127-335) and the device lambdas of fdtd1..8.cu (fdtd1.cu:61-64 ...
// fdtd8.cu:82-93) and jacobi.cu:31-38.
//
// Shape of the computation
//   * A workgroup owns a tile of TQ float4 "quads" (4*TQ cells, x unit-stride) by CY = LY*RY rows
//     and marches a chunk of z planes.  Thread (tx, ty) owns quad tx of rows ty, ty+LY, ...
//   * z neighbours live in a per-thread register queue of 2M+1 float4 (planes z-M..z+M); each
//     plane is read from HBM exactly once per tile as one coalesced 16-byte-per-lane load.
//   * x and y neighbours come from the centre plane staged in LDS: every thread publishes its
//     centre quad, and the tile's halo (M rows above/below, ceil(M/4) quads left/right, no
//     corners -- a star never reads them) is fetched by extra loads spread over the workgroup.
//     The x neighbours of a quad are the aligned quads left/right of it, so shifted operands are
//     picked by register naming, not by unaligned LDS reads or lane shuffles.
//   * LDS is double buffered: one barrier per plane.
//   * All global traffic goes through raw buffer loads/stores whose per-lane offset is pushed
//     out of range for lanes that must not touch memory (rows/quads outside the grid, shell
//     cells): the hardware returns 0 / drops the store.  The z loop therefore has no divergent
//     branch and a fixed number of memory operations per iteration, so the waits the compiler
//     places are counted (vmcnt(N)) and the loads of plane z+M+1 stay in flight across the
//     barrier, the FMA chain and the stores of plane z.
//   * Quads that straddle the shell in x are stored whole: the shell components carry the
//     value the output cube already holds there (read once per plane by those lanes only), so
//     a row needs exactly one store instruction.
//   * The z loop is unrolled once per queue slot with rotating slot names, so the queue advances
//     by renaming instead of register moves, and new planes are loaded straight into their final
//     slot PF planes ahead of their first use (more bytes in flight per CU).
//   * The update runs on cell pairs (v_pk_mul/add/fma_f32), stores are non-temporal: the output
//     is not read again by this sweep, and keeping it out of the way leaves the cache to the
//     halo rows neighbouring tiles share.
#pragma once
#include "sweep_common.h"

namespace fdtdx {

template <int ORDER>
__device__ __forceinline__ float pair_or_chain(float acc, float cp, float vp, float cm, float vm)
{
    if constexpr (ORDER == ORD_STAR_GOLDEN) {
        return acc + (cp * vp + cm * vm);
    } else {
        acc = __builtin_fmaf(cp, vp, acc);
        return __builtin_fmaf(cm, vm, acc);
    }
}

// the same update on two neighbouring cells at once
template <int ORDER>
__device__ __forceinline__ f32x2 pair_or_chain2(f32x2 acc, float cp, f32x2 vp, float cm, f32x2 vm)
{
    if constexpr (ORDER == ORD_STAR_GOLDEN) {
        return acc + (cp * vp + cm * vm);
    } else {
        acc = __builtin_elementwise_fma(f32x2{cp, cp}, vp, acc);
        return __builtin_elementwise_fma(f32x2{cm, cm}, vm, acc);
    }
}

constexpr int star_nhq(int M, int TQ, int LY, int RY)  // halo quads of a tile: M rows above and below, ceil(M/4) quads left and right
{
    return 2 * M * TQ + LY * RY * 2 * ((M + 3) / 4);
}
constexpr int star_nhk(int M, int TQ, int LY, int RY)
{
    return (2 * M * TQ + LY * RY * 2 * ((M + 3) / 4) + TQ * LY - 1) / (TQ * LY);
}

// Halo ring: a tile's halo quads of plane P are the own quads of the neighbouring tiles, which fetch plane P as their
// queue's newest plane M+PF-1 steps before P becomes the centre plane.  Fetched then, the halo finds the lines in the L2
// the neighbours share (fetched when P is the centre plane, they have long been evicted: at radius 8 the halo costs as
// much fabric traffic as if no tile shared anything).  Each thread parks its halo quads in an LDS ring of M+PF-1 planes
// until their plane comes up; the ring entries are private to the thread, so no barrier is involved.
#ifndef FDTDX_RING_MIN_M
#define FDTDX_RING_MIN_M 3
#endif
// Timing experiments only (results are wrong with any bit set): which part of a plane step costs what.
//   1: no neighbour reads and no arithmetic (the result is the own quad)      2: the halo quads skip the LDS ring (dropped)
//   4: no halo loads        8: no re-read of the output's shell components    16: no store        32: no LDS publish, no barrier
//   64: the y neighbours are not read from LDS (the own quad stands in: same arithmetic)   128: nor the x neighbours
//   1024: the last (partial) round of halo quads is skipped altogether: no load, no ring entry, no publish
//   256: the halo quads left and right of the tile are not loaded (the loads are issued with an out-of-range offset)   512: nor those above and below
#ifndef FDTDX_ABLATE
#define FDTDX_ABLATE 0
#endif
// Late publish: a step publishes the NEXT centre plane to LDS after its own arithmetic, behind one barrier at the top of the
// step; otherwise it publishes its own centre plane first and meets the barrier before the arithmetic.  The LDS stores of a
// publish are slow (a 16-byte store takes about 13 cycles to move from the registers, three times a 16-byte read) and nothing of
// the same wave overlaps them when the barrier follows at once; behind the arithmetic they overlap other waves' reads and
// multiplies.  Measured at 512^3 on fields that stay finite, 32 x 32-quad tiles: radius 7 gains 5 % (226 -> 215 us per sweep),
// radius 8 2-4 % (241 -> 232-236; it fits its 128 registers in this order only together with the same-step fetch below: 101
// registers spilled without), radius 6 is level (its default shape, two rows per lane on 512 lanes: level over 100 steps, 2 % slower over the
// first 20), radius 3-5 lose 2-6 % (their halo loads move behind the arithmetic with the
// publish).  Fetching the halo at the top of the step instead of its end was measured too: level at radius 7-8, 6-11 % slower
// at radius 3, 4 and 6.  So it is a per-instantiation choice (star_late).
// Same-step fetch: the shell components of a partly stored quad of the output are fetched by the step that stores it (as the
// step's first load, so that waiting for it waits for nothing else) instead of one step ahead: four registers fewer per row, no
// difference in time by itself (radius 6-8).  Not for the composite update, which reads the old output of every cell.
// FDTDX_LATE_PUBLISH / FDTDX_EO_SAME_STEP = 0 or 1 force either for every kernel of a translation unit (experiments).
#ifndef FDTDX_LATE_PUBLISH
#define FDTDX_LATE_PUBLISH -1
#endif
#ifndef FDTDX_EO_SAME_STEP
#define FDTDX_EO_SAME_STEP -1
#endif
// The shapes that run adjacent row pairs on 512 lanes together with the late publish (both described below): radius 8 on tiles of
// 16 x 64 and 32 x 32 quads x rows, radius 7 on 32 x 32.  With 256 registers per lane the two queues fit without the squeeze of the
// 1024-lane shapes, the pair reads 2M + 8 quads from LDS per plane instead of 2 (2M + 4), and the publish hides behind the
// arithmetic.  What that is worth depends on the values in the cubes (the sweeps run at the power limit): at 512^3 on fields kept
// near 0.5 by normalised coefficients radius 8 takes 224-229 us per sweep on 16 x 64 tiles and 231-234 on 32 x 32 against 237-244
// for 32 x 32 tiles on 1024 lanes (and 297 for the 16 x 64 tile with rows LY apart and the early publish); on the drivers' own
// fields, which grow by a factor of two per step, 232.7 (32 x 32) and 233-236 (16 x 64) against 235.7.  Radius 7: 213 against 217
// on the normalised fields, 220.5 against 217.3 on the drivers'.  Radius 5 and 6 gain nothing either way.
constexpr bool star_pairs_late(int M, int TQ, int LY, int RY, int ORDER)
{
    return ORDER == ORD_STAR_CHAIN && RY == 2 && ((M == 8 && ((TQ == 16 && LY == 32) || (TQ == 32 && LY == 16))) || (M == 7 && TQ == 32 && LY == 16));
}
// Adjacent rows: a lane with two rows owns rows 2*ty and 2*ty+1 instead of ty and ty+LY.  The y neighbours of the two rows then
// overlap (rows -M .. M+1 around the first: 2M reads from LDS for the pair instead of 4M, the two own quads come from registers);
// the update walks the radius once for both rows and hands each neighbour quad from the row that reads it first to the other.
// FDTDX_ADJ_ROWS = 0 or 1 forces either for every two-row kernel of a translation unit (experiments).
#ifndef FDTDX_ADJ_ROWS
#define FDTDX_ADJ_ROWS -1
#endif
constexpr bool star_adjacent(int M, int TQ, int LY, int RY, int ORDER)
{
    if (RY != 2 || !(ORDER == ORD_STAR_CHAIN || ORDER == ORD_STAR_GOLDEN)) return false;
    if (FDTDX_ADJ_ROWS >= 0) return FDTDX_ADJ_ROWS != 0;
    // the host loops' order at radius 5: 32 x 32 tiles on 512 lanes run 527 Gcell/s with rows LY apart, 557-567 with adjacent rows
    // (one row per lane on 32 x 16 tiles: 487-493; from radius 6 up two rows per lane spill in this order)
    if (ORDER == ORD_STAR_GOLDEN && M == 5 && TQ == 32 && LY == 16) return true;
    return star_pairs_late(M, TQ, LY, RY, ORDER);
}
constexpr bool star_late(int M, int TQ, int LY, int RY, int ORDER)
{
    if (FDTDX_LATE_PUBLISH >= 0) return FDTDX_LATE_PUBLISH != 0;
    if (star_pairs_late(M, TQ, LY, RY, ORDER)) return true;
    if (ORDER == ORD_STAR_GOLDEN && M == 8 && TQ == 32 && LY == 16 && RY == 1) return true;  // 385-388 -> 408-417 Gcell/s at 512^3 (radius 5-7: no gain)
    return ORDER == ORD_STAR_CHAIN && (M == 7 || M == 8) && TQ == 32 && LY == 32 && RY == 1;
}
constexpr bool star_eo_same_step(int M, int TQ, int LY, int RY, int ORDER)
{
    if (is_composite(ORDER)) return false;
    if (FDTDX_EO_SAME_STEP >= 0) return FDTDX_EO_SAME_STEP != 0;
    return star_late(M, TQ, LY, RY, ORDER);
}
// quads of a ring of `planes` planes: the halo quads of each plane, and one spare quad for every thread that has no halo
// quad in its last round (it parks the zeros its masked load returns there, so that the ring traffic needs no branch)
constexpr int star_ring_quads(int M, int TQ, int LY, int RY, int planes)
{
    return planes * star_nhq(M, TQ, LY, RY) + star_nhk(M, TQ, LY, RY) * TQ * LY - star_nhq(M, TQ, LY, RY);
}
constexpr int star_ring_planes(int M, int TQ, int LY, int RY, int ORDER, int PF)
{
    const int images = 2 * (LY * RY + 2 * M) * (TQ + 2 * ((M + 3) / 4)) * 16;
    const int ring = star_ring_quads(M, TQ, LY, RY, M + PF - 1 - (star_late(M, TQ, LY, RY, ORDER) ? 1 : 0)) * 16;
    // not in the separate-multiply-add order from radius 5 up: those kernels are bound by their arithmetic and registers, and
    // measured 1-10 % slower with the ring's extra LDS traffic (radius 3: 7 % faster, radius 4: level)
    if (ORDER == ORD_STAR_GOLDEN && M >= 5) return 0;
    return (M >= FDTDX_RING_MIN_M && !is_composite(ORDER) && images + ring <= 160 * 1024) ? M + PF - 1 - (star_late(M, TQ, LY, RY, ORDER) ? 1 : 0) : 0;
}

// Per-thread constants of the march.
template <int RY, int NHK>
struct Lane {
    unsigned goff[RY];    // byte offset of the own quad of row j inside a plane (or OOB)
    unsigned soff[RY];    // same for the store (OOB unless the row is computed and the quad has interior cells)
    unsigned eoff[RY];    // byte offset for re-reading the output's shell components (partial quads only)
    unsigned hgoff[NHK];  // halo quads this thread fetches
    int hloff[NHK];       // where they go in the LDS image (quad index), -1: none
    int lown[RY];         // own quad in the LDS image
    bool m0, m1, m2, m3;  // which cells of the own quad are interior in x
    int yrow[RY], x0;     // global y of row j and x of the quad's first cell (composite update only)
};

// State of the composite update (main.cu) carried along the march; unused by the other orders.
template <int RY>
struct Comp {
    float4 vq[RY];    // second input (vel) of the centre plane, prefetched one plane ahead
    float4 sq[RY];    // running sum of the fused product-accumulate, likewise
    const float *pv;  // vel, centre plane
    float *ps;        // sum, centre plane (nullptr: no fused product-accumulate)
    double kd[8];
    int zsel, ysel, xsel;  // cells on these planes are scaled later (filt): their product is added then, not here
};

// One plane of the march.  The queue holds planes z-M .. z+M+PF; logical slot k (plane z-M+k)
// lives in physical slot (k + PH) % QT, so advancing z by one is a renaming.  Slots 2M+1 ..
// 2M+PF-1 are loads still in flight; the load issued here (plane z+M+PF) takes the slot that
// plane z-M-1 has just left.
template <int M, int TQ, int LY, int RY, int ORDER, int PF, int PH>
__device__ __forceinline__ void plane_step(float4 (&q)[RY][2 * M + 1 + PF], float4 (&eo)[RY],
                                           float4 (&hv)[star_nhk(M, TQ, LY, RY)], const Lane<RY, star_nhk(M, TQ, LY, RY)> &L,
                                           float4 *buf, float4 *bufn, float4 *ring, int &rslot, const float *&pn, const float *&pc, float *&po, const float *&pe,
                                           long long ep, bool more_n, bool more_c, bool more_c2, bool more_h,
                                           unsigned plane_bytes, const float (&c)[7], Comp<RY> &cs, int gz)
{
    constexpr int HQ = (M + 3) / 4;
    constexpr int LPQ = TQ + 2 * HQ;
    constexpr int NHK = star_nhk(M, TQ, LY, RY);
    constexpr int QT = 2 * M + 1 + PF;
    constexpr int RP = star_ring_planes(M, TQ, LY, RY, ORDER, PF);
    constexpr int NT = TQ * LY;
    constexpr int NHQ = star_nhq(M, TQ, LY, RY);
#define QS(k) (((k) + PH) % QT)

    // publish a centre plane + its halo: this step's own plane into `buf` ahead of the barrier, or (late publish) the next
    // step's plane into `bufn` behind this step's arithmetic
    auto publish = [&](float4 *dst, int slot) {
        if constexpr (!(FDTDX_ABLATE & 32)) {
#pragma unroll
            for (int j = 0; j < RY; j++) dst[L.lown[j]] = q[j][slot];
        }
        if constexpr ((FDTDX_ABLATE & 2) != 0) {
            // halo quads go nowhere
        } else if constexpr (RP > 0) {
            // the halo of the published plane leaves the ring; its slot takes the halo the step before fetched
#pragma unroll
            for (int k = 0; k < NHK; k++) {
                const int hidx = (int)threadIdx.x + k * NT;
                // a wave whose lanes all lie past the last halo quad has none in this round and skips the ring (radius 8, 32 x 32
                // quads: 6 of 16 waves, 2.5 % of the sweep's time; radius 7: 1 %)
                if ((FDTDX_ABLATE & 1024) && NHK > 1 && k == NHK - 1) continue;
                if (k == NHK - 1 && __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6) * 64 + k * NT >= NHQ) continue;
                float4 *rs = ring + ((k < NHK - 1 || hidx < NHQ) ? rslot * NHQ + hidx : RP * NHQ + hidx - NHQ);
                const float4 h = *rs;
                *rs = hv[k];
                if (L.hloff[k] >= 0) dst[L.hloff[k]] = h;
            }
            rslot = (rslot + 1 == RP) ? 0 : rslot + 1;
        } else {
#pragma unroll
            for (int k = 0; k < NHK; k++)
                if (L.hloff[k] >= 0) dst[L.hloff[k]] = hv[k];
        }
    };
    constexpr bool LATE = star_late(M, TQ, LY, RY, ORDER);
    if constexpr (LATE) {
        // plane z is complete in `buf` (the step before, or the prologue, published it) and nobody reads `bufn` any more
        if constexpr (!(FDTDX_ABLATE & 32)) __syncthreads();
    } else {
        publish(buf, QS(M));
    }

    // prefetches; where there is no such plane the descriptor is empty and the loads return 0
    const unsigned nbn = more_n ? plane_bytes : 0u, nbc = more_c ? plane_bytes : 0u;
    const long long pf = plane_bytes >> 2;
    const __amdgpu_buffer_rsrc_t rn = make_rsrc(pn, nbn);        // plane z+M+PF
    // no ring: the halo of the plane the next publish needs -- plane z+1, or (late publish, which works a step ahead) z+2
    const __amdgpu_buffer_rsrc_t rc = LATE ? make_rsrc(pc + 2 * pf, more_c2 ? plane_bytes : 0u) : make_rsrc(pc + pf, nbc);
    // shell components of output plane z+1: from the output cube, or from the strip kept beside it
    const __amdgpu_buffer_rsrc_t ron = pe ? make_rsrc(pe + ep, more_c ? (unsigned)(ep * 4) : 0u) : make_rsrc(po + pf, nbc);
    const __amdgpu_buffer_rsrc_t ro = make_rsrc(po, plane_bytes);
    auto load_halo = [&]() {
        if constexpr (!(FDTDX_ABLATE & 4)) {
            // ring: the halo of the plane the own quads are fetched of (only planes that become a centre plane of this chunk need
            // one); no ring: the halo of the next plane to be published
#pragma unroll
            for (int k = 0; k < ((FDTDX_ABLATE & 1024) && NHK > 1 ? NHK - 1 : NHK); k++) hv[k] = RP > 0 ? buf_ld4(rn, more_h ? L.hgoff[k] : OOB) : buf_ld4(rc, L.hgoff[k]);
        }
    };
    float4 eo_cur[RY], vq_cur[RY], sq_cur[RY];
    const bool fuse_mul = is_composite(ORDER) && cs.ps != nullptr;  // uniform
    constexpr bool EO_NOW = star_eo_same_step(M, TQ, LY, RY, ORDER);
#pragma unroll
    for (int j = 0; j < RY; j++) {
        if constexpr (EO_NOW)  // ahead of the other loads: loads complete in the order of issue, and this one the step itself waits for
            eo_cur[j] = buf_ld4(pe ? make_rsrc(pe, (unsigned)(ep * 4)) : ro, L.eoff[j]);
        else
            eo_cur[j] = eo[j];
        q[j][QS(QT - 1)] = buf_ld4(rn, L.goff[j]);
    }
    if constexpr (is_composite(ORDER)) {  // the second input of plane z+1, for the cells this thread computes
        const __amdgpu_buffer_rsrc_t rv = make_rsrc(cs.pv + pf, nbc);
        const __amdgpu_buffer_rsrc_t rs = make_rsrc(cs.ps + pf, fuse_mul ? nbc : 0u);
#pragma unroll
        for (int j = 0; j < RY; j++) {
            vq_cur[j] = cs.vq[j];
            cs.vq[j] = buf_ld4s(rv, L.soff[j]);
            sq_cur[j] = cs.sq[j];
            cs.sq[j] = buf_ld4s(rs, L.soff[j]);
        }
    }
    if constexpr (!LATE) load_halo();
    if constexpr (!(FDTDX_ABLATE & 8) && !EO_NOW) {
#pragma unroll
        for (int j = 0; j < RY; j++)  // streamed past the caches where measured faster: full-row tiles (+2-4 % at radius 1 and 3; 3 % slower with x tiles)
            eo[j] = (is_composite(ORDER) || TQ >= 128) ? buf_ld4s(ron, L.eoff[j]) : buf_ld4(ron, L.eoff[j]);
    }
    if constexpr (!LATE && !(FDTDX_ABLATE & 32)) __syncthreads();

    const float c0 = c[0], cxp = c[1], cxm = c[2], cyp = c[3], cym = c[4], czp = c[5], czm = c[6];
    constexpr bool ADJ = star_adjacent(M, TQ, LY, RY, ORDER);
    float adj_acc[RY][4];
    if constexpr (ADJ) {
        // rows y and y+1 of one lane together: R(k) = the quad k rows below the first row's.  Row 0 reads R(-r) and R(r), row 1
        // R(1-r) and R(r+1): every quad but R(-M) and R(M+1) serves both rows, a radius step apart, and R(0), R(1) are the own quads
        constexpr int XB = 4 * HQ;
        float xw[2][4 * (2 * HQ + 1)];
#pragma unroll
        for (int j = 0; j < 2; j++) {
            const float4 *row = buf + L.lown[j];
#pragma unroll
            for (int s = -HQ; s <= HQ; s++) {
                float4 v;
                if (s == 0 || (FDTDX_ABLATE & 128)) {
                    v = q[j][QS(M)];
                } else {
                    f32x4 w = *reinterpret_cast<const f32x4 *>(&row[s]);
                    if (M % 4 != 0 && (s == -HQ || s == HQ)) asm("" : "+v"(w));  // whole-quad read (see the one-row path below)
                    v = make_float4(w[0], w[1], w[2], w[3]);
                }
                xw[j][4 * (s + HQ) + 0] = v.x, xw[j][4 * (s + HQ) + 1] = v.y, xw[j][4 * (s + HQ) + 2] = v.z, xw[j][4 * (s + HQ) + 3] = v.w;
            }
        }
#define XW(j, i) f32x2{xw[j][(i)], xw[j][(i) + 1]}
        f32x2 a01[2] = {c0 * XW(0, XB), c0 * XW(1, XB)}, a23[2] = {c0 * XW(0, XB + 2), c0 * XW(1, XB + 2)};
        const float4 *r0 = buf + L.lown[0];
        float4 up_prev = q[0][QS(M)], dn_prev = q[1][QS(M)];
#pragma unroll
        for (int r = 1; r <= M; r++) {
            const float4 up0 = (FDTDX_ABLATE & 64) ? q[0][QS(M)] : r0[-r * LPQ], dn1 = (FDTDX_ABLATE & 64) ? q[1][QS(M)] : r0[(r + 1) * LPQ];
            const float4 ypj[2] = {dn_prev, dn1}, ymj[2] = {up0, up_prev};
#pragma unroll
            for (int j = 0; j < 2; j++) {
                const float4 yp = ypj[j], ym = ymj[j];
                const float4 zp = q[j][QS(M + r)], zm = q[j][QS(M - r)];
                a01[j] = pair_or_chain2<ORDER>(a01[j], cxp, XW(j, XB + r), cxm, XW(j, XB - r));
                a01[j] = pair_or_chain2<ORDER>(a01[j], cyp, f32x2{yp.x, yp.y}, cym, f32x2{ym.x, ym.y});
                a01[j] = pair_or_chain2<ORDER>(a01[j], czp, f32x2{zp.x, zp.y}, czm, f32x2{zm.x, zm.y});
                a23[j] = pair_or_chain2<ORDER>(a23[j], cxp, XW(j, XB + 2 + r), cxm, XW(j, XB + 2 - r));
                a23[j] = pair_or_chain2<ORDER>(a23[j], cyp, f32x2{yp.z, yp.w}, cym, f32x2{ym.z, ym.w});
                a23[j] = pair_or_chain2<ORDER>(a23[j], czp, f32x2{zp.z, zp.w}, czm, f32x2{zm.z, zm.w});
            }
            up_prev = up0, dn_prev = dn1;
        }
#undef XW
#pragma unroll
        for (int j = 0; j < 2; j++) adj_acc[j][0] = a01[j].x, adj_acc[j][1] = a01[j].y, adj_acc[j][2] = a23[j].x, adj_acc[j][3] = a23[j].y;
    }
    constexpr bool ADJ_MATH = ADJ && !(FDTDX_ABLATE & 1);
#pragma unroll
    for (int j = 0; j < RY; j++) {
        const float4 *row = buf + L.lown[j];
        // x window: quads [-HQ .. +HQ] around the own quad, as a flat register array
        float xs[4 * (2 * HQ + 1)];
#pragma unroll
        for (int s = -HQ; s <= HQ; s++) {
            float4 v;
            if (s != 0 && M % 4 != 0 && (s == -HQ || s == HQ)) {
                // the outermost quads need 1..3 components only; read them whole all the same: the
                // narrower reads the compiler would pick (4/8/12 bytes at a 16-byte lane stride) run
                // into 2- to 4-way bank conflicts
                f32x4 w = *reinterpret_cast<const f32x4 *>(&row[s]);
                asm("" : "+v"(w));
                v = make_float4(w[0], w[1], w[2], w[3]);
            } else {
                v = (s == 0 || (FDTDX_ABLATE & 128)) ? q[j][QS(M)] : row[s];
            }
            xs[4 * (s + HQ) + 0] = v.x;
            xs[4 * (s + HQ) + 1] = v.y;
            xs[4 * (s + HQ) + 2] = v.z;
            xs[4 * (s + HQ) + 3] = v.w;
        }
        constexpr int XB = 4 * HQ;  // index of the own quad's first cell in xs
        float acc[4];
        if constexpr (is_composite(ORDER)) {
            // main.cu:104-118 / :55-72: cz*z + cy*y + cx*x + vel*scal*(C + wx*SUM(X+r + X-r) + wy*SUM(Y..) + wz*SUM(Z..)) - old
            float yps[M][4], zps[M][4];  // (+r) + (-r) pairs in float, as both the host and the device form them
#pragma unroll
            for (int r = 1; r <= M; r++) {
                const float4 yp = row[r * LPQ], ym = row[-r * LPQ];
                const float4 zp = q[j][QS(M + r)], zm = q[j][QS(M - r)];
                yps[r - 1][0] = yp.x + ym.x, yps[r - 1][1] = yp.y + ym.y, yps[r - 1][2] = yp.z + ym.z, yps[r - 1][3] = yp.w + ym.w;
                zps[r - 1][0] = zp.x + zm.x, zps[r - 1][1] = zp.y + zm.y, zps[r - 1][2] = zp.z + zm.z, zps[r - 1][3] = zp.w + zm.w;
            }
            const float velv[4] = {vq_cur[j].x, vq_cur[j].y, vq_cur[j].z, vq_cur[j].w};
            const float oldv[4] = {eo_cur[j].x, eo_cur[j].y, eo_cur[j].z, eo_cur[j].w};
#pragma unroll
            for (int i = 0; i < 4; i++) {
                if constexpr (ORDER == ORD_COMP_GOLDEN) {
                    double s = (double)xs[XB + i];
#pragma unroll
                    for (int r = 1; r <= M; r++) s = s + cs.kd[5] * (double)(xs[XB + i + r] + xs[XB + i - r]);
#pragma unroll
                    for (int r = 1; r <= M; r++) s = s + cs.kd[6] * (double)yps[r - 1][i];
#pragma unroll
                    for (int r = 1; r <= M; r++) s = s + cs.kd[7] * (double)zps[r - 1][i];
                    double t = cs.kd[0] * (double)gz + cs.kd[1] * (double)L.yrow[j];
                    t = t + cs.kd[2] * (double)(L.x0 + i);
                    t = t + (double)velv[i] * cs.kd[3] / cs.kd[4] * s;
                    acc[i] = (float)(t - (double)oldv[i]);
                } else {
                    const float wxf = (float)cs.kd[5], wyf = (float)cs.kd[6], wzf = (float)cs.kd[7];
                    float s = xs[XB + i];
#pragma unroll
                    for (int r = 1; r <= M; r++) s = __builtin_fmaf(wxf, xs[XB + i + r] + xs[XB + i - r], s);
#pragma unroll
                    for (int r = 1; r <= M; r++) s = __builtin_fmaf(wyf, yps[r - 1][i], s);
#pragma unroll
                    for (int r = 1; r <= M; r++) s = __builtin_fmaf(wzf, zps[r - 1][i], s);
                    float t = (float)cs.kd[0] * (float)(unsigned)gz;
                    t = __builtin_fmaf((float)cs.kd[1], (float)(unsigned)L.yrow[j], t);
                    t = __builtin_fmaf((float)cs.kd[2], (float)(unsigned)(L.x0 + i), t);
                    t = __builtin_fmaf(velv[i] * (float)(cs.kd[3] / cs.kd[4]), s, t);
                    acc[i] = t - oldv[i];
                }
            }
            if (fuse_mul) {
                // sum += out*in (main.cu:129-133) for the cells no later operation of the step changes: cells on
                // the planes filt scales get their product there; the few cells inject touches are redone by the host
                const bool rowsel = (gz == cs.zsel) || (L.yrow[j] == cs.ysel);
                const bool mi[4] = {L.m0, L.m1, L.m2, L.m3};
                float sv[4] = {sq_cur[j].x, sq_cur[j].y, sq_cur[j].z, sq_cur[j].w};
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    const float prod = (ORDER == ORD_COMP_GOLDEN) ? sv[i] + acc[i] * xs[XB + i] : __builtin_fmaf(acc[i], xs[XB + i], sv[i]);
                    sv[i] = (mi[i] && !rowsel && (L.x0 + i != cs.xsel)) ? prod : sv[i];
                }
                buf_st4(make_rsrc(cs.ps, plane_bytes), L.soff[j], sv[0], sv[1], sv[2], sv[3]);
            }
        } else if constexpr (ADJ_MATH) {
            acc[0] = adj_acc[j][0], acc[1] = adj_acc[j][1], acc[2] = adj_acc[j][2], acc[3] = adj_acc[j][3];
        } else if constexpr ((FDTDX_ABLATE & 1) != 0) {
            acc[0] = xs[XB] + hv[0].x, acc[1] = xs[XB + 1], acc[2] = xs[XB + 2], acc[3] = xs[XB + 3];  // (the halo loads stay alive)
        } else if constexpr (ORDER == ORD_STAR_CHAIN || ORDER == ORD_STAR_GOLDEN) {
            // two cells per instruction (v_pk_mul/add/fma_f32): cells 0,1 and 2,3 of the quad
#define XP(i) f32x2{xs[(i)], xs[(i) + 1]}
            f32x2 a01 = c0 * XP(XB), a23 = c0 * XP(XB + 2);
#pragma unroll
            for (int r = 1; r <= M; r++) {
                const float4 yp = (FDTDX_ABLATE & 64) ? q[j][QS(M)] : row[r * LPQ], ym = (FDTDX_ABLATE & 64) ? q[j][QS(M)] : row[-r * LPQ];
                const float4 zp = q[j][QS(M + r)], zm = q[j][QS(M - r)];
                a01 = pair_or_chain2<ORDER>(a01, cxp, XP(XB + r), cxm, XP(XB - r));
                a01 = pair_or_chain2<ORDER>(a01, cyp, f32x2{yp.x, yp.y}, cym, f32x2{ym.x, ym.y});
                a01 = pair_or_chain2<ORDER>(a01, czp, f32x2{zp.x, zp.y}, czm, f32x2{zm.x, zm.y});
                a23 = pair_or_chain2<ORDER>(a23, cxp, XP(XB + 2 + r), cxm, XP(XB + 2 - r));
                a23 = pair_or_chain2<ORDER>(a23, cyp, f32x2{yp.z, yp.w}, cym, f32x2{ym.z, ym.w});
                a23 = pair_or_chain2<ORDER>(a23, czp, f32x2{zp.z, zp.w}, czm, f32x2{zm.z, zm.w});
            }
#undef XP
            acc[0] = a01.x, acc[1] = a01.y, acc[2] = a23.x, acc[3] = a23.y;
        } else {
            // jacobi.cu:31-38 order: ce*E + cw*W + cs*S + cn*N + ct*T + cb*B + cc*C
            // coefficient slots: c0=cc, cxp=ce, cxm=cw, cyp=cs, cym=cn, czp=ct, czm=cb
            const float4 yp = row[LPQ], ym = row[-LPQ];
            const float4 zp = q[j][QS(M + 1)], zm = q[j][QS(M - 1)];
            const float ypv[4] = {yp.x, yp.y, yp.z, yp.w}, ymv[4] = {ym.x, ym.y, ym.z, ym.w};
            const float zpv[4] = {zp.x, zp.y, zp.z, zp.w}, zmv[4] = {zm.x, zm.y, zm.z, zm.w};
#pragma unroll
            for (int i = 0; i < 4; i++) {
                if constexpr (ORDER == ORD_JACOBI_GOLDEN) {
                    acc[i] = cxp * xs[XB + i + 1] + cxm * xs[XB + i - 1] + cyp * ypv[i] + cym * ymv[i] + czp * zpv[i] +
                             czm * zmv[i] + c0 * xs[XB + i];
                } else {
                    float a = cxp * xs[XB + i + 1];
                    a = __builtin_fmaf(cxm, xs[XB + i - 1], a);
                    a = __builtin_fmaf(cyp, ypv[i], a);
                    a = __builtin_fmaf(cym, ymv[i], a);
                    a = __builtin_fmaf(czp, zpv[i], a);
                    a = __builtin_fmaf(czm, zmv[i], a);
                    acc[i] = __builtin_fmaf(c0, xs[XB + i], a);
                }
            }
        }
        // (ablated store: the offset is out of range unless a result is a NaN of one particular payload, which keeps the results alive)
        buf_st4(ro, (FDTDX_ABLATE & 16) ? (__float_as_uint(acc[0] + acc[1] + acc[2] + acc[3]) == 0x7fc12345u ? L.soff[j] : OOB) : L.soff[j],
                L.m0 ? acc[0] : eo_cur[j].x, L.m1 ? acc[1] : eo_cur[j].y, L.m2 ? acc[2] : eo_cur[j].z, L.m3 ? acc[3] : eo_cur[j].w);
    }
    if constexpr (LATE) {
        publish(bufn, QS(M + 1));  // plane z+1: the next step's centre plane (its halo: from the ring, or fetched by the step before)
        load_halo();
    }
    pn += pf;
    pc += pf;
    po += pf;
    if (pe) pe += ep;
    if constexpr (is_composite(ORDER)) {
        cs.pv += pf;
        if (fuse_mul) cs.ps += pf;
    }
#undef QS
}

template <int M, int TQ, int LY, int RY, int ORDER, int PF>
__global__ __launch_bounds__(TQ *LY) void star_sweep_kernel(const SweepParams p)
{
    static_assert(ORDER == ORD_STAR_CHAIN || ORDER == ORD_STAR_GOLDEN || is_composite(ORDER) || M == 1, "jacobi is radius 1");
    constexpr int HQ = (M + 3) / 4;           // halo quads on each side in x
    constexpr int NT = TQ * LY;               // threads
    constexpr int CY = LY * RY;               // computed rows per tile
    constexpr int LROWS = CY + 2 * M;         // LDS rows (tile + y halo)
    constexpr int LPQ = TQ + 2 * HQ;          // LDS row pitch in quads (tile + x halo)
    constexpr int NHY = 2 * M * TQ;           // halo quads above/below
    constexpr int NHQ = NHY + CY * 2 * HQ;    // + halo quads left/right
    constexpr int NHK = star_nhk(M, TQ, LY, RY);
    static_assert(NHK == (NHQ + NT - 1) / NT, "halo count");
    constexpr int QT = 2 * M + 1 + PF;        // queue slots

    constexpr int RP = star_ring_planes(M, TQ, LY, RY, ORDER, PF);  // planes of halo parked in LDS (0: none)

    __shared__ float4 lds[2][LROWS * LPQ];
    __shared__ float4 ring[RP > 0 ? star_ring_quads(M, TQ, LY, RY, RP) : 1];

    const int t = threadIdx.x;
    const int tx = t % TQ, ty = t / TQ;
    int bx, by, bz;
    logical_tile(p, bx, by, bz);
    const int q0 = bx * TQ;           // first quad of the tile
    const int y0 = p.h + by * CY;     // first computed row of the tile
    int zs, ze, zrange_end;
    chunk_planes(p, bz, zs, ze, zrange_end);
    if (zs >= ze) return;                     // whole workgroup leaves together

    Lane<RY, NHK> L;
    // ---- own quads -------------------------------------------------------------------------
    const int qx = q0 + tx;
    const bool qok = qx < p.pxq;
    const int x0 = 4 * qx;
    L.m0 = (x0 + 0 >= p.h) && (x0 + 0 < p.nx - p.h);
    L.m1 = (x0 + 1 >= p.h) && (x0 + 1 < p.nx - p.h);
    L.m2 = (x0 + 2 >= p.h) && (x0 + 2 < p.nx - p.h);
    L.m3 = (x0 + 3 >= p.h) && (x0 + 3 < p.nx - p.h);
    L.x0 = x0;
    const bool mall = L.m0 && L.m1 && L.m2 && L.m3;
    const bool many = L.m0 || L.m1 || L.m2 || L.m3;
#pragma unroll
    for (int j = 0; j < RY; j++) {
        const int lrow_own = star_adjacent(M, TQ, LY, RY, ORDER) ? RY * ty + j : ty + j * LY;  // row of the tile
        const int y = y0 + lrow_own;
        const unsigned off = (unsigned)(y * p.px + 4 * qx) * 4u;
        const bool rok = qok && (y < p.ny);
        const bool cok = qok && (y < p.ny - p.h) && many;
        L.goff[j] = rok ? off : OOB;
        L.soff[j] = cok ? off : OOB;
        // the composite update reads the old output of every cell it computes, not only the shell components
        L.eoff[j] = (cok && (!mall || is_composite(ORDER))) ? off : OOB;
        if (!is_composite(ORDER) && p.shell)  // strip entry (row, side): the left such quad of a row is the one whose first cell is shell
            L.eoff[j] = (cok && !mall) ? (unsigned)(2 * y + (L.m0 ? 1 : 0)) * 16u : OOB;
        L.lown[j] = (M + lrow_own) * LPQ + HQ + tx;
        L.yrow[j] = y;
    }
    // ---- halo quads assigned to this thread --------------------------------------------------
#pragma unroll
    for (int k = 0; k < NHK; k++) {
        const int hidx = t + k * NT;
        int lrow, lcol;
        if (hidx < NHY) {
            const int rr = hidx / TQ, cc = hidx % TQ;
            lrow = (rr < M) ? rr : CY + rr;
            lcol = HQ + cc;
        } else {
            const int h2 = hidx - NHY;
            const int rr = h2 / (2 * HQ), cc = h2 % (2 * HQ);
            lrow = M + rr;
            lcol = (cc < HQ) ? cc : TQ + cc;
        }
        const int gy = y0 - M + lrow;
        const int gq = q0 - HQ + lcol;
        const bool hin = (k < NHK - 1) || (hidx < NHQ);
        const bool hok = hin && (gy < p.ny) && (gq >= 0) && (gq < p.pxq) && !((FDTDX_ABLATE & 256) && hidx >= NHY) && !((FDTDX_ABLATE & 512) && hidx < NHY);
        L.hgoff[k] = hok ? (unsigned)(gy * p.px + 4 * gq) * 4u : OOB;
        L.hloff[k] = hin ? lrow * LPQ + lcol : -1;
    }

    // ---- prologue: fill the z queue for centre plane zs ---------------------------------------
    const unsigned plane_bytes = (unsigned)(p.plane * 4);
    const float *pin = p.in + (long long)(zs - M) * p.plane;  // plane zs-M
    float4 q[RY][QT];
    float4 eo[RY];
    float4 hv[NHK];
    // q[k] = plane zs-M+k for k < QT-1; slot QT-1 is loaded by the first step
    const int zlast = zrange_end + M;  // one past the last plane the sweep may read
#pragma unroll
    for (int j = 0; j < RY; j++) {
#pragma unroll
        for (int k = 0; k < QT - 1; k++)
            q[j][k] = buf_ld4(make_rsrc(pin + (long long)k * p.plane, (zs - M + k < zlast) ? plane_bytes : 0u), L.goff[j]);
        q[j][QT - 1] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    const float *pc = p.in + (long long)zs * p.plane;  // centre plane
    float *po = p.out + (long long)zs * p.plane;
    // late publish: the first centre plane and its halo go to LDS here; everything after it is one plane further on
    constexpr bool LATE = star_late(M, TQ, LY, RY, ORDER);
    constexpr int H0 = LATE ? 1 : 0;  // first plane (counted from zs) whose halo a step of the loop publishes
    if constexpr (LATE) {
#pragma unroll
        for (int j = 0; j < RY; j++) lds[0][L.lown[j]] = q[j][M];
#pragma unroll
        for (int k = 0; k < NHK; k++) {
            const float4 h = buf_ld4(make_rsrc(pc, plane_bytes), L.hgoff[k]);
            if (L.hloff[k] >= 0) lds[0][L.hloff[k]] = h;
        }
    }
    if constexpr (RP > 0) {
        // ring slot i: halo of plane zs+H0+i; the halo of plane zs+H0+RP is in flight like the one every step fetches
#pragma unroll
        for (int i = 0; i < RP; i++)
#pragma unroll
            for (int k = 0; k < NHK; k++) {
                const int hidx = t + k * NT;
                ring[(k < NHK - 1 || hidx < NHQ) ? i * NHQ + hidx : RP * NHQ + hidx - NHQ] =
                    buf_ld4(make_rsrc(pc + (long long)(H0 + i) * p.plane, (zs + H0 + i < ze) ? plane_bytes : 0u), L.hgoff[k]);
            }
#pragma unroll
        for (int k = 0; k < NHK; k++)
            hv[k] = buf_ld4(make_rsrc(pc + (long long)(H0 + RP) * p.plane, (zs + H0 + RP < ze) ? plane_bytes : 0u), L.hgoff[k]);
    } else {
#pragma unroll
        for (int k = 0; k < NHK; k++) hv[k] = buf_ld4(make_rsrc(pc + (long long)H0 * p.plane, (zs + H0 < ze) ? plane_bytes : 0u), L.hgoff[k]);
    }
    const bool strip = !is_composite(ORDER) && p.shell != nullptr;  // uniform
    const long long ep = strip ? p.shell_plane : 0;
    const float *pe = strip ? p.shell + (long long)zs * ep : nullptr;
#pragma unroll
    for (int j = 0; j < RY; j++) eo[j] = buf_ld4(strip ? make_rsrc(pe, (unsigned)(ep * 4)) : make_rsrc(po, plane_bytes), L.eoff[j]);
    const float *pn = pc + (long long)(M + PF) * p.plane;  // plane z+M+PF, loaded while plane z is computed
    const int zn_end = min(ze, zlast - (M + PF));          // first z whose plane z+M+PF is not needed/not there

    float c[7];
#pragma unroll
    for (int i = 0; i < 7; i++) c[i] = p.c[i];
    // composite update: constants, and the second input / running sum of the first centre plane
    Comp<RY> cs;
    cs.pv = nullptr, cs.ps = nullptr;
    if constexpr (is_composite(ORDER)) {
#pragma unroll
        for (int i = 0; i < 8; i++) cs.kd[i] = p.kd[i];
        cs.zsel = p.sel[0], cs.ysel = p.sel[1], cs.xsel = p.sel[2];
        cs.pv = p.vel + (long long)zs * p.plane;
        cs.ps = p.sum ? p.sum + (long long)zs * p.plane : nullptr;
#pragma unroll
        for (int j = 0; j < RY; j++) {
            cs.vq[j] = buf_ld4s(make_rsrc(cs.pv, plane_bytes), L.soff[j]);
            cs.sq[j] = buf_ld4s(make_rsrc(cs.ps, p.sum ? plane_bytes : 0u), L.soff[j]);
        }
    }

    // one copy of the plane body per queue slot: the queue advances by renaming, never by moves
    int z = zs;
    int par = 0;
    int rslot = 0;
    while (z < ze) {
        unroll_phases<0, QT>([&](auto ph) {
            if (z < ze) {
                plane_step<M, TQ, LY, RY, ORDER, PF, decltype(ph)::value>(q, eo, hv, L, lds[par], lds[par ^ 1], ring, rslot, pn, pc, po, pe, ep, z < zn_end, z + 1 < ze,
                                                                        z + 2 < ze, z + M + PF < ze,
                                                                        plane_bytes, c, cs, p.zg0 + z);
                par ^= 1;
#ifdef FDTDX_DRIFT_PROBE
                if (p.dbg && t == 0) p.dbg[blockIdx.x * 256 + (z - zs)] = wall_clock64();
#endif
                z++;
            }
        });
    }
}

template <int M, int TQ, int LY, int RY, int ORDER, int PF>
void star_launch(const SweepParams &p, dim3 grid, hipStream_t s)
{
    hipLaunchKernelGGL((star_sweep_kernel<M, TQ, LY, RY, ORDER, PF>), grid, dim3(TQ * LY), 0, s, p);
}

template <int M, int TQ, int LY, int RY, int ORDER, int PF>
struct StarRegistrar {
    StarRegistrar()
    {
        constexpr int HQ = (M + 3) / 4;
        VariantInfo v;
        v.M = M;
        v.order = ORDER;
        v.tq = TQ;
        v.ly = LY;
        v.ry = RY;
        v.pf = PF;
        v.threads = TQ * LY;
        v.lds_bytes = 2 * (LY * RY + 2 * M) * (TQ + 2 * HQ) * 16 + (star_ring_planes(M, TQ, LY, RY, ORDER, PF) ? star_ring_quads(M, TQ, LY, RY, star_ring_planes(M, TQ, LY, RY, ORDER, PF)) * 16 : 0);
        v.launch = &star_launch<M, TQ, LY, RY, ORDER, PF>;
        register_variant(v);
    }
};

#define FDTDX_STAR_VARIANT_PF(M, TQ, LY, RY, ORDER, PF) \
    static ::fdtdx::StarRegistrar<M, TQ, LY, RY, ORDER, PF> reg_##M##_##TQ##_##LY##_##RY##_##ORDER##_##PF;
#define FDTDX_STAR_VARIANT(M, TQ, LY, RY, ORDER) FDTDX_STAR_VARIANT_PF(M, TQ, LY, RY, ORDER, 1)

}  // namespace fdtdx
