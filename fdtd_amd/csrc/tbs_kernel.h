// tbs_kernel.h -- two fused time steps of the 13-point sweep (fdtdspc.cu:64-73) in one launch.
//
// The organisation of the fused radius-1 kernel (tb_kernel.h; the reference's own fused kernel is jacobi.cu:115-400)
// carried to a stencil with diagonals: one WAVE owns RY consecutive rows over 64*RX quads, x neighbours come from the
// neighbouring lanes by DPP wave rotates, z neighbours are register queues of three planes per time level that rotate
// by renaming.  What the diagonals change:
//   * plane z+1 contributes row y+1 (cells x and x+1) and plane z-1 row y-1 (cells x and x-1): a wave needs the row
//     BELOW its last row at planes z and z+1 and the row ABOVE its first row at planes z-1 and z.  Every iteration a
//     wave publishes, per time level, its first row of the two newer planes of its queue and its last row of the two
//     older ones (LDS, one buffer with a barrier before the next publish), so nothing read from a neighbour has to be
//     kept in registers across iterations.
//   * a level-2 plane needs its level-1 neighbours' rows published first, so it trails level 1 by two planes:
//     the iteration that consumes input plane p makes level-1 plane p-1 and stores level-2 plane p-3.
// A tile of NW*RY rows recomputes 2 rows on each side and stores NW*RY-4; rows wider than 64*RX quads are cut into
// x tiles overlapping by one quad.
//
// Fixed shell (one cell wide; fdtdspc.cu:43-45 never writes it, and the two cubes start with different shells,
// fdtdspc.cu:28-29): levels of equal parity see equal shells.  Shell cells of level 1 come from `mid` (the "odd" cube),
// those of level 0 from `even` when it differs from `in` (a cube that holds level-0 values in its interior but the
// other cube's shell, which is what the previous launch of an in-place two-cube loop leaves), and the shell of `out`
// is never written, whatever it holds: partially stored quads are written cell by cell.  With that a launch equals
// two single steps bit for bit, in the device order (every "+ c*v" one FMA) or the host order (separate IEEE
// multiplies and adds, pairs parenthesised as fdtdspc.cu:46-52).
#pragma once
#include "tb_kernel.h"

namespace fdtdx {

// Two neighbouring cells at once.  The seven operands that sit in the two cells' own columns (centre, y+-1, z+-1 and the
// two z/y diagonals) are aligned register pairs and go through v_pk_mul/add/fma_f32; the six operands shifted by one cell
// in x are not, and are applied to the two halves of the accumulator one by one.  Same operations in the same order per
// cell as the scalar form, so the same bits: device order (fdtdspc.cu:66-72, every "+ c*v" one FMA) or host order
// (fdtdspc.cu:46-52, separate IEEE multiplies and adds, pairs parenthesised).
template <int GOLDEN>
__device__ __forceinline__ f32x2 cell13x2(const float (&c)[13], f32x2 C, float xp0, float xp1, float xm0, float xm1, f32x2 yp, f32x2 ym, f32x2 zp,
                                          f32x2 zm, float ypxp0, float ypxp1, float ymxm0, float ymxm1, f32x2 zpyp, f32x2 zmym, float zpypxp0,
                                          float zpypxp1, float zmymxm0, float zmymxm1)
{
    f32x2 a = c[0] * C;
    if constexpr (GOLDEN) {
        a.x = a.x + (c[1] * xp0 + c[2] * xm0), a.y = a.y + (c[1] * xp1 + c[2] * xm1);
        a = a + (c[3] * yp + c[4] * ym);
        a = a + (c[5] * zp + c[6] * zm);
        a.x = a.x + (c[7] * ypxp0 + c[8] * ymxm0), a.y = a.y + (c[7] * ypxp1 + c[8] * ymxm1);
        a = a + (c[9] * zpyp + c[10] * zmym);
        a.x = a.x + (c[11] * zpypxp0 + c[12] * zmymxm0), a.y = a.y + (c[11] * zpypxp1 + c[12] * zmymxm1);
    } else {
        auto fma2 = [](float k, f32x2 v, f32x2 acc) { return __builtin_elementwise_fma(f32x2{k, k}, v, acc); };
        a.x = __builtin_fmaf(c[1], xp0, a.x), a.y = __builtin_fmaf(c[1], xp1, a.y);
        a.x = __builtin_fmaf(c[2], xm0, a.x), a.y = __builtin_fmaf(c[2], xm1, a.y);
        a = fma2(c[3], yp, a);
        a = fma2(c[4], ym, a);
        a = fma2(c[5], zp, a);
        a = fma2(c[6], zm, a);
        a.x = __builtin_fmaf(c[7], ypxp0, a.x), a.y = __builtin_fmaf(c[7], ypxp1, a.y);
        a.x = __builtin_fmaf(c[8], ymxm0, a.x), a.y = __builtin_fmaf(c[8], ymxm1, a.y);
        a = fma2(c[9], zpyp, a);
        a = fma2(c[10], zmym, a);
        a.x = __builtin_fmaf(c[11], zpypxp0, a.x), a.y = __builtin_fmaf(c[11], zpypxp1, a.y);
        a.x = __builtin_fmaf(c[12], zmymxm0, a.x), a.y = __builtin_fmaf(c[12], zmymxm1, a.y);
    }
    return a;
}

// One time level of one plane for row j of a wave's RY rows.  zm/cen/zp: the wave's rows of planes z-1, z, z+1;
// up_zm/up_z: the row above the wave's first row at planes z-1 and z; dn_z/dn_zp: the row below its last row at
// planes z and z+1.
template <int J, int RY, int RX, int GOLDEN>
__device__ __forceinline__ void spc_row(float4 (&res)[RX], const float4 (&zm)[RY][RX], const float4 (&cen)[RY][RX], const float4 (&zp)[RY][RX],
                                        const float4 (&up_zm)[RX], const float4 (&up_z)[RX], const float4 (&dn_z)[RX],
                                        const float4 (&dn_zp)[RX], const float (&c)[13], int lane)
{
    constexpr int JP = (J < RY - 1) ? J + 1 : J, JM = (J > 0) ? J - 1 : J;
    float lw[RX], rx[RX], dlw[RX], urx[RX], zmdlw[RX], zpurx[RX];
#pragma unroll
    for (int i = 0; i < RX; i++) {
        const float4 U = (J < RY - 1) ? cen[JP][i] : dn_z[i], D = (J > 0) ? cen[JM][i] : up_z[i];
        const float4 ZPU = (J < RY - 1) ? zp[JP][i] : dn_zp[i], ZMD = (J > 0) ? zm[JM][i] : up_zm[i];
        lw[i] = dpp_from_prev_lane(cen[J][i].w);
        rx[i] = dpp_from_next_lane(cen[J][i].x);
        dlw[i] = dpp_from_prev_lane(D.w);
        urx[i] = dpp_from_next_lane(U.x);
        zmdlw[i] = dpp_from_prev_lane(ZMD.w);
        zpurx[i] = dpp_from_next_lane(ZPU.x);
    }
#pragma unroll
    for (int i = 0; i < RX; i++) {
        const float4 C = cen[J][i];
        const float4 U = (J < RY - 1) ? cen[JP][i] : dn_z[i], D = (J > 0) ? cen[JM][i] : up_z[i];
        const float4 ZP = zp[J][i], ZM = zm[J][i];
        const float4 ZPU = (J < RY - 1) ? zp[JP][i] : dn_zp[i], ZMD = (J > 0) ? zm[JM][i] : up_zm[i];
        // lane 0's left neighbour is lane 63 of the previous quad column, lane 63's right neighbour lane 0 of the next
        const bool sl = (i > 0) && (lane == 0), sr = (i < RX - 1) && (lane == 63);
        const int il = i > 0 ? i - 1 : 0, ir = i < RX - 1 ? i + 1 : i;
        const float Lw = sl ? lw[il] : lw[i], DLw = sl ? dlw[il] : dlw[i], ZMDLw = sl ? zmdlw[il] : zmdlw[i];
        const float Rx = sr ? rx[ir] : rx[i], URx = sr ? urx[ir] : urx[i], ZPURx = sr ? zpurx[ir] : zpurx[i];
        const f32x2 r01 = cell13x2<GOLDEN>(c, f32x2{C.x, C.y}, C.y, C.z, Lw, C.x, f32x2{U.x, U.y}, f32x2{D.x, D.y}, f32x2{ZP.x, ZP.y},
                                           f32x2{ZM.x, ZM.y}, U.y, U.z, DLw, D.x, f32x2{ZPU.x, ZPU.y}, f32x2{ZMD.x, ZMD.y}, ZPU.y, ZPU.z,
                                           ZMDLw, ZMD.x);
        const f32x2 r23 = cell13x2<GOLDEN>(c, f32x2{C.z, C.w}, C.w, Rx, C.y, C.z, f32x2{U.z, U.w}, f32x2{D.z, D.w}, f32x2{ZP.z, ZP.w},
                                           f32x2{ZM.z, ZM.w}, U.w, URx, D.y, D.z, f32x2{ZPU.z, ZPU.w}, f32x2{ZMD.z, ZMD.w}, ZPU.w, ZPURx,
                                           ZMD.y, ZMD.z);
        res[i] = make_float4(r01.x, r01.y, r23.x, r23.y);
    }
}

__device__ __forceinline__ float4 select4(unsigned m, const float4 &a, const float4 &b)  // component k: bit k of m ? a : b
{
    return make_float4((m & 1u) ? a.x : b.x, (m & 2u) ? a.y : b.y, (m & 4u) ? a.z : b.z, (m & 8u) ? a.w : b.w);
}

// SEG > 1: the strip shape of tb_kernel.h for the few quads a row has left after its last full x tile: a wave row is SEG
// independent segments of 64/SEG quads, segment s working on y tile by*SEG + s.  What differs from one lane to the next
// then includes the row: "this row is shell throughout" is a per-lane fact (a wave's segments lie in different y tiles), so
// the choice between two buffer descriptors that the whole-width shapes make per wave becomes two loads and a select.
template <int NW, int RY, int RX, int GOLDEN, int SEG = 1>
__global__ __launch_bounds__(NW * 64) void tb_spc_kernel(const TbParams p)
{
    constexpr int KT = 2;
    constexpr int ROWS = NW * RY;
    constexpr int CYO = ROWS - 2 * KT;  // rows a tile stores
    static_assert(CYO > 0, "tile too small for its overlap");
    static_assert(SEG == 1 || (RX == 1 && 64 % SEG == 0 && 64 / SEG >= 4), "strips are cut out of single-quad wave rows");
    constexpr int WQ = 64 * RX / SEG;  // quads of a tile row
    // rows the waves show each other, per level: [first row @newest plane | first row @centre plane | last row @centre plane |
    // last row @oldest plane][quad column][lane].  One buffer: written at the top of an iteration, read during it, with a
    // barrier on either side.
    __shared__ float4 ex[KT][NW][4][RX][64];

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
    const int seg = (SEG == 1) ? 0 : lane / WQ;          // per lane in a strip shape
    const int yt0 = 1 - KT + (by * SEG + seg) * CYO;  // first row the tile loads (may be negative)
    const int yown0 = yt0 + KT, yown1 = yown0 + CYO;

    // per quad: byte offset inside a plane (rows/quads outside the plane: out of range) and flags:
    // bits 0-3: component k is an interior cell in x and y; bit 5: the quad is stored.
    // per row: shq = offset of the one quad of the row, among those this lane holds, that has x shell cells (the host
    // makes sure there is at most one: rows of exactly 65 quads are not run through this kernel); yfull bit j: every cell
    // of row j is shell (or the row is outside the grid)
    unsigned goff[RY][RX], flags[RY][RX], shq[RY];
    unsigned yfull = 0;
#pragma unroll
    for (int j = 0; j < RY; j++) {
        const int y = yt0 + w * RY + j;
        const bool ydom = (y >= 0) && (y < p.ny);
        const bool yin = (y >= 1) && (y < p.ny - 1);
        const bool yown = (y >= yown0) && (y < yown1);
        if (!yin) yfull |= 1u << j;
        shq[j] = OOB;
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
            goff[j][i] = qdom ? (unsigned)(y * p.px + x0) * 4u : OOB;
            flags[j][i] = m | ((qdom && yown && xown && m != 0u) ? 32u : 0u);
            if (qdom && yin && x0 < p.nx && m != 0xFu) shq[j] = goff[j][i];
        }
    }
    if constexpr (SEG == 1) yfull = __builtin_amdgcn_readfirstlane(yfull);  // (per lane in a strip)
    float c[13];
#pragma unroll
    for (int k = 0; k < 13; k++) c[k] = p.c[k];

    const unsigned plane_bytes = (unsigned)(p.plane * 4);
    const bool patch = p.even != p.in;  // level-0 shell cells come from `even`
    auto rsrc_of = [&](const float *cube, int z) { return make_rsrc(cube + (long long)z * p.plane, (z >= 0 && z < p.nz) ? plane_bytes : 0u); };
    auto zshell = [&](int z) { return (p.zg0 + z <= 0) || (p.zg0 + z >= p.nzg - 1); };  // the whole plane is shell

    // Merge the shell cells of plane z of `cube` into the wave's rows of that plane: x shell cells from the quad loaded
    // beside the plane (sx), rows and planes that are shell throughout by loading them whole here (few waves, few planes).
    auto merge_shell = [&](float4 (&q)[RY][RX], const float4 (&sx)[RY], const float *cube, int z) {
        const bool zsh = zshell(z);
#pragma unroll
        for (int j = 0; j < RY; j++) {
            if constexpr (SEG > 1) {
                const bool full = zsh || ((yfull >> j) & 1u);  // per lane
                if (__builtin_amdgcn_ballot_w64(full) != 0) {  // uniform, and rare: the first and last y tiles only
                    const __amdgpu_buffer_rsrc_t r = rsrc_of(cube, z);
#pragma unroll
                    for (int i = 0; i < RX; i++) {
                        const float4 t = buf_ld4(r, full ? goff[j][i] : OOB);
                        q[j][i] = full ? t : select4(flags[j][i], q[j][i], sx[j]);
                    }
                } else {
#pragma unroll
                    for (int i = 0; i < RX; i++) q[j][i] = select4(flags[j][i], q[j][i], sx[j]);
                }
            } else if (zsh || ((yfull >> j) & 1u)) {  // uniform
                const __amdgpu_buffer_rsrc_t r = rsrc_of(cube, z);
#pragma unroll
                for (int i = 0; i < RX; i++) q[j][i] = buf_ld4(r, goff[j][i]);
            } else {
#pragma unroll
                for (int i = 0; i < RX; i++) q[j][i] = select4(flags[j][i], q[j][i], sx[j]);
            }
        }
    };

    // Queues, by phase PH = iteration number mod 3, slots (OLD, MID, NEW) = ((PH+0)%3, (PH+1)%3, (PH+2)%3).  The iteration
    // that consumes input plane pz finds level 0: planes pz-2, pz-1, pz; level 1: planes pz-4, pz-3, pz-2 (the newest still
    // without its shell cells).  It asks for input plane pz+1 (and the shell cells that go with it and with level-1 plane
    // pz-1) first, so that the loads have the whole iteration to land and are older than the iteration's stores (the wait
    // for them then does not wait for the stores); makes level-2 plane pz-3 (stored) and level-1 plane pz-1 (over plane
    // pz-4); and moves the loaded plane over plane pz-2.
    float4 q0[3][RY][RX], q1[3][RY][RX], nx[RY][RX];
    float4 mvx[RY], evx[RY];  // x shell cells: `mid` at the level-1 plane being made, `even` at the input plane in flight
    const int pz0 = zs - 1;
#pragma unroll
    for (int k = 0; k < 3; k++) {
        const int z = pz0 - 2 + k;
        const __amdgpu_buffer_rsrc_t r = rsrc_of(p.in, z), re = rsrc_of(p.even, patch ? z : -1);
        const bool zsh = zshell(z);
#pragma unroll
        for (int j = 0; j < RY; j++)
#pragma unroll
            for (int i = 0; i < RX; i++) {
                float4 v = buf_ld4(r, goff[j][i]);
                if (patch) v = select4(zsh ? 0u : flags[j][i], v, buf_ld4(re, goff[j][i]));
                q0[k][j][i] = v;
                q1[k][j][i] = make_float4(0.f, 0.f, 0.f, 0.f);
            }
    }
#pragma unroll
    for (int j = 0; j < RY; j++) mvx[j] = evx[j] = make_float4(0.f, 0.f, 0.f, 0.f);

    const int wu = (w > 0) ? w - 1 : 0, wd = (w < NW - 1) ? w + 1 : NW - 1;
    int pz = pz0;
    const int pend = ze + 3;  // the last iteration stores level-2 plane ze-1 = pz-3

    auto iteration = [&](auto ph) {
        constexpr int PH = decltype(ph)::value;
        constexpr int OLD = PH % 3, MID = (PH + 1) % 3, NEW = (PH + 2) % 3;
        merge_shell(q1[NEW], mvx, p.mid, pz - 2);
        {   // the loads of this iteration, ahead of everything else
            const bool more = pz + 1 < pend;
            const __amdgpu_buffer_rsrc_t rn = rsrc_of(p.in, more ? pz + 1 : -1);
            const __amdgpu_buffer_rsrc_t re = rsrc_of(p.even, (more && patch) ? pz + 1 : -1);
            const __amdgpu_buffer_rsrc_t rm = rsrc_of(p.mid, more ? pz - 1 : -1);
#pragma unroll
            for (int j = 0; j < RY; j++) {
                // a row (or plane) that is shell throughout is read from `even` in the first place: a choice between two
                // descriptors, uniform over the wave, instead of a second load later
                const bool whole = more && patch && (zshell(pz + 1) || ((yfull >> j) & 1u));
                if constexpr (SEG > 1) {  // `whole` differs from lane to lane: both loads, each lane's offset in range for one of them
#pragma unroll
                    for (int i = 0; i < RX; i++) {
                        const float4 a = buf_ld4(rn, whole ? OOB : goff[j][i]), b = buf_ld4(re, whole ? goff[j][i] : OOB);
                        nx[j][i] = whole ? b : a;
                    }
                } else {
#pragma unroll
                    for (int i = 0; i < RX; i++) nx[j][i] = buf_ld4(whole ? re : rn, goff[j][i]);
                }
                mvx[j] = buf_ld4(rm, shq[j]);
                evx[j] = buf_ld4(re, shq[j]);
            }
        }
        // publish, level 0: first row @pz and @pz-1, last row @pz-1 and @pz-2; level 1: the same two planes lower
#pragma unroll
        for (int i = 0; i < RX; i++) {
            ex[0][w][0][i][lane] = q0[NEW][0][i];
            ex[0][w][1][i][lane] = q0[MID][0][i];
            ex[0][w][2][i][lane] = q0[MID][RY - 1][i];
            ex[0][w][3][i][lane] = q0[OLD][RY - 1][i];
            ex[1][w][0][i][lane] = q1[NEW][0][i];
            ex[1][w][1][i][lane] = q1[MID][0][i];
            ex[1][w][2][i][lane] = q1[MID][RY - 1][i];
            ex[1][w][3][i][lane] = q1[OLD][RY - 1][i];
        }
        __syncthreads();
        // One level of the plane in the middle of a queue, row by row; the rows above the wave's first row and below its
        // last row are read where they are needed.  (Every wave computes every row: the two rows at each edge of the tile
        // come out wrong and are neither stored nor used by a row that is, and an instruction stream without branches
        // lets the waits on the loads above be counted past the stores.)
        auto level = [&](auto lv, const float4 (&zm)[RY][RX], const float4 (&cen)[RY][RX], const float4 (&zp)[RY][RX], auto &&sink) {
            constexpr int L = decltype(lv)::value;
            unroll_phases<0, RY>([&](auto jt) {
                constexpr int J = decltype(jt)::value;
                float4 up_zm[RX], up_z[RX], dn_z[RX], dn_zp[RX], res[RX];
#pragma unroll
                for (int i = 0; i < RX; i++) {
                    if constexpr (J == 0) up_z[i] = ex[L][wu][2][i][lane], up_zm[i] = ex[L][wu][3][i][lane];
                    if constexpr (J == RY - 1) dn_zp[i] = ex[L][wd][0][i][lane], dn_z[i] = ex[L][wd][1][i][lane];
                }
                spc_row<J, RY, RX, GOLDEN>(res, zm, cen, zp, up_zm, up_z, dn_z, dn_zp, c, lane);
                sink(jt, res);
                __builtin_amdgcn_sched_barrier(0);
            });
        };
        {   // level 2 at plane pz-3 from level-1 planes pz-4, pz-3, pz-2
            const bool st = (pz - 3 >= zs) && (pz - 3 < ze);
            const __amdgpu_buffer_rsrc_t ro = make_rsrc(p.out + (long long)(pz - 3) * p.plane, st ? plane_bytes : 0u);
            level(PhaseTag<1>{}, q1[OLD], q1[MID], q1[NEW], [&](auto jt, const float4 (&res)[RX]) {
                constexpr int J = decltype(jt)::value;
#pragma unroll
                for (int i = 0; i < RX; i++) {
                    const unsigned f = flags[J][i], so = (f & 32u) ? goff[J][i] : OOB;
                    // whole quads in one store; quads that straddle the x shell cell by cell, so that `out` keeps its shell
                    buf_st4(ro, (f & 15u) == 15u ? so : OOB, res[i].x, res[i].y, res[i].z, res[i].w);
                    const unsigned sp = (f & 15u) != 15u ? so : OOB;
                    buf_st1(ro, (f & 1u) ? sp : OOB, res[i].x);
                    buf_st1(ro, (f & 2u) ? sp + 4u : OOB, res[i].y);
                    buf_st1(ro, (f & 4u) ? sp + 8u : OOB, res[i].z);
                    buf_st1(ro, (f & 8u) ? sp + 12u : OOB, res[i].w);
                }
            });
        }
        __builtin_amdgcn_sched_barrier(0);
        // level 1 at plane pz-1 from level-0 planes pz-2, pz-1, pz, over level-1 plane pz-4 (shell cells: next iteration)
        level(PhaseTag<0>{}, q0[OLD], q0[MID], q0[NEW], [&](auto jt, const float4 (&res)[RX]) {
            constexpr int J = decltype(jt)::value;
#pragma unroll
            for (int i = 0; i < RX; i++) q1[OLD][J][i] = res[i];
        });
        __syncthreads();  // every wave has read the rows it needs before the next ones are published
        // input plane pz+1 takes the place of plane pz-2
#pragma unroll
        for (int j = 0; j < RY; j++)
#pragma unroll
            for (int i = 0; i < RX; i++) q0[OLD][j][i] = nx[j][i];
        if (patch && !zshell(pz + 1)) {  // the x shell cells of the rows that are not shell throughout
#pragma unroll
            for (int j = 0; j < RY; j++)
#pragma unroll
                for (int i = 0; i < RX; i++) q0[OLD][j][i] = select4(((yfull >> j) & 1u) ? 0xFu : flags[j][i], q0[OLD][j][i], evx[j]);
        }
        pz++;
    };

    while (pz < pend) {
        unroll_phases<0, 3>([&](auto ph) {
            if (pz < pend) iteration(ph);
        });
    }
}

template <int NW, int RY, int RX, int GOLDEN, int SEG>
void tbs_launch(const TbParams &p, dim3 grid, hipStream_t s)
{
    hipLaunchKernelGGL((tb_spc_kernel<NW, RY, RX, GOLDEN, SEG>), grid, dim3(NW * 64), 0, s, p);
}

template <int NW, int RY, int RX, int GOLDEN, int SEG = 1>
struct TbsRegistrar {
    TbsRegistrar()
    {
        TbVariantInfo v;
        v.order = GOLDEN ? ORD_SPC_GOLDEN : ORD_SPC_CHAIN;
        v.radius = 1;
        v.kt = 2;
        v.nw = NW;
        v.ry = RY;
        v.rx = RX;
        v.threads = NW * 64;
        v.lds_bytes = 2 * NW * 4 * RX * 64 * 16;
        v.shells = 1;
        v.seg = SEG;
        v.launch = &tbs_launch<NW, RY, RX, GOLDEN, SEG>;
        register_tb_variant(v);
    }
};
#define FDTDX_TBS_VARIANT(NW, RY, RX, GOLDEN) static ::fdtdx::TbsRegistrar<NW, RY, RX, GOLDEN> tbsreg_##NW##_##RY##_##RX##_##GOLDEN;
#define FDTDX_TBS_STRIP_VARIANT(NW, RY, GOLDEN, SEG) static ::fdtdx::TbsRegistrar<NW, RY, 1, GOLDEN, SEG> tbsstrip_##NW##_##RY##_##GOLDEN##_##SEG;

}  // namespace fdtdx
