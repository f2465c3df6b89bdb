// yee_kernel.h -- staggered-grid E/H updates: a 3-D six-field Yee step and the 2-D ex/ey/hz scheme.
//
// EXTENSION: the reference ships scalar ping-pong stencils only (fdtd1..8.cu, fdtdspc.cu, jacobi.cu);
// it has no E/H kernels, so nothing of it pins these.  Their definition is the CPU restatement in
// oracle/fdtd_oracle.c (oracle_yee3d_half, oracle_yee2d_step), against which they are bit-exact.
//
// 3-D: one half step updates three fields from the other three (H from curl E, or E from curl H), on
// the interior [1, n-1) of the same padded cubes the sweeps use.  A thread owns one quad column and
// marches z: the two fields differenced along z stay in registers from one plane to the next, the
// y and x neighbours are plain loads of lines the neighbouring lanes/rows fetch anyway (cache hits).
// Compulsory traffic: 3 fields read + 3 fields read-modify-written = 36 bytes per cell and half step.
#pragma once
#include "sweep_common.h"

namespace fdtdx {

struct YeeParams {
    float *f[6];  // Ex, Ey, Ez, Hx, Hy, Hz
    long long plane;
    int px, nx, ny;
    int z_lo, z_hi, zchunk;
    float coef;
};

// the updated fields are read once and written once per half step and not touched again before the next
// half step: streamed past the caches, which then hold the differenced fields' neighbour lines
__device__ __forceinline__ float4 ld4_stream(const float *p)
{
    const f32x4 f = __builtin_nontemporal_load(reinterpret_cast<const f32x4 *>(p));
    return make_float4(f[0], f[1], f[2], f[3]);
}
__device__ __forceinline__ void st4_stream(float *p, const float (&v)[4])
{
    f32x4 f;
    f[0] = v[0], f[1] = v[1], f[2] = v[2], f[3] = v[3];
    __builtin_nontemporal_store(f, reinterpret_cast<f32x4 *>(p));
}

__device__ __forceinline__ float dpp_prev_lane(float v)  // lane l receives lane l-1 (lane 0: lane 63)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x13C, 0xF, 0xF, true));
}

__device__ __forceinline__ float upd(float a, float c, float d, int golden) { return golden ? a + c * d : __builtin_fmaf(c, d, a); }

// WHICH = 0: H -= ch*curl(E) with forward differences; WHICH = 1: E += ce*curl(H) with backward differences.
// Field roles: (A, B, C) are differenced, (U, V, W) updated:
//   U -= c*((C[y+s] - C) - (B[z+s] - B));  V -= c*((A[z+s] - A) - (C[x+s] - C));  W -= c*((B[x+s] - B) - (A[y+s] - A))
// with s = +1, c = ch for H; for E the same shape holds with s = -1 and the sign of every difference flipped,
// which the code below writes out (it is what the oracle computes, operand for operand).
template <int WHICH, int GOLDEN>
__global__ __launch_bounds__(256) void yee3d_half_kernel(const YeeParams p)
{
    const int qx = blockIdx.x * 64 + (threadIdx.x & 63);
    const int y = 1 + blockIdx.y * 4 + (threadIdx.x >> 6);
    const int zs = p.z_lo + blockIdx.z * p.zchunk, ze = min(zs + p.zchunk, p.z_hi);
    const int x0 = 4 * qx;
    if (y >= p.ny - 1 || x0 >= p.nx - 1 || zs >= ze) return;
    const float *A = p.f[WHICH ? 3 : 0], *B = p.f[WHICH ? 4 : 1], *C = p.f[WHICH ? 5 : 2];
    float *U = p.f[WHICH ? 0 : 3], *V = p.f[WHICH ? 1 : 4], *W = p.f[WHICH ? 2 : 5];
    const int s = WHICH ? -1 : 1;
    const float c = WHICH ? p.coef : -p.coef;
    const long long row = (long long)y * p.px + x0;
    const long long sy = (long long)s * p.px, sz = (long long)s * p.plane;
    // the x neighbour of a quad's last (H) / first (E) cell lies in the adjacent quad
    const int xn = WHICH ? max(x0 - 1, 0) - x0 : 4;
    bool m[4];
#pragma unroll
    for (int i = 0; i < 4; i++) m[i] = (x0 + i >= 1) && (x0 + i < p.nx - 1);
    long long a = (long long)zs * p.plane + row;
    float4 Az, Bz;  // A and B at the plane the z difference reaches (z+1 for H, z-1 for E) ...
    float4 A0 = ld4(A + a), B0 = ld4(B + a);  // ... and at the current plane
    if (WHICH) Az = ld4(A + a + sz), Bz = ld4(B + a + sz);
    for (int z = zs; z < ze; z++, a += p.plane) {
        if (!WHICH) Az = ld4(A + a + sz), Bz = ld4(B + a + sz);
        const float4 C0 = ld4(C + a), Cy = ld4(C + a + sy), Ay = ld4(A + a + sy);
        const float Cxn = C[a + xn], Bxn = B[a + xn];
        const float a0[4] = {A0.x, A0.y, A0.z, A0.w}, b0[4] = {B0.x, B0.y, B0.z, B0.w}, c0[4] = {C0.x, C0.y, C0.z, C0.w};
        const float az[4] = {Az.x, Az.y, Az.z, Az.w}, bz[4] = {Bz.x, Bz.y, Bz.z, Bz.w};
        const float cy[4] = {Cy.x, Cy.y, Cy.z, Cy.w}, ay[4] = {Ay.x, Ay.y, Ay.z, Ay.w};
        float cx[4], bx[4];  // C and B one cell along x
        if (WHICH) {
            cx[0] = Cxn, cx[1] = c0[0], cx[2] = c0[1], cx[3] = c0[2];
            bx[0] = Bxn, bx[1] = b0[0], bx[2] = b0[1], bx[3] = b0[2];
        } else {
            cx[0] = c0[1], cx[1] = c0[2], cx[2] = c0[3], cx[3] = Cxn;
            bx[0] = b0[1], bx[1] = b0[2], bx[2] = b0[3], bx[3] = Bxn;
        }
        float4 u4 = ld4_stream(U + a), v4 = ld4_stream(V + a), w4 = ld4_stream(W + a);
        float u[4] = {u4.x, u4.y, u4.z, u4.w}, v[4] = {v4.x, v4.y, v4.z, v4.w}, w[4] = {w4.x, w4.y, w4.z, w4.w};
#pragma unroll
        for (int i = 0; i < 4; i++) {
            float du, dv, dw;
            if (WHICH) {  // backward differences: (f - f[-1])
                du = (c0[i] - cy[i]) - (b0[i] - bz[i]);
                dv = (a0[i] - az[i]) - (c0[i] - cx[i]);
                dw = (b0[i] - bx[i]) - (a0[i] - ay[i]);
            } else {      // forward differences: (f[+1] - f)
                du = (cy[i] - c0[i]) - (bz[i] - b0[i]);
                dv = (az[i] - a0[i]) - (cx[i] - c0[i]);
                dw = (bx[i] - b0[i]) - (ay[i] - a0[i]);
            }
            // computed for every cell, kept where the cell is interior (selects: per-component branches cost more)
            const float un = upd(u[i], c, du, GOLDEN), vn = upd(v[i], c, dv, GOLDEN), wn = upd(w[i], c, dw, GOLDEN);
            u[i] = m[i] ? un : u[i], v[i] = m[i] ? vn : v[i], w[i] = m[i] ? wn : w[i];
        }
        st4_stream(U + a, u), st4_stream(V + a, v), st4_stream(W + a, w);
        if (WHICH) {
            Az = A0, Bz = B0;  // the current plane is the next one's z-1
            if (z + 1 < ze) A0 = ld4(A + a + p.plane), B0 = ld4(B + a + p.plane);
        } else {
            A0 = Az, B0 = Bz;  // z+1 becomes the current plane
        }
    }
}

// ---- one launch for the whole step: H half step and E half step fused, out of place -----------------
// The two half steps of one time step read and write every field twice (72 bytes per cell).  Fused, a
// thread that has just made H at (z, y, x) needs the new H at z-1 (its own previous plane: registers),
// at y-1 (the thread row below: LDS, one extra halo row of threads per tile recomputes it at the tile
// edge) and at x-1 (the neighbouring lane: DPP; lane 0 recomputes that one cell) to finish E at the
// same cell: every field is read once and written once, 48 bytes per cell.  That cannot be done in
// place (a neighbouring tile would find fields half updated), so the step goes from six cubes into six
// others, which must already carry the shell (the step never changes it).  Same arithmetic, operand
// for operand, as the two half-step kernels above.
struct YeeFusedParams {
    const float *eo[3], *ho[3];  // Ex, Ey, Ez / Hx, Hy, Hz before the step
    float *en[3], *hn[3];        // after the step
    long long plane;
    int px, nx, ny, nz;
    int z_lo, z_hi, zchunk;
    int gx, gy, gz;  // x tiles, y tiles, z chunks (1-D launch of gx*gy*gz workgroups, dealt to the XCDs in runs)
    float ce, ch;
};

template <int LY, int GOLDEN>
__global__ __launch_bounds__(64 * (LY + 1)) void yee3d_fused_kernel(const YeeFusedParams p)
{
    __shared__ float4 sx[2][LY + 1][64], sz[2][LY + 1][64];  // new Hx, Hz of the current plane, one row per wave
    const int lane = threadIdx.x & 63;
    const int ty = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // 0: the row below the tile (H only)
    int bx, by, bz;
    {
        SweepParams g;  // only the grid fields are read by logical_tile
        g.gx = p.gx, g.gy = p.gy, g.gz = p.gz;
        logical_tile(g, bx, by, bz);  // neighbouring tiles share an L2: the halo row is a neighbour's own row
    }
    const int x0 = 4 * (bx * 64 + lane);
    const int y = by * LY + ty;  // the tile stores rows 1 + by*LY .. by*LY + LY
    const int zs = p.z_lo + bz * p.zchunk, ze = min(zs + p.zchunk, p.z_hi);
    if (zs >= ze) return;  // the whole workgroup
    const bool dom = (y < p.ny) && (x0 < p.nx);
    const bool yin = (y >= 1) && (y < p.ny - 1);
    bool m[4];
#pragma unroll
    for (int i = 0; i < 4; i++) m[i] = dom && yin && (x0 + i >= 1) && (x0 + i < p.nx - 1);
    const bool stores = (ty >= 1) && (m[0] || m[1] || m[2] || m[3]);
    const bool left = dom && (ty >= 1) && (lane == 0) && (x0 >= 4);  // this lane recomputes H at cell x0-1
    const float nch = -p.ch;
    // Lanes outside the grid read the grid's first quad instead (their results are masked and never stored), a row
    // y+1 or a cell x+4 that does not exist is replaced by the own row / quad: every load is unconditional, and
    // nothing is read beyond the cubes.
    const long long row = dom ? (long long)y * p.px + x0 : 0;
    const bool upok = dom && (y + 1 < p.ny);
    const int upo = upok ? p.px : 0;
    const bool more = dom && (x0 + 4 < p.nx);
    const int ro = more ? 4 : 0;
    auto ld = [&](const float *f, long long a) { return ld4(f + a); };
    auto lds_ = [&](const float *f, long long a) { return ld4_stream(f + a); };
    auto ldu = [&](const float *f, long long a) { return ld4(f + a + upo); };

    // Everything plane z contributes from memory.  Whatever is needed of E at plane z+1 is fetched now, while the
    // neighbouring lanes and rows fetch the same lines as their own data, and carried in registers to the next
    // plane: a line asked for again one plane later has left the L2 by then.
    struct In {
        float4 ez, ex1, ey1;   // Ez at z; Ex, Ey at z+1 (they become the next plane's own Ex, Ey)
        float4 h[3];           // H before the step
        float4 ezu, exu1;      // row y+1: Ez at z, Ex at z+1
        float ezr, eyr1;       // the cell right of the quad: Ez at z, Ey at z+1
        float lhy, lhz, lez, lex1, lexu1, ley1;  // lane 0 only, cell x0-1: Hy, Hz, Ez at z; Ex, Ex of row y+1, Ey at z+1
    };
    struct Carry {
        float4 ex, ey, exu;    // own Ex, Ey and Ex of row y+1 at the current plane
        float eyr, lex, lexu, ley;
    };
    auto fetch = [&](int z, In &v) {
        const long long a = (long long)z * p.plane + row, a1 = a + p.plane;
        v.ez = ld(p.eo[2], a);
        v.ex1 = ld(p.eo[0], a1), v.ey1 = ld(p.eo[1], a1);
        v.h[0] = lds_(p.ho[0], a), v.h[1] = lds_(p.ho[1], a), v.h[2] = lds_(p.ho[2], a);  // read once: streamed
        v.ezu = ldu(p.eo[2], a), v.exu1 = ldu(p.eo[0], a1);
        v.ezr = p.eo[2][a + ro], v.eyr1 = p.eo[1][a1 + ro];
        if (left) {
            const long long b = a - 1, b1 = a1 - 1;
            v.lhy = p.ho[1][b], v.lhz = p.ho[2][b], v.lez = p.eo[2][b];
            v.lex1 = p.eo[0][b1], v.lexu1 = p.eo[0][b1 + upo], v.ley1 = p.eo[1][b1];
        }
    };
    auto advance = [&](Carry &c, const In &v) {
        c.ex = v.ex1, c.ey = v.ey1, c.exu = v.exu1, c.eyr = v.eyr1, c.lex = v.lex1, c.lexu = v.lexu1, c.ley = v.ley1;
    };
    // H after the step at plane z for this thread's quad
    auto h_plane = [&](int z, const In &v, const Carry &c, float4 (&h)[3]) {
        h[0] = v.h[0], h[1] = v.h[1], h[2] = v.h[2];
        const bool zin = (z >= 1) && (z < p.nz - 1);
        if (!zin) return;  // uniform: a shell plane keeps its H
        const float ex0[4] = {c.ex.x, c.ex.y, c.ex.z, c.ex.w}, ey0[4] = {c.ey.x, c.ey.y, c.ey.z, c.ey.w};
        const float ez0[4] = {v.ez.x, v.ez.y, v.ez.z, v.ez.w};
        const float exz[4] = {v.ex1.x, v.ex1.y, v.ex1.z, v.ex1.w}, eyz[4] = {v.ey1.x, v.ey1.y, v.ey1.z, v.ey1.w};
        const float ezy[4] = {v.ezu.x, v.ezu.y, v.ezu.z, v.ezu.w}, exy[4] = {c.exu.x, c.exu.y, c.exu.z, c.exu.w};
        const float ezx[4] = {ez0[1], ez0[2], ez0[3], v.ezr}, eyx[4] = {ey0[1], ey0[2], ey0[3], c.eyr};
        float hx[4] = {h[0].x, h[0].y, h[0].z, h[0].w}, hy[4] = {h[1].x, h[1].y, h[1].z, h[1].w}, hz[4] = {h[2].x, h[2].y, h[2].z, h[2].w};
#pragma unroll
        for (int i = 0; i < 4; i++) {  // computed for every cell, kept where the cell is interior (selects, no branches)
            const float nx_ = upd(hx[i], nch, (ezy[i] - ez0[i]) - (eyz[i] - ey0[i]), GOLDEN);
            const float ny_ = upd(hy[i], nch, (exz[i] - ex0[i]) - (ezx[i] - ez0[i]), GOLDEN);
            const float nz_ = upd(hz[i], nch, (eyx[i] - ey0[i]) - (exy[i] - ex0[i]), GOLDEN);
            hx[i] = m[i] ? nx_ : hx[i], hy[i] = m[i] ? ny_ : hy[i], hz[i] = m[i] ? nz_ : hz[i];
        }
        h[0] = make_float4(hx[0], hx[1], hx[2], hx[3]), h[1] = make_float4(hy[0], hy[1], hy[2], hy[3]);
        h[2] = make_float4(hz[0], hz[1], hz[2], hz[3]);
    };

    In cur;
    Carry c;
    float4 hprev[3], h[3];
    {   // H after the step at plane zs-1; what the loop carries is loaded directly for that one plane
        const long long a = (long long)(zs - 1) * p.plane + row;
        c.ex = ld(p.eo[0], a), c.ey = ld(p.eo[1], a), c.exu = ldu(p.eo[0], a);
        c.eyr = p.eo[1][a + ro];
        c.lex = c.lexu = c.ley = 0.f;
        if (left) c.lex = p.eo[0][a - 1], c.lexu = p.eo[0][a - 1 + upo], c.ley = p.eo[1][a - 1];
        fetch(zs - 1, cur);
        h_plane(zs - 1, cur, c, hprev);
        advance(c, cur);
    }
    int par = 0;
    for (int z = zs; z < ze; z++, par ^= 1) {
        const long long a = (long long)z * p.plane + row;
        fetch(z, cur);  // (fetching a whole plane ahead into a second register set was measured: no faster, fewer waves fit)
        h_plane(z, cur, c, h);
        sx[par][ty][lane] = h[0], sz[par][ty][lane] = h[2];
        // new Hy, Hz at the cell left of the quad: the neighbouring lane has them, lane 0 works them out
        float hyl = dpp_prev_lane(h[1].w), hzl = dpp_prev_lane(h[2].w);
        if (left) {
            hyl = cur.lhy, hzl = cur.lhz;
            if (yin) {
                hyl = upd(hyl, nch, (cur.lex1 - c.lex) - (cur.ez.x - cur.lez), GOLDEN);
                hzl = upd(hzl, nch, (c.ey.x - c.ley) - (c.lexu - c.lex), GOLDEN);
            }
        }
        __syncthreads();
        if (ty >= 1) {  // uniform per wave
            // whole-quad LDS reads (the single components the compiler would pick, at a 16-byte lane stride, conflict)
            f32x4 wx = *reinterpret_cast<const f32x4 *>(&sx[par][ty - 1][lane]), wz = *reinterpret_cast<const f32x4 *>(&sz[par][ty - 1][lane]);
            asm("" : "+v"(wx), "+v"(wz));
            const float4 hxd = make_float4(wx[0], wx[1], wx[2], wx[3]), hzd = make_float4(wz[0], wz[1], wz[2], wz[3]);
            const float hx[4] = {h[0].x, h[0].y, h[0].z, h[0].w}, hy[4] = {h[1].x, h[1].y, h[1].z, h[1].w}, hz[4] = {h[2].x, h[2].y, h[2].z, h[2].w};
            const float hxy[4] = {hxd.x, hxd.y, hxd.z, hxd.w}, hzy[4] = {hzd.x, hzd.y, hzd.z, hzd.w};
            const float hxz[4] = {hprev[0].x, hprev[0].y, hprev[0].z, hprev[0].w}, hyz[4] = {hprev[1].x, hprev[1].y, hprev[1].z, hprev[1].w};
            const float hyx[4] = {hyl, hy[0], hy[1], hy[2]}, hzx[4] = {hzl, hz[0], hz[1], hz[2]};
            float ex[4] = {c.ex.x, c.ex.y, c.ex.z, c.ex.w}, ey[4] = {c.ey.x, c.ey.y, c.ey.z, c.ey.w}, ez[4] = {cur.ez.x, cur.ez.y, cur.ez.z, cur.ez.w};
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const float nx_ = upd(ex[i], p.ce, (hz[i] - hzy[i]) - (hy[i] - hyz[i]), GOLDEN);
                const float ny_ = upd(ey[i], p.ce, (hx[i] - hxz[i]) - (hz[i] - hzx[i]), GOLDEN);
                const float nz_ = upd(ez[i], p.ce, (hy[i] - hyx[i]) - (hx[i] - hxy[i]), GOLDEN);
                ex[i] = m[i] ? nx_ : ex[i], ey[i] = m[i] ? ny_ : ey[i], ez[i] = m[i] ? nz_ : ez[i];
            }
            if (stores) {
                st4_stream(p.hn[0] + a, hx), st4_stream(p.hn[1] + a, hy), st4_stream(p.hn[2] + a, hz);
                st4_stream(p.en[0] + a, ex), st4_stream(p.en[1] + a, ey), st4_stream(p.en[2] + a, ez);
            }
        }
        hprev[0] = h[0], hprev[1] = h[1];
        advance(c, cur);
    }
}

// ---- 2-D ex/ey/hz on [NX][NY] arrays, j fastest -----------------------------------------------------
// E half: ey[0][j] = t; ey[i][j] -= ce*(hz[i][j] - hz[i-1][j]) (i >= 1); ex[i][j] -= ce*(hz[i][j] - hz[i][j-1]) (j >= 1)
template <int GOLDEN>
__global__ void yee2d_e_kernel(float *ex, float *ey, const float *hz, int nx, int ny, float t, float ce)
{
    const int j = blockIdx.x * blockDim.x + threadIdx.x, i = blockIdx.y;
    if (j >= ny) return;
    const long long a = (long long)i * ny + j;
    const float h = hz[a];
    if (i == 0)
        ey[a] = t;
    else
        ey[a] = upd(ey[a], -ce, h - hz[a - ny], GOLDEN);
    if (j >= 1) ex[a] = upd(ex[a], -ce, h - hz[a - 1], GOLDEN);
}
// H half: hz[i][j] -= ch*(((ex[i][j+1] - ex[i][j]) + ey[i+1][j]) - ey[i][j]) (i < NX-1, j < NY-1)
template <int GOLDEN>
__global__ void yee2d_h_kernel(const float *ex, const float *ey, float *hz, int nx, int ny, float ch)
{
    const int j = blockIdx.x * blockDim.x + threadIdx.x, i = blockIdx.y;
    if (j >= ny - 1 || i >= nx - 1) return;
    const long long a = (long long)i * ny + j;
    hz[a] = upd(hz[a], -ch, ((ex[a + 1] - ex[a]) + ey[a + ny]) - ey[a], GOLDEN);
}

}  // namespace fdtdx
