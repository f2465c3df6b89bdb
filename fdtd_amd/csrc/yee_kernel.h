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
            if (m[i]) u[i] = upd(u[i], c, du, GOLDEN), v[i] = upd(v[i], c, dv, GOLDEN), w[i] = upd(w[i], c, dw, GOLDEN);
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
