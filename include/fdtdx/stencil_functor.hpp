// fdtdx/stencil_functor.hpp -- the reference's generic functor interface, for hipcc (gfx950).
//
// The closed programs of fdtdx.h cover the stencils the reference ships.  User-written stencils
// reach the device through this header instead: it keeps the calling conventions of the
// reference's propagate / inject / filt templates (stencil.h:684-768, 810-823) so that a driver's
// device lambdas compile unchanged:
//
//   sweep   f(z, y, x, addr, out, zl, sz, yl, sy, xl, sx, args...)      (fdtd1.cu:61, stencil.h:90,167,235)
//             z, y, x  global logical coordinates of the cell
//             addr     linear index of the cell in the local padded cube (ghost planes included)
//             out      base of the output cube: the functor writes out[addr]
//             zl/yl/xl the cell seen along each axis: neighbour r is zl[r*sz], yl[r*sy], xl[r*sx]
//           FDTD = true  ("star" mode): the three views live in different memories (in the reference registers
//             for z, the LDS plane for y and x; here three register views), so only pure-axis offsets are
//             legal (stencil.h:127-171);
//           FDTD = false ("box" mode): all three alias the input cube with strides (plane, pitch, 1)
//             and mixed offsets such as yl[sy+sz] are legal (stencil.h:75-93, fdtdspc.cu:70-72).
//   inject  f(i, args...) for i < n                                      (stencil.h:337-343)
//   filt    f(z, y, x, addr, args...) for every cell of the planes this rank owns (stencil.h:345-355)
//
// Cubes, transfers, decomposition and halo regions come from fdtdx::Stencil (the C ABI); this
// header only adds the templated launches, which is why it needs hipcc.  The star-mode kernel is
// the reference's scheme (z queue in registers, centre plane in LDS) laid out for wave64: tiles of
// 128 cells x 16 rows, one 16-byte quad per thread, a z queue that rotates by renaming, masked buffer
// loads issued a plane ahead, the functor called once per cell.  The box-mode kernel serves all three
// views from LDS images of the tile's planes.  This is the general path; the closed programs of
// fdtdx.h run the tuned kernels.
#pragma once
#include <hip/hip_runtime.h>

#include <algorithm>
#include <limits>

#include <map>
#include "stencil.hpp"

namespace fdtdx {

struct FunctorGeom {
    long long plane;       // floats per plane
    int px, nx, ny;        // row pitch, logical extents
    long long zg0;         // global z of local plane 0
    int z_lo, z_hi;        // local planes to visit
    int zchunk;            // planes per workgroup
    int nz_local;          // planes of the local cube (ghost planes included)
    int gx, gy, gz;        // tiles along x, y and z chunks (the marching kernels are launched as one row of gx*gy*gz workgroups)
};

namespace functor_detail {
// Raw buffer loads: a descriptor (base, byte count) in scalar registers plus a 32-bit per-lane byte offset.  An offset at
// or beyond the byte count reads as zero, which masks lanes (quads outside the grid) without a branch in the z loop.
typedef unsigned int u32x4 __attribute__((__vector_size__(16)));
typedef float f32x4 __attribute__((__vector_size__(16)));
constexpr unsigned OOB = 0x80000000u;  // beyond any plane (planes stay below 2^31 bytes, checked on the host)
__device__ __forceinline__ float4 plane_ld4(const void *plane, unsigned bytes, unsigned voff)
{
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(plane), 0, bytes, 0x00020000);
    const f32x4 f = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, voff, 0, 0));
    return make_float4(f[0], f[1], f[2], f[3]);
}
__device__ __forceinline__ float comp(const float4 &v, int i) { return i == 0 ? v.x : i == 1 ? v.y : i == 2 ? v.z : v.w; }
// compile-time loop f(Phase<PH>{}) ... f(Phase<N-1>{}): z loops whose register queue rotates by renaming
template <int I>
struct Phase {
    static constexpr int value = I;
};
template <int PH, int N, class F>
__device__ __forceinline__ void phases(F &&f)
{
    if constexpr (PH < N) {
        f(Phase<PH>{});
        phases<PH + 1, N>(f);
    }
}
// Workgroups are dealt round-robin over the 8 XCDs, each with an L2 of its own: hardware workgroup b and b+8 share one.
// Renumber so that every XCD owns a run of neighbouring tiles, which then find each other's halo rows in their L2.
// One-to-one for any count; placement changes speed only.
__device__ __forceinline__ void logical_tile(const FunctorGeom &g, int &bx, int &by, int &bz)
{
    const int nb = g.gx * g.gy * g.gz, b = (int)blockIdx.x;
    const int xcd = b & 7, slot = b >> 3, qn = nb >> 3, rn = nb & 7;
    const int lb = (xcd < rn ? xcd * (qn + 1) : rn * (qn + 1) + (xcd - rn) * qn) + slot;
    bx = lb % g.gx;
    by = (lb / g.gx) % g.gy;
    bz = lb / (g.gx * g.gy);
}
}  // namespace functor_detail

// ---- star mode --------------------------------------------------------------------------------------
// A workgroup owns a tile of TXQ quads (4*TXQ cells, x unit-stride) by LY*RY rows and marches a chunk of z planes.
// Thread (tx, ty) owns quad tx of rows ty, ty+LY, ...: it keeps the z neighbours of those cells in a register queue
// (planes z-M .. z+M+1, the last one a load in flight; the z loop is unrolled once per queue slot so that the queue
// advances by renaming, not by moves), publishes the centre quads to an LDS image of the current plane together with
// the halo rows and halo quads assigned to it (offsets worked out once, before the march), and calls the functor once
// per cell.  The reference's prop_inner/prop_center kernels (stencil.h:127-289) hand the functor zl into registers and
// yl/xl into their LDS plane; here all three are register views with stride 1 (the thread has read the quads around its
// own from the image beforehand, 16 bytes at a time), which the star-mode contract allows: only pure-axis offsets are legal.  The image is double-buffered: one barrier per plane; the
// loads of the next plane are issued before the barrier and land while the functor runs.
template <typename T, int M, int TXQ, int LY, int RY, typename gsz, typename gssz, typename F, typename... Args>
__global__ __launch_bounds__(TXQ *LY) void functor_star_kernel(FunctorGeom g, T *out, const T *in, F f, Args... args)
{
    using namespace functor_detail;
    static_assert(sizeof(T) == 4, "quads of four 4-byte cells");
    constexpr int HQ = (M + 3) / 4;                      // halo quads on each side in x
    constexpr int LPQ = TXQ + 2 * HQ, LX = 4 * LPQ;      // LDS row: quads, cells
    constexpr int CY = LY * RY, LROWS = CY + 2 * M, NT = TXQ * LY;
    constexpr int NHY = 2 * M * TXQ, NHQ = NHY + CY * 2 * HQ;  // halo quads: rows above/below, quads left/right (no corners)
    constexpr int NHK = (NHQ + NT - 1) / NT;
    constexpr int QT = 2 * M + 2;                        // queue slots: planes z-M .. z+M and the load in flight
    __shared__ float4 lds[2][LROWS * LPQ];
    // The tile's halo quads of plane P are own quads of the neighbouring tiles, which fetch plane P M steps before it is
    // the centre plane; fetched then as well, the halo finds the lines in the L2 the tiles share (M steps later they
    // are long evicted).  Each thread parks its halo quads in a ring of M planes until their plane comes up; a thread
    // without a halo quad in its last round parks the zeros of its masked load in a spare quad, so nothing branches.
    constexpr int RINGQ = M * NHQ + NHK * NT - NHQ;
    constexpr int RP = (M >= 3 && (2 * LROWS * LPQ + RINGQ) * 16 <= 160 * 1024) ? M : 0;
    __shared__ float4 ring[RP > 0 ? RINGQ : 1];
    // rows and planes are whole numbers of quads and cubes are 16-byte aligned (fdtdx_geometry): said out loud so
    // that the compiler may merge what the functor stores for the four cells of a quad
    __builtin_assume((g.px & 3) == 0);
    __builtin_assume((g.plane & 3) == 0);
    out = static_cast<T *>(__builtin_assume_aligned(out, 16));
    const int t = threadIdx.x;
    int bx, by, bz;
    logical_tile(g, bx, by, bz);
    const int q0 = bx * TXQ, y0 = M + by * CY, pxq = g.px >> 2;
    // Which thread owns which quad.  Quads that straddle the shell in x (at most one column of the tile on each side) take
    // the cell-by-cell path below, and a wave with one such lane pays for that path in full; so the threads are dealt out
    // with those columns first: they all land in the first wave, and every other wave runs the straight path only.
    int ec[2], ne = 0;
    if (M % 4 != 0 && bx == 0) ec[ne++] = M / 4;
    if ((g.nx - M) % 4 != 0) {
        const int cr = (g.nx - M) / 4 - q0;
        if (cr >= 0 && cr < TXQ && (ne == 0 || cr != ec[0])) ec[ne++] = cr;
    }
    int tx, ty;
    if (t < ne * LY) {
        ty = t / ne, tx = ec[t % ne];
    } else {
        const int j = t - ne * LY, w = TXQ - ne;
        ty = j / w, tx = j % w;
        if (ne > 0 && tx >= ec[0]) tx++;
        if (ne > 1 && tx >= ec[1]) tx++;
    }
    const int x0 = 4 * (q0 + tx);
    const int zs = g.z_lo + bz * g.zchunk;
    const int ze = min(zs + g.zchunk, g.z_hi);
    if (zs >= ze) return;
    const unsigned plane_bytes = (unsigned)(g.plane * 4);
    const bool xall = x0 >= M && x0 + 3 < g.nx - M;      // all four cells of the quad are interior in x
    unsigned goff[RY], hgoff[NHK];
    int lown[RY], hloff[NHK], yrow[RY];
    bool ycomp[RY];
#pragma unroll
    for (int j = 0; j < RY; j++) {
        const int y = y0 + ty + j * LY;
        yrow[j] = y;
        ycomp[j] = y < g.ny - M;
        goff[j] = (q0 + tx < pxq && y < g.ny) ? (unsigned)(y * g.px + x0) * 4u : OOB;
        lown[j] = (M + ty + j * LY) * LPQ + HQ + tx;
    }
#pragma unroll
    for (int k = 0; k < NHK; k++) {
        const int h = t + k * NT;
        int lrow, lcol;
        if (h < NHY) {
            const int r = h / TXQ;
            lrow = (r < M) ? r : CY + r;
            lcol = HQ + h % TXQ;
        } else {
            const int h2 = h - NHY, c = h2 % (2 * HQ);
            lrow = M + h2 / (2 * HQ);
            lcol = (c < HQ) ? c : TXQ + c;
        }
        const int gy = y0 - M + lrow, gq = q0 - HQ + lcol;
        const bool mine = h < NHQ;
        hgoff[k] = (mine && gy < g.ny && gq >= 0 && gq < pxq) ? (unsigned)(gy * g.px + 4 * gq) * 4u : OOB;
        hloff[k] = mine ? lrow * LPQ + lcol : -1;
    }
    // planes beyond the local cube do not exist: their descriptor is empty (the queue slot ahead of the last plane)
    auto plane_of = [&](int z) { return in + (long long)z * g.plane; };
    auto bytes_of = [&](int z) { return z < g.nz_local ? plane_bytes : 0u; };
    float4 q[RY][QT], hv[NHK];
#pragma unroll
    for (int j = 0; j < RY; j++) {
#pragma unroll
        for (int k = 0; k < QT - 1; k++) q[j][k] = plane_ld4(plane_of(zs - M + k), bytes_of(zs - M + k), goff[j]);
        q[j][QT - 1] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    auto ring_at = [&](int slot, int k) -> float4 & {
        const int h = t + k * NT;
        return ring[(k < NHK - 1 || h < NHQ) ? slot * NHQ + h : RP * NHQ + h - NHQ];
    };
    if constexpr (RP > 0) {
#pragma unroll
        for (int i = 0; i < RP; i++)
#pragma unroll
            for (int k = 0; k < NHK; k++) ring_at(i, k) = plane_ld4(plane_of(zs + i), bytes_of(zs + i), hgoff[k]);
    }
#pragma unroll
    for (int k = 0; k < NHK; k++) hv[k] = plane_ld4(plane_of(zs + RP), bytes_of(zs + RP), hgoff[k]);

    int z = zs, par = 0, rslot = 0;
    while (z < ze) {
        phases<0, QT>([&](auto ph) {
            constexpr int PH = decltype(ph)::value;  // logical slot k (plane z-M+k) lives in physical slot (k+PH) % QT
            if (z >= ze) return;
            float4 *buf = lds[par];
#pragma unroll
            for (int j = 0; j < RY; j++) buf[lown[j]] = q[j][(M + PH) % QT];
            if constexpr (RP > 0) {
                // this plane's halo leaves the ring; its slot takes the halo of plane z+M, fetched by the step before
#pragma unroll
                for (int k = 0; k < NHK; k++) {
                    float4 &slot = ring_at(rslot, k);
                    const float4 h = slot;
                    slot = hv[k];
                    if (hloff[k] >= 0) buf[hloff[k]] = h;
                }
                rslot = (rslot + 1 == RP) ? 0 : rslot + 1;
            } else {
#pragma unroll
                for (int k = 0; k < NHK; k++)
                    if (hloff[k] >= 0) buf[hloff[k]] = hv[k];
            }
            // next plane: own quads of plane z+M+1 into the slot plane z-M-1 has left, halo of plane z+1 (ring: z+M+1)
            const bool more = z + 1 < ze;
#pragma unroll
            for (int j = 0; j < RY; j++)
                q[j][(QT - 1 + PH) % QT] = plane_ld4(plane_of(z + M + 1), more ? bytes_of(z + M + 1) : 0u, goff[j]);
#pragma unroll
            for (int k = 0; k < NHK; k++) hv[k] = plane_ld4(plane_of(z + 1 + RP), more ? bytes_of(z + 1 + RP) : 0u, hgoff[k]);
            __syncthreads();  // the other image is only rewritten after the next barrier, when everyone has left this plane's reads behind
#pragma unroll
            for (int j = 0; j < RY; j++) {
                // the neighbourhood of the row's quad, read from the image as whole quads (16-byte reads at a 16-byte lane
                // stride are free of bank conflicts; what the functor does not touch is dropped by the compiler)
                const float4 *row = buf + lown[j];
                float4 yq[2 * M + 1], xq[2 * HQ + 1];
#pragma unroll
                for (int r = -M; r <= M; r++) yq[r + M] = (r == 0) ? q[j][(M + PH) % QT] : row[r * LPQ];
#pragma unroll
                for (int s = -HQ; s <= HQ; s++) xq[s + HQ] = (s == 0) ? q[j][(M + PH) % QT] : row[s];
                const gsz addr0 = (gsz)((long long)z * g.plane + (long long)yrow[j] * g.px + x0);
                auto cell = [&](auto ic) {
                    constexpr int i = decltype(ic)::value;
                    // the cell along each axis, as the functor's zl / yl / xl expect it: names for registers, not copies
                    T zv[2 * M + 1], yv[2 * M + 1], xv[2 * M + 1];
#pragma unroll
                    for (int k = 0; k <= 2 * M; k++) {
                        zv[k] = comp(q[j][(k + PH) % QT], i);
                        yv[k] = comp(yq[k], i);
                        constexpr int XB = 4 * HQ + i - M;  // cell i-M+k of the row, counted from the first cell of xq
                        xv[k] = comp(xq[(XB + k) / 4], (XB + k) % 4);
                    }
                    f((gsz)(g.zg0 + z), (gsz)yrow[j], (gsz)(x0 + i), (gsz)(addr0 + i), out, &zv[M], (gssz)1, &yv[M], (gssz)1, &xv[M], (gssz)1,
                      args...);
                };
                if (ycomp[j] && xall) {
                    // one straight block, so that the functor's four stores can go out as one
                    cell(Phase<0>{}), cell(Phase<1>{}), cell(Phase<2>{}), cell(Phase<3>{});
                    asm volatile("" ::: "memory");  // ends the block: the compiler would otherwise move this path's last store into a tail shared with the path below, and split the 16-byte store
                } else if (ycomp[j]) {
                    // (last cell first: were both paths to end with the same cell, the compiler would share that tail
                    // between them and split the straight path's 16-byte store)
                    if (x0 + 3 >= M && x0 + 3 < g.nx - M) cell(Phase<3>{});
                    if (x0 + 2 >= M && x0 + 2 < g.nx - M) cell(Phase<2>{});
                    if (x0 + 1 >= M && x0 + 1 < g.nx - M) cell(Phase<1>{});
                    if (x0 + 0 >= M && x0 + 0 < g.nx - M) cell(Phase<0>{});
                }
            }
            par ^= 1;
            z++;
        });
    }
}

// ---- box mode ---------------------------------------------------------------------------------------
// The functor may combine offsets along the three axes, so all three views must be one memory with fixed strides.
// Here that memory is LDS: a workgroup owns a tile of TX cells by TY rows, marches a chunk of z planes and keeps the
// planes z-M .. z+M+1 of the tile with its whole halo (corners included) as LDS images, so that every neighbour the
// functor touches is an LDS read and each plane leaves HBM once per tile.  The functor needs ONE z stride, which a
// ring of images cannot give where it wraps; every plane is therefore written to two images, QT slots apart, and the
// window around the centre plane is taken where it does not wrap.  With 2M+2 planes held, the image a new plane
// overwrites was last read two iterations earlier: one barrier per plane.  Lane i of a wave owns cell i of a row
// (4-byte LDS reads at unit stride: no bank conflicts), each thread computes CX x CY cells of the tile.
template <int M, int TX, int TY>
struct FunctorBoxShape {
    static constexpr int HQ = (M + 3) / 4, LXQ = TX / 4 + 2 * HQ, LX = 4 * LXQ, LROWS = TY + 2 * M;
    static constexpr int QT = 2 * M + 2, IMG = LROWS * LX;          // floats per image
    static constexpr size_t lds_bytes = (size_t)2 * QT * IMG * 4;
};
template <typename T, int M, int TX, int TY, int NT, typename gsz, typename gssz, typename F, typename... Args>
__global__ __launch_bounds__(NT) void functor_box_lds_kernel(FunctorGeom g, T *out, const T *in, F f, Args... args)
{
    using namespace functor_detail;
    using S = FunctorBoxShape<M, TX, TY>;
    static_assert(sizeof(T) == 4 && TX % 64 == 0 && NT % 64 == 0, "waves of 64 cells along x");
    constexpr int HQ = S::HQ, LXQ = S::LXQ, LX = S::LX, LROWS = S::LROWS, QT = S::QT, IMG = S::IMG;
    constexpr int NQ = LROWS * LXQ, NK = (NQ + NT - 1) / NT;       // quads of an image, per thread
    constexpr int RG = NT / 64, CX = TX / 64, CY = TY / RG;        // row groups; cells per thread along x and y
    static_assert(TY % RG == 0, "rows divide over the waves");
    __shared__ float4 img[2 * QT * IMG / 4];
    T *cells = reinterpret_cast<T *>(img);
    const int t = threadIdx.x, lane = t % 64, rg = t / 64;
    int bx, by, bz;
    logical_tile(g, bx, by, bz);
    const int xt = bx * TX, y0 = M + by * TY, pxq = g.px >> 2;
    const int zs = g.z_lo + bz * g.zchunk;
    const int ze = min(zs + g.zchunk, g.z_hi);
    if (zs >= ze) return;
    const unsigned plane_bytes = (unsigned)(g.plane * 4);
    unsigned goff[NK];
    int loff[NK];
#pragma unroll
    for (int k = 0; k < NK; k++) {
        const int h = t + k * NT, lrow = h / LXQ, lcol = h % LXQ;
        const int gy = y0 - M + lrow, gq = (xt >> 2) - HQ + lcol;
        const bool mine = h < NQ;
        goff[k] = (mine && gy < g.ny && gq >= 0 && gq < pxq) ? (unsigned)(gy * g.px + 4 * gq) * 4u : OOB;
        loff[k] = mine ? h : -1;
    }
    auto fetch = [&](float4 (&v)[NK], int z, bool wanted) {
        const unsigned bytes = (wanted && z < g.nz_local) ? plane_bytes : 0u;
#pragma unroll
        for (int k = 0; k < NK; k++) v[k] = plane_ld4(in + (long long)z * g.plane, bytes, goff[k]);
    };
    auto publish = [&](const float4 (&v)[NK], int z) {
        const int s = ((z % QT) + QT) % QT;
#pragma unroll
        for (int k = 0; k < NK; k++)
            if (loff[k] >= 0) img[s * (IMG / 4) + loff[k]] = v[k], img[(s + QT) * (IMG / 4) + loff[k]] = v[k];
    };
    float4 v[NK];
    for (int z = zs - M; z < zs + M; z++) {
        fetch(v, z, true);
        publish(v, z);
    }
    fetch(v, zs + M, true);
    for (int z = zs; z < ze; z++) {
        publish(v, z + M);
        fetch(v, z + M + 1, z + 1 < ze);
        __syncthreads();
        const int s = z % QT, c = s < M ? s + QT : s;             // centre image: c-M .. c+M do not wrap
#pragma unroll
        for (int b = 0; b < CY; b++) {
            const int ly = rg + b * RG, y = y0 + ly;
            if (y >= g.ny - M) continue;
#pragma unroll
            for (int a = 0; a < CX; a++) {
                const int lx = lane + 64 * a, x = xt + lx;
                if (x < M || x >= g.nx - M) continue;
                T *p = cells + c * IMG + (M + ly) * LX + 4 * HQ + lx;
                f((gsz)(g.zg0 + z), (gsz)y, (gsz)x, (gsz)((long long)z * g.plane + (long long)y * g.px + x), out, p, (gssz)IMG, p, (gssz)LX, p,
                  (gssz)1, args...);
            }
        }
    }
}

// The same interface on the input cube itself (strides plane, pitch, 1), one cell per thread: for radii whose LDS
// images would not fit.
template <typename T, int M, typename gsz, typename gssz, typename F, typename... Args>
__global__ void functor_box_kernel(FunctorGeom g, T *out, const T *in, F f, Args... args)
{
    const int x = M + blockIdx.x * blockDim.x + threadIdx.x;
    const int y = M + blockIdx.y;
    const int z = g.z_lo + blockIdx.z;
    if (x >= g.nx - M) return;
    const long long a = (long long)z * g.plane + (long long)y * g.px + x;
    T *c = const_cast<T *>(in) + a;
    f((gsz)(g.zg0 + z), (gsz)y, (gsz)x, (gsz)a, out, c, (gssz)g.plane, c, (gssz)g.px, c, (gssz)1, args...);
}

template <typename gsz, typename F, typename... Args>
__global__ void functor_inject_kernel(gsz n, F f, Args... args)
{
    const gsz i = (gsz)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) f(i, args...);
}

template <typename gsz, typename F, typename... Args>
__global__ void functor_filt_kernel(FunctorGeom g, F f, Args... args)
{
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    if (x >= g.nx) return;
    const int y = blockIdx.y, z = g.z_lo + blockIdx.z;
    f((gsz)(g.zg0 + z), (gsz)y, (gsz)x, (gsz)((long long)z * g.plane + (long long)y * g.px + x), args...);
}

#define FDTDX_HIP_ASSERT(expr)                                                                   \
    do {                                                                                         \
        hipError_t e_ = (expr);                                                                  \
        if (e_ != hipSuccess) {                                                                  \
            std::fprintf(stderr, "ERROR AT [ %s ] : %s\n", #expr, hipGetErrorString(e_));        \
            std::abort();                                                                        \
        }                                                                                        \
    } while (0)

// The reference's class with its functor-taking methods (template header of stencil.h:357).
template <typename DataType, size_t ZSize, size_t YSize, size_t XSize, size_t M, bool FDTD = true, typename gpu_size_t = uint32_t,
          typename gpu_signed_size_t = int32_t, size_t BLOCK_SIZE = 32, size_t THREADS = 256, size_t PADDINGL = 0, size_t PADDINGR = 0>
class FunctorStencil : public Stencil<DataType, ZSize, YSize, XSize, M, FDTD, gpu_size_t, gpu_signed_size_t, BLOCK_SIZE, THREADS, PADDINGL, PADDINGR> {
    using Base = Stencil<DataType, ZSize, YSize, XSize, M, FDTD, gpu_size_t, gpu_signed_size_t, BLOCK_SIZE, THREADS, PADDINGL, PADDINGR>;

    hipStream_t stream_ = nullptr;
    std::map<std::string, std::pair<void *, size_t>> flat_;  // malloc<T>(name, length): flat device buffers (stencil.h:440-447), bytes

    FunctorGeom geom(uint64_t z_lo, uint64_t z_hi) const
    {
        const fdtdx_geometry &g = this->geometry();
        // gpu_size_t addresses must reach the whole local cube (the reference's uint32_t limit, fdtd1.cu:13-14)
        if (sizeof(gpu_size_t) < 8 && g.cube_elems > (uint64_t)std::numeric_limits<gpu_size_t>::max()) {
            std::fprintf(stderr, "ERROR: local cube of %llu elements needs a 64-bit gpu_size_t\n", (unsigned long long)g.cube_elems);
            std::abort();
        }
        // the kernels mask lanes through 32-bit byte offsets inside a plane
        if (g.plane * 4 >= (uint64_t)0x80000000u) {
            std::fprintf(stderr, "ERROR: planes of %llu elements are beyond the functor kernels' 2^31-byte plane limit\n", (unsigned long long)g.plane);
            std::abort();
        }
        FunctorGeom f;
        f.plane = (long long)g.plane, f.px = (int)g.pitch_x, f.nx = (int)XSize, f.ny = (int)YSize;
        f.zg0 = (long long)g.first_global, f.z_lo = (int)z_lo, f.z_hi = (int)z_hi, f.zchunk = (int)(z_hi - z_lo);
        f.nz_local = (int)(g.cube_elems / g.plane);
        f.gx = f.gy = f.gz = 1;
        return f;
    }

    // planes per workgroup: per_cu workgroups for each of the 256 CUs and no more (a second round would run nearly empty),
    // chunks of at least 12*M planes (every chunk re-reads 2M planes to fill its z window); returns the number of workgroups
    static unsigned chunking(FunctorGeom &g, unsigned gx, unsigned gy, int per_cu)
    {
        const long long nz = (long long)g.z_hi - g.z_lo;
        long long chunks = std::max<long long>(1, 256 * per_cu / ((long long)gx * gy));
        chunks = std::max<long long>(1, std::min(chunks, nz / (long long)std::max<size_t>(12 * M, 16)));
        g.zchunk = (int)((nz + chunks - 1) / chunks);
        g.gx = (int)gx, g.gy = (int)gy, g.gz = (int)((nz + g.zchunk - 1) / g.zchunk);
        return gx * gy * (unsigned)g.gz;
    }

    // the functor overloads stand beside the closed-program methods of the base class
    template <typename F>
    using if_functor = std::enable_if_t<!std::is_convertible<F, std::string>::value && !std::is_same<std::decay_t<F>, fdtdx_program>::value>;

public:
    using Base::Base;
    using Base::filt;
    using Base::inject;
    using Base::propagate;
    using Base::propagateHaloButtomBackup;
    using Base::propagateHaloTopBackup;
    using Base::propagatePlanes;
    ~FunctorStencil()
    {
        for (auto &kv : flat_) (void)hipFree(kv.second.first);
    }

    // ---- named flat buffers and address lists (stencil.h:440-551) -------------------------------------------------------
    // malloc<T>(name, length, is_address): an address list (is_address: indices into the grid, filtered and re-based by
    // transferToGPU) or a plain device array of `length` T.
    template <typename T = DataType>
    T *malloc(const std::string &name, size_t length, bool is_address = false)
    {
        if (is_address) {
            static_assert(sizeof(T) == sizeof(uint64_t) || !std::is_integral<T>::value, "address lists hold 64-bit indices (size_t in main.cu:96)");
            this->mallocAddr(name, length);
            return reinterpret_cast<T *>(this->getAddr(name));
        }
        if (flat_.count(name)) {
            std::fprintf(stderr, "ERROR: buffer '%s' exists\n", name.c_str());  // stencil.h:443
            std::abort();
        }
        void *p = nullptr;
        FDTDX_HIP_ASSERT(hipMalloc(&p, sizeof(T) * std::max<size_t>(length, 1)));
        flat_[name] = {p, sizeof(T) * length};
        return static_cast<T *>(p);
    }
    template <typename T = DataType>
    T *get(const std::string &name)
    {
        auto it = flat_.find(name);
        if (it != flat_.end()) return static_cast<T *>(it->second.first);
        if (sizeof(T) == sizeof(uint64_t) && std::is_integral<T>::value) return reinterpret_cast<T *>(this->getAddr(name));
        return reinterpret_cast<T *>(this->getCube(name));  // stencil.h:479-484: cubes live in the same pool
    }
    // side-buffer twins of an address list (stencil.h:486-498): there are none here
    template <typename T = DataType>
    T *getHaloTop(const std::string &)
    {
        return nullptr;
    }
    template <typename T = DataType>
    T *getHaloButtom(const std::string &)
    {
        return nullptr;
    }
    template <typename T>
    void transferToGPU(const std::string &name, T *cpu, size_t length)
    {
        auto it = flat_.find(name);
        if (it != flat_.end()) {
            if (sizeof(T) * length > it->second.second) {
                std::fprintf(stderr, "ERROR: transferToGPU('%s'): %zu bytes into a buffer of %zu\n", name.c_str(), sizeof(T) * length, it->second.second);
                std::abort();
            }
            FDTDX_HIP_ASSERT(hipMemcpy(it->second.first, cpu, sizeof(T) * length, hipMemcpyHostToDevice));
            return;
        }
        // an address list: global unpadded linear indices, kept where this rank owns the plane and re-based (stencil.h:526-543)
        static_assert(sizeof(T) == sizeof(uint64_t) || !std::is_integral<T>::value, "address lists hold 64-bit indices");
        this->transferAddrToGPU(name, reinterpret_cast<const uint64_t *>(cpu), length);
    }
    template <typename T>
    void transferToCPU(T *cpu, const std::string &name, size_t length)
    {
        auto it = flat_.find(name);
        const void *src = it != flat_.end() ? it->second.first : static_cast<const void *>(this->getAddr(name));
        FDTDX_HIP_ASSERT(hipMemcpy(cpu, src, sizeof(T) * length, hipMemcpyDeviceToHost));
    }
    void free(const std::string &name)
    {
        auto it = flat_.find(name);
        if (it == flat_.end()) return Base::free(name);
        FDTDX_HIP_ASSERT(hipFree(it->second.first));
        flat_.erase(it);
    }

    // ---- the side-buffer replay of the reference's multi-rank loop, functor forms (stencil.h:770-808, 825-841) -----------
    // The reference computes the planes a neighbour needs TWICE per step: first in copies of the slab's first/last 3M
    // planes (propagateHalo*Backup, then inject and filt replayed on those copies, main.cu:153-158), which it sends while
    // the main pass runs, then again in the main pass (main.cu:160-163).  The copies exist because the passes are arbitrary
    // functors that may read what they overwrite (main.cu:118 subtracts the old output), so the early result cannot be
    // written into the cube itself.  Here the cube has no copies: the functor forms of the replay are accepted and do
    // nothing (their address lists are empty, getAddrHalo*Size), the main pass computes every owned plane once, and
    // commCubeHaloBackup sends its boundary planes, stream-ordered behind it, straight into the neighbour's ghost planes.
    // Same fields as the reference's loop; the exchange follows the main pass instead of running beside it.  (Closed
    // programs keep the overlap: the fdtdx_program forms of these methods sweep the boundary planes first.)
    template <typename Function, typename... Arguments, typename = if_functor<Function>>
    void propagateHaloTopBackup(const std::string &, const std::string &, bool, Function, Arguments...)
    {
    }
    template <typename Function, typename... Arguments, typename = if_functor<Function>>
    void propagateHaloButtomBackup(const std::string &, const std::string &, bool, Function, Arguments...)
    {
    }
    template <typename Function, typename... Arguments, typename = if_functor<Function>>
    void filtHaloTopBackup(Function, Arguments...)
    {
    }
    template <typename Function, typename... Arguments, typename = if_functor<Function>>
    void filtHaloButtomBackup(Function, Arguments...)
    {
    }

    // The stream the functor launches go to (the reference's templates take no stream: their launches go to the
    // null stream, which is the default here too).  The closed-program methods of the base class take theirs per call.
    void setStream(hipStream_t s) { stream_ = s; }
    hipStream_t stream() const { return stream_; }

    // stencil.h:684-768 on local planes [z_lo, z_hi); the whole-slab form below is the reference's propagate
    template <typename Function, typename... Arguments, typename = if_functor<Function>>
    void propagatePlanes(const std::string &output, const std::string &input, uint64_t z_lo, uint64_t z_hi, Function kernel,
                         Arguments... args)
    {
        if (z_hi <= z_lo) return;
        DataType *out = this->getCube(output), *in = this->getCube(input);
        FunctorGeom g = geom(z_lo, z_hi);
        if constexpr (FDTD) {
            // 256 cells x 16 rows where rows are long enough, else 128 x 16; one quad per thread (at 512^3 and radius 3:
            // 514 Gcell/s against 503 for 512 x 8 and 465 for 128 x 16)
#if defined(FDTDX_FUNCTOR_TXQ)
            constexpr int TXQ = FDTDX_FUNCTOR_TXQ, LY = FDTDX_FUNCTOR_LY, RY = FDTDX_FUNCTOR_RY;
#else
            constexpr int TXQ = (XSize >= 256) ? 64 : 32, LY = 16, RY = 1;
#endif
            const unsigned gx = (unsigned)(((XSize + 3) / 4 + TXQ - 1) / TXQ), gy = (unsigned)((YSize - 2 * M + LY * RY - 1) / (LY * RY));
            const unsigned wgs = chunking(g, gx, gy, TXQ * LY <= 512 ? 2 : 1);
            hipLaunchKernelGGL((functor_star_kernel<DataType, (int)M, TXQ, LY, RY, gpu_size_t, gpu_signed_size_t, Function, Arguments...>),
                               dim3(wgs), dim3(TXQ * LY), 0, stream_, g, out, in, kernel, args...);
        } else if constexpr (FunctorBoxShape<(int)M, 128, 16>::lds_bytes <= 160 * 1024) {
            constexpr int TX = 128, TY = 16, NT = 512;
            const unsigned gx = (unsigned)((XSize + TX - 1) / TX), gy = (unsigned)((YSize - 2 * M + TY - 1) / TY);
            const unsigned wgs = chunking(g, gx, gy, (int)(160 * 1024 / FunctorBoxShape<(int)M, 128, 16>::lds_bytes));
            hipLaunchKernelGGL((functor_box_lds_kernel<DataType, (int)M, TX, TY, NT, gpu_size_t, gpu_signed_size_t, Function, Arguments...>),
                               dim3(wgs), dim3(NT), 0, stream_, g, out, in, kernel, args...);
        } else {
            const unsigned bx = 256;
            for (uint64_t z = z_lo; z < z_hi; z += 65535) {
                g.z_lo = (int)z;
                hipLaunchKernelGGL((functor_box_kernel<DataType, (int)M, gpu_size_t, gpu_signed_size_t, Function, Arguments...>),
                                   dim3((unsigned)((XSize - 2 * M + bx - 1) / bx), (unsigned)(YSize - 2 * M),
                                        (unsigned)std::min<uint64_t>(65535, z_hi - z)),
                                   dim3(bx), 0, stream_, g, out, in, kernel, args...);
            }
        }
        FDTDX_HIP_ASSERT(hipGetLastError());
    }
    template <typename Function, typename... Arguments, typename = if_functor<Function>>
    void propagate(const std::string &output, const std::string &input, bool spill, Function kernel, Arguments... args)
    {
        (void)spill;
        const fdtdx_geometry &g = this->geometry();
        propagatePlanes(output, input, g.update_lo, g.update_hi, kernel, args...);
    }

    // stencil.h:810-815
    template <typename Function, typename... Arguments, typename = if_functor<Function>>
    void inject(size_t length, Function kernel, Arguments... args)
    {
        if (!length) return;
        hipLaunchKernelGGL((functor_inject_kernel<gpu_size_t, Function, Arguments...>), dim3((unsigned)((length + 255) / 256)), dim3(256), 0, stream_,
                           (gpu_size_t)length, kernel, args...);
        FDTDX_HIP_ASSERT(hipGetLastError());
    }

    // stencil.h:817-823: every cell (shell included) of the planes this rank owns
    template <typename Function, typename... Arguments, typename = if_functor<Function>>
    void filt(Function kernel, Arguments... args)
    {
        const fdtdx_geometry &gg = this->geometry();
        FunctorGeom g = geom(gg.ghost_lo, gg.ghost_lo + gg.owned_planes);
        for (uint64_t z = gg.ghost_lo; z < gg.ghost_lo + gg.owned_planes; z += 65535) {
            g.z_lo = (int)z;
            hipLaunchKernelGGL((functor_filt_kernel<gpu_size_t, Function, Arguments...>),
                               dim3((unsigned)((XSize + 255) / 256), (unsigned)YSize,
                                    (unsigned)std::min<uint64_t>(65535, gg.ghost_lo + gg.owned_planes - z)),
                               dim3(256), 0, stream_, g, kernel, args...);
        }
        FDTDX_HIP_ASSERT(hipGetLastError());
    }
};

}  // namespace fdtdx
