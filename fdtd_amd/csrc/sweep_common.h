// sweep_common.h -- shared declarations between the HIP sweep kernels and the host ABI.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace fdtdx {

// Arithmetic orders implemented by the kernels (see fdtdx.h: FDTDX_ORDER_*).
enum : int {
    ORD_STAR_CHAIN = 0,    // fdtdN.cu device lambda with FMA contraction
    ORD_STAR_GOLDEN = 1,   // fdtdN.cu host golden loop, separate mul/add
    ORD_JACOBI_CHAIN = 2,  // jacobi.cu:31-38 with FMA contraction
    ORD_JACOBI_GOLDEN = 3, // jacobi.cu:478-485, separate mul/add
    ORD_COMP_CHAIN = 4,    // main.cu:104-118 device lambda: float constants, FMA contraction
    ORD_COMP_GOLDEN = 5,   // main.cu:55-72 host loop: double constants, evaluated in double, rounded once
    ORD_SPC_CHAIN = 6,     // fdtdspc.cu:66-72 device lambda with FMA contraction (fused kernels' registry only)
    ORD_SPC_GOLDEN = 7     // fdtdspc.cu:46-52 host loop
};
constexpr bool is_composite(int order) { return order == ORD_COMP_CHAIN || order == ORD_COMP_GOLDEN; }

// Everything one sweep launch needs.  Passed by value (kernarg segment).
struct SweepParams {
    const float *in;
    float *out;
    long long plane;  // floats between consecutive z planes
    int px;           // floats between consecutive rows (multiple of 4)
    int pxq;          // px / 4
    int nx, ny;       // logical extents of a plane
    int h;            // shell width in x and y that is never written (= radius)
    int z_lo, z_hi;   // local planes [z_lo, z_hi) to update
    int zchunk;       // planes marched by one workgroup
    int gx, gy, gz;   // logical grid of tiles: x tiles, y tiles, z chunks (launched as gx*gy*gz blocks in 1-D)
    int gz1, z_lo2, z_hi2;  // z chunks gz1 .. gz-1 belong to a second plane range [z_lo2, z_hi2) (gz1 == gz: there is none)
    float c[13];
    // Shell components of the quads that straddle the x shell, kept beside the output cube for the time loops: a strip of
    // two quads per row (the row's left and right such quad) and `shell_plane` floats per plane, holding what the output
    // cube holds there (its shell never changes).  nullptr: they are read from the output cube itself.
    const float *shell;
    long long shell_plane;
    // composite update only (main.cu): second input cube, global z of local plane 0, and the
    // constants {cz, cy, cx, snum, sden, wx, wy, wz} as doubles
    const float *vel;
    int zg0;
    double kd[8];
    float *sum;   // fused product-accumulate target (nullptr: none)
    int sel[3];   // global z, y, x of the planes filt scales afterwards (-1: none)
#ifdef FDTDX_DRIFT_PROBE
    unsigned long long *dbg;
#endif
};

// Description of one compiled tile shape.
struct VariantInfo {
    int M;      // radius
    int order;  // ORD_*
    int tq;     // computed quads (float4) per tile row
    int ly;     // thread rows
    int ry;     // rows per thread
    int pf;     // planes prefetched ahead of the queue
    int threads;
    int lds_bytes;
    void (*launch)(const SweepParams &, dim3 grid, hipStream_t);
};

__device__ __forceinline__ float4 ld4(const float *p) { return *reinterpret_cast<const float4 *>(p); }

// Raw buffer access: a 128-bit descriptor (base, byte count) in scalar registers plus a 32-bit
// per-lane byte offset.  An offset at or beyond the byte count is out of range: the load returns
// zeros and the store is dropped, which is how lanes are masked without branches.
// the builtins traffic in GCC-style vectors; whole-vector bit casts only (a __builtin_bit_cast of
// a single vector ELEMENT reads element 0 whichever index is written)
#ifndef FDTDX_STORE_AUX
#define FDTDX_STORE_AUX 2
#endif
typedef unsigned int u32x4 __attribute__((__vector_size__(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((__vector_size__(16)));
constexpr unsigned OOB = 0x80000000u;  // beyond any plane (planes are limited to 2^31-1 bytes)

__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void *p, unsigned bytes)
{
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p), 0, bytes, 0x00020000);
}
__device__ __forceinline__ float4 buf_ld4(__amdgpu_buffer_rsrc_t r, unsigned voff)
{
    const f32x4 f = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, voff, 0, 0));
    return make_float4(f[0], f[1], f[2], f[3]);
}
// the same load for data read once per sweep (second inputs, old output): streamed past the caches
#ifndef FDTDX_STREAM_AUX
#define FDTDX_STREAM_AUX 2
#endif
__device__ __forceinline__ float4 buf_ld4s(__amdgpu_buffer_rsrc_t r, unsigned voff)
{
    const f32x4 f = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, voff, 0, FDTDX_STREAM_AUX));
    return make_float4(f[0], f[1], f[2], f[3]);
}
__device__ __forceinline__ void buf_st4(__amdgpu_buffer_rsrc_t r, unsigned voff, float a, float b, float c, float d)
{
    f32x4 f;
    f[0] = a;
    f[1] = b;
    f[2] = c;
    f[3] = d;
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, f), r, voff, 0, FDTDX_STORE_AUX);
}

// one cell (partially stored quads: the cells of a quad that are not shell cells)
__device__ __forceinline__ void buf_st1(__amdgpu_buffer_rsrc_t r, unsigned voff, float a)
{
    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, a), r, voff, 0, FDTDX_STORE_AUX);
}

// Workgroups are dealt round-robin over the 8 XCDs (each with a private L2): hardware block b and
// b+8 share an L2.  Remap so that each XCD owns a contiguous run of logical tiles -- tiles that
// are neighbours in y (and x) then fetch each other's halo rows from the same L2.  Bijective for
// any block count.  Placement only affects speed, never results.
__device__ __forceinline__ void logical_tile(const SweepParams &p, int &bx, int &by, int &bz)
{
    const int nb = p.gx * p.gy * p.gz;
    const int b = blockIdx.x;
    const int xcd = b & 7, slot = b >> 3;
    const int qn = nb >> 3, rn = nb & 7;
    const int lb = (xcd < rn ? xcd * (qn + 1) : rn * (qn + 1) + (xcd - rn) * qn) + slot;
    bx = lb % p.gx;
    const int t = lb / p.gx;
    by = t % p.gy;
    bz = t / p.gy;
}

// The planes [zs, ze) of z chunk bz, and the end of the plane range it belongs to.  A launch covers one plane range or
// two (the planes next to either neighbour of a slab, swept ahead of the interior in one launch).
template <class P>
__device__ __forceinline__ void chunk_planes(const P &p, int bz, int &zs, int &ze, int &range_end)
{
    const bool second = bz >= p.gz1;
    const int lo = second ? p.z_lo2 : p.z_lo;
    range_end = second ? p.z_hi2 : p.z_hi;
    zs = lo + (second ? bz - p.gz1 : bz) * p.zchunk;
    ze = min(zs + p.zchunk, range_end);
}

// Compile-time loop: f(PhaseTag<PH>{}) for PH in [PH, N), used to unroll z loops whose register
// queues rotate by renaming.
template <int I>
struct PhaseTag {
    static constexpr int value = I;
};
template <int PH, int N, class F>
__device__ __forceinline__ void unroll_phases(F &&f)
{
    if constexpr (PH < N) {
        f(PhaseTag<PH>{});
        unroll_phases<PH + 1, N>(f);
    }
}

// Registry, filled by the per-radius translation units.
int num_variants();
const VariantInfo *variant(int i);
void register_variant(const VariantInfo &v);

// spc-13 (fdtdspc.cu) variants live in their own registry: order is FDTDX_ORDER_*.
struct SpcVariantInfo {
    int order;
    int tq, ly, ry, threads, lds_bytes;
    void (*launch)(const SweepParams &, dim3 grid, hipStream_t);
};
int num_spc_variants();
const SpcVariantInfo *spc_variant(int i);
void register_spc_variant(const SpcVariantInfo &v);

}  // namespace fdtdx
