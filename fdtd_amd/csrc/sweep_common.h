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
    ORD_JACOBI_GOLDEN = 3  // jacobi.cu:478-485, separate mul/add
};

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
    int zchunk;       // planes marched by one workgroup (blockIdx.z selects the chunk)
    float c[13];
};

// Description of one compiled tile shape.
struct VariantInfo {
    int M;      // radius
    int order;  // ORD_*
    int tq;     // computed quads (float4) per tile row
    int ly;     // thread rows
    int ry;     // rows per thread
    int threads;
    int lds_bytes;
    void (*launch)(const SweepParams &, dim3 grid, hipStream_t);
};

__device__ __forceinline__ float4 ld4(const float *p) { return *reinterpret_cast<const float4 *>(p); }

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
