// mall_reuse_probe.hip -- does the 256 MB Infinity Cache keep what a sweep has just written?
//
// A time loop ping-pongs between two cubes: step t writes cube A front to back, step t+1 reads A.  If step t+1
// walks A in the SAME direction, the planes it reads first are the ones written longest ago (gone from the
// cache for cubes larger than it); walked in the OPPOSITE direction it starts with the planes written last.
// This probe times a ping-pong copy loop both ways, for several buffer sizes, with plain and non-temporal
// stores, as a yardstick for alternating the march direction of the sweeps between time steps.
//
//   build: hipcc -O3 --offload-arch=gfx950 tools/mall_reuse_probe.hip -o tools/mall_reuse_probe
//   run:   tools/mall_reuse_probe
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

#define CHECK(e)                                                          \
    do {                                                                  \
        hipError_t r_ = (e);                                              \
        if (r_ != hipSuccess) {                                           \
            std::fprintf(stderr, "%s: %s\n", #e, hipGetErrorString(r_)); \
            std::exit(1);                                                 \
        }                                                                 \
    } while (0)

typedef unsigned int u32x4 __attribute__((__vector_size__(16)));

__device__ __forceinline__ __amdgpu_buffer_rsrc_t rsrc(const void *p, unsigned bytes)
{
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p), 0, bytes, 0x00020000);
}

// 256 workgroups of 1024 lanes; the buffer is cut into 256 contiguous chunks (like the z chunks x tiles of a sweep), each
// workgroup walks its chunk in segments of 1024 x UNROLL quads, front to back or back to front
template <int UNROLL, int ST_AUX>
__global__ __launch_bounds__(1024) void chunk_copy(const float *in, float *out, long long nquads, int reverse)
{
    const long long per = 1024LL * UNROLL;
    const long long nseg = (nquads + per - 1) / per;
    const long long segs_per_wg = (nseg + gridDim.x - 1) / gridDim.x;
    const long long s0 = blockIdx.x * segs_per_wg, s1 = min(nseg, s0 + segs_per_wg);
    for (long long i = 0; i < s1 - s0; i++) {
        const long long s = reverse ? s1 - 1 - i : s0 + i;
        const long long q0 = s * per;
        const unsigned bytes = (unsigned)((nquads - q0 < per ? nquads - q0 : per) * 16);
        const __amdgpu_buffer_rsrc_t ri = rsrc(in + q0 * 4, bytes), ro = rsrc(out + q0 * 4, bytes);
        u32x4 v[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; u++) v[u] = __builtin_amdgcn_raw_buffer_load_b128(ri, (threadIdx.x + u * 1024) * 16u, 0, 0);
#pragma unroll
        for (int u = 0; u < UNROLL; u++) __builtin_amdgcn_raw_buffer_store_b128(v[u], ro, (threadIdx.x + u * 1024) * 16u, 0, ST_AUX);
    }
}

template <int ST_AUX>
static void run(float *a, float *b, long long bytes, bool alternate, const char *what)
{
    const long long nquads = bytes / 16;
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    const int reps = 40;
    for (int pass = 0; pass < 2; pass++) {  // first pass warms up
        CHECK(hipEventRecord(e0));
        for (int t = 0; t < reps; t++) {
            const float *in = (t & 1) ? a : b;
            float *out = (t & 1) ? b : a;
            hipLaunchKernelGGL((chunk_copy<4, ST_AUX>), dim3(256), dim3(1024), 0, 0, in, out, nquads, alternate ? (t & 1) : 0);
        }
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
    }
    float ms;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    const double us = ms * 1e3 / reps;
    std::printf("buffer %7.1f MB  store=%s  %-22s: %8.1f us per pass  %5.2f TB/s (read + written)\n", bytes / 1e6, ST_AUX ? "nt   " : "plain", what, us,
                2.0 * bytes / us * 1e-6);
    std::fflush(stdout);
}

int main()
{
    const long long sizes[] = {32LL << 20, 64LL << 20, 100LL << 20, 128LL << 20, 192LL << 20, 256LL << 20, 570425344LL /* a 512^3 cube at pitch 544 */, 2LL << 30};
    float *a, *b;
    CHECK(hipMalloc(&a, 2LL << 30));
    CHECK(hipMalloc(&b, 2LL << 30));
    CHECK(hipMemset(a, 0, 2LL << 30));
    CHECK(hipMemset(b, 0, 2LL << 30));
    // keep the device busy for a while first: its clocks settle only after some tens of milliseconds of work
    for (int t = 0; t < 200; t++) hipLaunchKernelGGL((chunk_copy<4, 2>), dim3(256), dim3(1024), 0, 0, a, b, (2LL << 30) / 16, 0);
    CHECK(hipDeviceSynchronize());
    for (long long s : sizes) {
        run<2>(a, b, s, false, "same direction");
        run<2>(a, b, s, true, "alternating direction");
        run<0>(a, b, s, false, "same direction");
        run<0>(a, b, s, true, "alternating direction");
    }
    return 0;
}
