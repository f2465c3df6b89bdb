// Times when each workgroup of a star sweep finishes each plane: how far apart do the workgroups that share an L2 run?
// (Measured: neighbours in y stay within one plane of each other, so a halo row misses the L2 only when it is fetched
// many plane steps after the neighbour fetched it -- which is why the kernels park halos in an LDS ring.)
// build (in a directory that travels to the GPU box, not gpurun_out/):
//   hipcc -std=c++17 -O3 --offload-arch=gfx950 -Iinclude tools/drift_probe.hip -o build_tmp/drift_probe
#define FDTDX_DRIFT_PROBE 1
#include "../fdtd_amd/csrc/star_kernel.h"
#include <cstdio>
#include <vector>
#include <algorithm>
#include <cstdlib>
using namespace fdtdx;
namespace fdtdx { void register_variant(const VariantInfo &) {} }
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
template <int M, int TQ, int LY, int RY>
int run(int n)
{
    const long long plane = (long long)n * n, cells = plane * n;
    float *a, *b; unsigned long long *dbg;
    CK(hipMalloc(&a, cells * 4)); CK(hipMalloc(&b, cells * 4));
    std::vector<float> h(cells);
    for (long long i = 0; i < cells; i++) h[i] = (float)((i * 2654435761u) & 0xffff) / 65536.f;
    CK(hipMemcpy(a, h.data(), cells * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(b, h.data(), cells * 4, hipMemcpyHostToDevice));
    SweepParams p{};
    p.plane = plane; p.px = n; p.pxq = n / 4; p.nx = n; p.ny = n; p.h = M; p.z_lo = M; p.z_hi = n - M;
    const int cy = LY * RY;
    p.gx = (n / 4 + TQ - 1) / TQ; p.gy = (n - 2 * M + cy - 1) / cy;
    const int tiles = p.gx * p.gy;
    p.gz = std::max(1, 256 / tiles); p.zchunk = (n - 2 * M + p.gz - 1) / p.gz; p.gz1 = p.gz; p.z_lo2 = p.z_hi2 = 0;
    for (int i = 0; i < 13; i++) p.c[i] = 0.01f * (i + 1);
    p.shell = nullptr; p.shell_plane = 0;
    const int nb = tiles * p.gz;
    CK(hipMalloc(&dbg, (size_t)nb * 256 * 8)); CK(hipMemset(dbg, 0, (size_t)nb * 256 * 8));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    p.dbg = nullptr;
    for (int it = 0; it < 40; it++) {
        p.in = (it & 1) ? b : a; p.out = (it & 1) ? a : b;
        if (it == 20) CK(hipEventRecord(e0));
        if (it == 39) { CK(hipEventRecord(e1)); p.dbg = dbg; }
        hipLaunchKernelGGL((star_sweep_kernel<M, TQ, LY, RY, 0, 1>), dim3(nb), dim3(TQ * LY), 0, 0, p);
    }
    CK(hipDeviceSynchronize());
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("M=%d tile %dx%d ry=%d grid %dx%dx%d zchunk %d: %.1f us per launch\n", M, TQ, LY, RY, p.gx, p.gy, p.gz, p.zchunk, ms * 1e3 / 19);
    std::vector<unsigned long long> t((size_t)nb * 256);
    CK(hipMemcpy(t.data(), dbg, t.size() * 8, hipMemcpyDeviceToHost));
    unsigned long long t0 = ~0ull;
    for (int b2 = 0; b2 < nb; b2++) t0 = std::min(t0, t[(size_t)b2 * 256]);
    // per XCD (hardware block b: XCD b % 8): spread of the time plane k was finished over the XCD's workgroups, in planes' worth of time
    const int ks[] = {0, 10, 30, 60, 90, p.zchunk - 2};
    for (int k : ks) {
        double worst = 0, mean = 0, rate = 0;
        for (int x = 0; x < 8; x++) {
            unsigned long long lo = ~0ull, hi = 0;
            for (int b2 = x; b2 < nb; b2 += 8) { unsigned long long v = t[(size_t)b2 * 256 + k]; lo = std::min(lo, v); hi = std::max(hi, v); }
            worst = std::max(worst, (double)(hi - lo)); mean += (double)(hi - lo) / 8;
        }
        for (int b2 = 0; b2 < nb; b2++) rate += (double)(t[(size_t)b2 * 256 + p.zchunk - 2] - t[(size_t)b2 * 256]) / (p.zchunk - 2) / nb;
        printf("  plane %3d: spread inside an XCD mean %.2f us, worst %.2f us  (one plane takes %.3f us; first finish of plane 0 at +%.2f us)\n",
               k, mean * 0.01, worst * 0.01, rate * 0.01, 0.0);
    }
    // neighbours in y inside an XCD: logical tile lb = slot-run; neighbours in y differ by gx in lb, i.e. by gx in slot
    {
        double s = 0; int cnt = 0;
        for (int b2 = 0; b2 + 8 * p.gx < nb; b2++) {
            int slot = b2 >> 3; if ((slot / p.gx) % p.gy == p.gy - 1) continue;
            for (int k = 20; k < p.zchunk - 2; k += 20) { long long d = (long long)t[(size_t)b2 * 256 + k] - (long long)t[(size_t)(b2 + 8 * p.gx) * 256 + k]; s += llabs(d); cnt++; }
        }
        printf("  |t(tile) - t(neighbour in y)| mean %.2f us\n", s / cnt * 0.01);
    }
    hipFree(a); hipFree(b); hipFree(dbg);
    return 0;
}
int main()
{
    if (run<8, 32, 32, 1>(512)) return 1;
    if (run<4, 128, 8, 1>(512)) return 1;
    if (run<6, 32, 16, 2>(512)) return 1;
    return 0;
}
