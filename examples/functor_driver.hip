// functor_driver.hip -- user-written device lambdas through fdtdx/stencil_functor.hpp, checked against
// the closed programs of the C ABI on the same inputs.
//   build: hipcc -std=c++17 -O3 -fno-slp-vectorize --offload-arch=gfx950 -Iinclude examples/functor_driver.hip -Lfdtd_amd -lfdtdx
//          -Wl,-rpath,$PWD/fdtd_amd -o examples/functor_driver        (fdtd_amd/csrc/Makefile does it)
//   run:   ./functor_driver        -> one line per case, exit status 0 when every case agrees
//          ./functor_driver bench  -> throughput of a radius-3 lambda at 512^3 beside the closed program's
// The lambdas have the signature the reference's drivers use (fdtd3.cu:66-73, fdtdspc.cu:64-73,
// main.cu:102-133); grids are small so that the run takes a moment.
#include <cmath>
#include <cstdio>
#include <cstring>
#include <vector>

#include "fdtdx/stencil_functor.hpp"

constexpr size_t nz = 37, ny = 45, nx = 150;
typedef uint32_t gpu_size_t;
typedef int32_t gpu_signed_size_t;

// Agreement up to the rounding of a different FMA contraction: the compiler is free to fuse the
// lambda's products differently from the closed program's stated order, which moves results by an
// ulp of the terms involved; where terms cancel (main.cu subtracts the old field) the fair yardstick
// is the size of the terms, taken here as the largest magnitude in the field.
static int compare(const char *what, const std::vector<float> &a, const std::vector<float> &b)
{
    double top = 0, worst = 0;
    size_t diff = 0;
    for (size_t i = 0; i < b.size(); i++) top = std::fmax(top, std::fabs((double)b[i]));
    for (size_t i = 0; i < a.size(); i++) {
        if (std::memcmp(&a[i], &b[i], 4) != 0) diff++;
        const double d = std::fabs((double)a[i] - (double)b[i]) / std::fmax(std::fabs((double)b[i]), 0.01 * top);
        worst = std::fmax(worst, d);
    }
    std::printf("%-28s cells differing in bits: %zu of %zu, max difference relative to max(|value|, 1%% of field maximum) %.3g\n", what,
                diff, a.size(), worst);
    return worst <= 1e-5 ? 0 : 1;
}

template <class S>
static std::vector<float> fetch(S &st, const char *name)
{
    std::vector<float> h(nz * ny * nx);
    st.transferCubeToCPU(h.data(), name);
    return h;
}

// 512^3, radius 3, the lambda of fdtd3.cu:69-72 through the functor path and the same update as a closed program
static int bench()
{
    constexpr size_t n = 512;
    fdtdx::FunctorStencil<float, n, n, n, 3, true, gpu_size_t, gpu_signed_size_t> st;
    for (const char *c : {"p0", "p1"}) st.mallocCube(c, true);
    st.initMt("p0", "p1");
    auto star3 = [=] __device__(gpu_size_t, gpu_size_t, gpu_size_t, gpu_size_t addr, float *output, float *zl, gpu_signed_size_t sz,
                                float *yl, gpu_signed_size_t sy, float *xl, gpu_signed_size_t sx) {
        float acc = 0.01f * zl[0];
        for (int r = 1; r <= 3; r++) {
            acc += 0.02f * xl[r * sx];
            acc += 0.03f * xl[-r * sx];
            acc += 0.04f * yl[r * sy];
            acc += 0.05f * yl[-r * sy];
            acc += 0.06f * zl[r * sz];
            acc += 0.07f * zl[-r * sz];
        }
        output[addr] = acc;
    };
    const fdtdx_program prog = fdtdx::star_program(3);
    const int steps = 100;
    const double cells = double(n - 6) * double(n - 6) * double(n - 6);
    hipEvent_t e0, e1;
    FDTDX_HIP_ASSERT(hipEventCreate(&e0));
    FDTDX_HIP_ASSERT(hipEventCreate(&e1));
    for (int which = 0; which < 2; which++) {
        auto step = [&](int t) {
            const char *out = (t % 2 == 0) ? "p0" : "p1", *in = (t % 2 == 0) ? "p1" : "p0";
            if (which == 0)
                st.propagate(out, in, true, star3);
            else
                st.propagate(out, in, true, prog);
        };
        for (int t = 0; t < 300; t++) step(t);  // the device clock settles over the first tens of milliseconds
        FDTDX_HIP_ASSERT(hipEventRecord(e0, nullptr));
        for (int t = 0; t < steps; t++) step(t);
        FDTDX_HIP_ASSERT(hipEventRecord(e1, nullptr));
        FDTDX_HIP_ASSERT(hipEventSynchronize(e1));
        float ms = 0;
        FDTDX_HIP_ASSERT(hipEventElapsedTime(&ms, e0, e1));
        std::printf("%-34s %7.1f us per step  %7.1f Gcell/s\n", which == 0 ? "radius-3 lambda (functor path)" : "radius-3 closed program (C ABI)",
                    ms * 1e3 / steps, cells * steps / (ms * 1e-3) * 1e-9);
    }
    // the 13-point lambda of fdtdspc.cu:64-73 in box mode beside the closed program
    fdtdx::FunctorStencil<float, n, n, n, 1, false, gpu_size_t, gpu_signed_size_t> sb;
    for (const char *c : {"p0", "p1"}) sb.mallocCube(c, true);
    sb.initMt("p0", "p1");
    auto spc = [=] __device__(gpu_size_t, gpu_size_t, gpu_size_t, gpu_size_t addr, float *output, float *zl, gpu_signed_size_t sz, float *yl,
                              gpu_signed_size_t sy, float *xl, gpu_signed_size_t sx) {
        float acc = 0.01f * zl[0];
        acc += 0.02f * xl[sx];
        acc += 0.03f * xl[-sx];
        acc += 0.04f * yl[sy];
        acc += 0.05f * yl[-sy];
        acc += 0.06f * zl[sz];
        acc += 0.07f * zl[-sz];
        acc += 0.02f * xl[sx + sy];
        acc += 0.03f * xl[-sx - sy];
        acc += 0.04f * yl[sy + sz];
        acc += 0.05f * yl[-sy - sz];
        acc += 0.06f * zl[sz + sy + sx];
        acc += 0.07f * zl[-sz - sy - sx];
        output[addr] = acc;
    };
    const fdtdx_program sprog = fdtdx::spc13_program();
    const double scells = double(n - 2) * double(n - 2) * double(n - 2);
    for (int which = 0; which < 2; which++) {
        auto step = [&](int t) {
            const char *out = (t % 2 == 0) ? "p0" : "p1", *in = (t % 2 == 0) ? "p1" : "p0";
            if (which == 0)
                sb.propagate(out, in, true, spc);
            else
                sb.propagate(out, in, true, sprog);
        };
        for (int t = 0; t < 100; t++) step(t);
        FDTDX_HIP_ASSERT(hipEventRecord(e0, nullptr));
        for (int t = 0; t < steps; t++) step(t);
        FDTDX_HIP_ASSERT(hipEventRecord(e1, nullptr));
        FDTDX_HIP_ASSERT(hipEventSynchronize(e1));
        float ms = 0;
        FDTDX_HIP_ASSERT(hipEventElapsedTime(&ms, e0, e1));
        std::printf("%-34s %7.1f us per step  %7.1f Gcell/s\n", which == 0 ? "13-point lambda (box mode)" : "13-point closed program (C ABI)",
                    ms * 1e3 / steps, scells * steps / (ms * 1e-3) * 1e-9);
    }
    return 0;
}

// with "dump <dir>": what the lambdas computed, as raw fp32 [z][y][x] files, for a test to set beside an independent CPU computation
static const char *g_dump = nullptr;
static void dump(const char *name, const std::vector<float> &v)
{
    if (!g_dump) return;
    char path[1024];
    std::snprintf(path, sizeof(path), "%s/%s.f32", g_dump, name);
    FILE *f = std::fopen(path, "wb");
    if (!f || std::fwrite(v.data(), 4, v.size(), f) != v.size()) {
        std::fprintf(stderr, "cannot write %s\n", path);
        std::exit(1);
    }
    std::fclose(f);
}

int main(int argc, char **argv)
{
    if (argc > 1 && std::strcmp(argv[1], "bench") == 0) return bench();
    if (argc > 2 && std::strcmp(argv[1], "dump") == 0) g_dump = argv[2];
    int bad = 0;
    {   // radius-3 star in star mode: registers along z, LDS along y and x
        fdtdx::FunctorStencil<float, nz, ny, nx, 3, true, gpu_size_t, gpu_signed_size_t> st;
        for (const char *n : {"p0", "p1", "q0"}) st.mallocCube(n, true);
        st.initMt("p0", "p1");
        st.initMt("q0", "p1");
        auto star3 = [=] __device__(gpu_size_t, gpu_size_t, gpu_size_t, gpu_size_t addr, float *output, float *zl, gpu_signed_size_t sz,
                                    float *yl, gpu_signed_size_t sy, float *xl, gpu_signed_size_t sx) {
            float acc = 0.01f * zl[0];
            for (int r = 1; r <= 3; r++) {
                acc += 0.02f * xl[r * sx];
                acc += 0.03f * xl[-r * sx];
                acc += 0.04f * yl[r * sy];
                acc += 0.05f * yl[-r * sy];
                acc += 0.06f * zl[r * sz];
                acc += 0.07f * zl[-r * sz];
            }
            output[addr] = acc;
        };
        st.propagate("p0", "p1", true, star3);
        st.propagate("q0", "p1", true, fdtdx::star_program(3));
        st.sync();
        bad += compare("star radius 3 (star mode)", fetch(st, "p0"), fetch(st, "q0"));
        dump("star3", fetch(st, "p0"));
    }
    {   // 13-point stencil in box mode: mixed offsets
        fdtdx::FunctorStencil<float, nz, ny, nx, 1, false, gpu_size_t, gpu_signed_size_t> st;
        for (const char *n : {"p0", "p1", "q0"}) st.mallocCube(n, true);
        st.initMt("p0", "p1");
        st.initMt("q0", "p1");
        auto spc = [=] __device__(gpu_size_t, gpu_size_t, gpu_size_t, gpu_size_t addr, float *output, float *zl, gpu_signed_size_t sz,
                                  float *yl, gpu_signed_size_t sy, float *xl, gpu_signed_size_t sx) {
            float acc = 0.01f * zl[0];
            acc += 0.02f * xl[sx];
            acc += 0.03f * xl[-sx];
            acc += 0.04f * yl[sy];
            acc += 0.05f * yl[-sy];
            acc += 0.06f * zl[sz];
            acc += 0.07f * zl[-sz];
            acc += 0.02f * xl[sx + sy];
            acc += 0.03f * xl[-sx - sy];
            acc += 0.04f * yl[sy + sz];
            acc += 0.05f * yl[-sy - sz];
            acc += 0.06f * zl[sz + sy + sx];
            acc += 0.07f * zl[-sz - sy - sx];
            output[addr] = acc;
        };
        st.propagate("p0", "p1", true, spc);
        st.propagate("q0", "p1", true, fdtdx::spc13_program());
        st.sync();
        bad += compare("13-point (box mode)", fetch(st, "p0"), fetch(st, "q0"));
        dump("spc13", fetch(st, "p0"));
    }
    {   // main.cu's pipeline with lambdas: update with extra arguments, inject, filt, mul
        fdtdx::FunctorStencil<float, nz, ny, nx, 4, true, gpu_size_t, gpu_signed_size_t> st;
        for (const char *n : {"p0", "p1", "vel", "sum", "q0", "qsum"}) st.mallocCube(n, true);
        st.initMt("p0", "p1");
        st.initMt("q0", "p1");
        st.initMt("vel", "sum");
        FDTDX_HIP_ASSERT(hipMemset(st.getCube("sum"), 0, st.geometry().cube_elems * 4));
        FDTDX_HIP_ASSERT(hipMemset(st.getCube("qsum"), 0, st.geometry().cube_elems * 4));
        const size_t na = 500;
        std::vector<uint64_t> addr(na);
        for (size_t i = 0; i < na; i++) addr[i] = (i * 2654435761ull) % (nz * ny * nx);
        st.mallocAddr("addr", na);
        st.transferAddrToGPU("addr", addr.data(), na);
        // the same addresses in the local padded layout, for the lambda version
        std::vector<uint64_t> padded(na);
        const fdtdx_geometry &g = st.geometry();
        for (size_t i = 0; i < na; i++) {
            const uint64_t z = addr[i] / (ny * nx), y = (addr[i] % (ny * nx)) / nx, x = addr[i] % nx;
            padded[i] = z * g.plane + y * g.pitch_x + x;
        }
        uint64_t *dev_addr = nullptr;
        FDTDX_HIP_ASSERT(hipMalloc((void **)&dev_addr, na * 8));
        FDTDX_HIP_ASSERT(hipMemcpy(dev_addr, padded.data(), na * 8, hipMemcpyHostToDevice));
        auto update = [=] __device__(gpu_size_t z, gpu_size_t y, gpu_size_t x, gpu_size_t a, float *output, float *zl, gpu_signed_size_t sz,
                                     float *yl, gpu_signed_size_t sy, float *xl, gpu_signed_size_t sx, float *vel, float scal) {
            float s = zl[0];
            for (int r = 1; r <= 4; r++) s += 1.0f * (xl[r * sx] + xl[-r * sx]);
            for (int r = 1; r <= 4; r++) s += 2.0f * (yl[r * sy] + yl[-r * sy]);
            for (int r = 1; r <= 4; r++) s += 3.0f * (zl[r * sz] + zl[-r * sz]);
            output[a] = 0.5f * z - 0.2f * y - 0.3f * x + vel[a] * scal * s - output[a];
        };
        auto add_one = [=] __device__(gpu_size_t i, float *output, uint64_t *where, float add) { atomicAdd(&output[where[i]], add); };
        auto damp = [=] __device__(gpu_size_t z, gpu_size_t y, gpu_size_t x, gpu_size_t a, float *output) {
            if (z == nz / 2 || y == ny / 3 || x == nx / 4) output[a] *= 0.9f;
        };
        auto mul = [=] __device__(gpu_size_t z, gpu_size_t y, gpu_size_t x, gpu_size_t a, float *sum, float *p0, float *p1) {
            if (z >= 4 && y >= 4 && x >= 4 && z < nz - 4 && y < ny - 4 && x < nx - 4) sum[a] += p0[a] * p1[a];
        };
        st.propagate("p0", "p1", true, update, st.getCube("vel"), (float)(1.0 / (6 * 8 + 1)));
        st.inject(na, add_one, st.getCube("p0"), dev_addr, 1.0f);
        st.filt(damp, st.getCube("p0"));
        st.filt(mul, st.getCube("sum"), st.getCube("p0"), st.getCube("p1"));
        const fdtdx_composite comp = fdtdx::main_composite();
        st.propagateComposite("q0", "p1", "vel", comp);
        st.inject("q0", "addr", 1.0f);
        st.filt("q0", (int64_t)(nz / 2), (int64_t)(ny / 3), (int64_t)(nx / 4), 0.9);
        st.mulAcc("qsum", "q0", "p1", 4);
        st.sync();
        bad += compare("main.cu update+inject+filt", fetch(st, "p0"), fetch(st, "q0"));
        bad += compare("main.cu sum", fetch(st, "sum"), fetch(st, "qsum"));
        dump("main_p0", fetch(st, "p0"));
        dump("main_sum", fetch(st, "sum"));
        FDTDX_HIP_ASSERT(hipFree(dev_addr));
    }
    std::printf(bad ? "FAILED\n" : "all cases agree\n");
    return bad;
}
