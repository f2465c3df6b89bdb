// main_loop.hip -- the full-pipeline driver of the reference (main.cu) on fdtdx::FunctorStencil: the same template
// declaration (main.cu:17), the same four device lambdas (main.cu:102-133) and the same multi-rank loop body
// (main.cu:149-171), compiled by hipcc against include/fdtdx/stencil_functor.hpp.
//   build: hipcc -std=c++17 -O3 -fno-slp-vectorize --offload-arch=gfx950 -Iinclude examples/main_loop.hip -Lfdtd_amd -lfdtdx \
//                -Wl,-rpath,$PWD/fdtd_amd -o examples/main_loop
//   run:   examples/main_loop <nslabs> <steps> [dir]
// The grid is cut into <nslabs> z-slabs that all live in this one process (a box with one GPU; one process per slab would
// run the same body with commInit instead of commInitLocal).  With [dir]: the initial fields are read from dir/p0.f32,
// p1.f32, vel.f32 (unpadded [nz][ny][nx]) and dir/addr.u64 (na indices), and the final p0, p1 and sum are written back as
// out_p0.f32, out_p1.f32, out_sum.f32 -- for a test that sets them beside the CPU oracle.  Without: main.cu's own initial
// fields (main.cu:18-47).
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <memory>
#include <random>
#include <string>
#include <vector>

#include "fdtdx/stencil_functor.hpp"

const size_t nz = 44, ny = 36, nx = 150;  // a small grid (main.cu:8-10 uses 512^3); compile-time, as in the reference
const size_t na = 300;
typedef uint32_t gpu_size_t;
typedef int32_t gpu_signed_size_t;
typedef fdtdx::FunctorStencil<float, nz, ny, nx, 4, true, gpu_size_t, gpu_signed_size_t> Slab;  // main.cu:17

template <typename T>
static bool read_file(const std::string &path, std::vector<T> &v)
{
    std::ifstream f(path, std::ios::binary);
    return f && f.read(reinterpret_cast<char *>(v.data()), (std::streamsize)(v.size() * sizeof(T)));
}
template <typename T>
static void write_file(const std::string &path, const std::vector<T> &v)
{
    std::ofstream(path, std::ios::binary).write(reinterpret_cast<const char *>(v.data()), (std::streamsize)(v.size() * sizeof(T)));
}

int main(int argc, char **argv)
{
    const int nslabs = argc > 1 ? atoi(argv[1]) : 1;
    const size_t steps = argc > 2 ? (size_t)atoi(argv[2]) : 10;
    const std::string dir = argc > 3 ? argv[3] : "";
    if (nslabs < 1 || nslabs > 8) return 2;

    std::vector<float> p0(nz * ny * nx), p1(nz * ny * nx), vel(nz * ny * nx), sum(nz * ny * nx, 0.f);
    std::vector<size_t> addr(na);
    if (!dir.empty()) {
        if (!read_file(dir + "/p0.f32", p0) || !read_file(dir + "/p1.f32", p1) || !read_file(dir + "/vel.f32", vel) || !read_file(dir + "/addr.u64", addr)) {
            std::fprintf(stderr, "cannot read the initial fields from %s\n", dir.c_str());
            return 2;
        }
    } else {
        for (size_t z = 0; z < nz; z++) {  // main.cu:21-36: one generator per plane, p0, p1, vel per cell
            std::mt19937_64 g(19950918 + z);
            for (size_t i = z * ny * nx; i < (z + 1) * ny * nx; i++) {
                p0[i] = float(g()) / float(g.max());
                p1[i] = float(g()) / float(g.max());
                vel[i] = float(g()) / float(g.max());
            }
        }
        for (size_t i = 0; i < na; i++) addr[i] = (size_t)rand() % (nz * ny * nx);  // main.cu:37-41
    }

    std::vector<std::unique_ptr<Slab>> slabs;
    std::vector<fdtdx::Stencil<float, nz, ny, nx, 4, true, gpu_size_t, gpu_signed_size_t> *> raw;
    for (int r = 0; r < nslabs; r++) {
        slabs.emplace_back(new Slab((unsigned)r, (unsigned)nslabs, 0));
        raw.push_back(slabs.back().get());
    }
    if (nslabs > 1) Slab::commInitLocal(raw.data(), nslabs);

    for (auto &sp : slabs) {  // main.cu:92-101
        Slab &stencil = *sp;
        stencil.mallocCube("p0", true);
        stencil.mallocCube("p1", true);
        stencil.mallocCube("sum");
        stencil.mallocCube("vel");
        stencil.malloc<size_t>("addr", na, true);
        stencil.transferCubeToGPU("p0", p0.data());
        stencil.transferCubeToGPU("p1", p1.data());
        stencil.transferCubeToGPU("sum", sum.data());
        stencil.transferCubeToGPU("vel", vel.data());
        stencil.transferToGPU("addr", addr.data(), na);
    }
    // the four device lambdas of main.cu:102-133 (same signatures, same arithmetic in the same order)
    auto prop_kernel = [=] __device__(gpu_size_t z, gpu_size_t y, gpu_size_t x, gpu_size_t addr, float *output, float *zl, gpu_signed_size_t sz,
                                      float *yl, gpu_signed_size_t sy, float *xl, gpu_signed_size_t sx, float *vel, float scal) {
        float s = zl[0];
        for (int r = 1; r <= 4; r++) s += 1.0f * (xl[r * sx] + xl[-r * sx]);
        for (int r = 1; r <= 4; r++) s += 2.0f * (yl[r * sy] + yl[-r * sy]);
        for (int r = 1; r <= 4; r++) s += 3.0f * (zl[r * sz] + zl[-r * sz]);
        output[addr] = 0.5f * z - 0.2f * y - 0.3f * x + vel[addr] * scal * s - output[addr];
    };
    auto inject_kernel = [=] __device__(gpu_size_t i, float *output, size_t *addr, float add) { atomicAdd(&output[addr[i]], add); };
    auto filt_kernel = [=] __device__(gpu_size_t z, gpu_size_t y, gpu_size_t x, gpu_size_t addr, float *output) {
        if (z == nz / 2 || y == ny / 3 || x == nx / 4) output[addr] *= 0.9f;
    };
    auto mul_kernel = [=] __device__(gpu_size_t z, gpu_size_t y, gpu_size_t x, gpu_size_t addr, float *sum, float *p0, float *p1) {
        if (z >= 4 && y >= 4 && x >= 4 && z < nz - 4 && y < ny - 4 && x < nx - 4) sum[addr] += p0[addr] * p1[addr];
    };
    if (nslabs > 1)
        for (auto &sp : slabs) sp->barrier();
    auto start = std::chrono::steady_clock::now();
    std::string s0 = "p0", s1 = "p1";
    for (size_t t = 0; t < steps; t++) {
        for (auto &sp : slabs) {  // main.cu:151-163, per slab
            Slab &stencil = *sp;
            stencil.backupCubeHaloBackup(s0);
            stencil.backupCubeHaloBackup(s1);
            stencil.propagateHaloTopBackup(s0, s1, true, prop_kernel, stencil.getCubeHaloTopBackup("vel"), 1.0 / (6 * 8 + 1));
            stencil.propagateHaloButtomBackup(s0, s1, true, prop_kernel, stencil.getCubeHaloButtomBackup("vel"), 1.0 / (6 * 8 + 1));
            stencil.inject(stencil.getAddrHaloTopSize("addr"), inject_kernel, stencil.getCubeHaloTopBackup(s0), stencil.getHaloTop<size_t>("addr"), 1.0);
            stencil.inject(stencil.getAddrHaloButtomSize("addr"), inject_kernel, stencil.getCubeHaloButtomBackup(s0), stencil.getHaloButtom<size_t>("addr"), 1.0);
            stencil.filtHaloTopBackup(filt_kernel, stencil.getCubeHaloTopBackup(s0));
            stencil.filtHaloButtomBackup(filt_kernel, stencil.getCubeHaloButtomBackup(s0));
            stencil.sync();
            stencil.propagate(s0, s1, true, prop_kernel, stencil.getCube("vel"), 1.0 / (6 * 8 + 1));
            stencil.inject(stencil.getAddrSize("addr"), inject_kernel, stencil.getCube(s0), stencil.get<size_t>("addr"), 1.0);
            stencil.filt(filt_kernel, stencil.getCube(s0));
            stencil.filt(mul_kernel, stencil.getCube("sum"), stencil.getCube(s0), stencil.getCube(s1));
        }
        // main.cu:164-165: commCubeHaloBackup(s0); commCubeHaloBackup(s1).  Slabs of one process exchange in two phases
        // (start on every slab, then wait on every slab); one process per slab calls commCubeHaloBackup as the reference does.
        if (nslabs > 1)
            for (const std::string &name : {s0, s1}) {
                for (auto &sp : slabs) sp->exchangeStart(name);
                for (auto &sp : slabs) sp->exchangeWait();
            }
        for (auto &sp : slabs) {  // main.cu:166-169
            Slab &stencil = *sp;
            stencil.sync();
            stencil.restoreCubeHaloBackup(s0);
            stencil.restoreCubeHaloBackup(s1);
            stencil.sync();
        }
        std::swap(s0, s1);
    }
    std::printf("GPU time %.6lfs\n", std::chrono::duration<double>(std::chrono::steady_clock::now() - start).count());
    // main.cu:177-179: every rank ends with the whole grid (one call gathers the slabs of this process)
    slabs[0]->transferCubeToCPU(p0.data(), s0);
    slabs[0]->transferCubeToCPU(p1.data(), s1);
    slabs[0]->transferCubeToCPU(sum.data(), "sum");
    if (!dir.empty()) {
        write_file(dir + "/out_p0.f32", p0);
        write_file(dir + "/out_p1.f32", p1);
        write_file(dir + "/out_sum.f32", sum);
    }
    double chk = 0.0;
    for (float v : p0) chk += v;
    std::printf("checksum p0 %.6f\n", chk);
    return 0;
}
