// fdtd_driver.cpp -- the flow of the reference drivers (fdtd1.cu:15-100) on top of fdtdx/stencil.hpp.
//   build: g++ -std=c++17 -O2 -Iinclude examples/fdtd_driver.cpp -Lfdtd_amd -lfdtdx -Wl,-rpath,$PWD/fdtd_amd -o fdtd_driver
//   run:   ./fdtd_driver [radius 1..8 | spc | jacobi] [steps]
// Grid sizes are compile-time constants like in the reference (fdtd1.cu:8-10).
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "fdtdx/stencil.hpp"

constexpr size_t nz = 512, ny = 512, nx = 512;

static uint32_t crc32_of(const void *data, size_t n)
{
    static uint32_t table[256];
    if (!table[1])
        for (uint32_t i = 0; i < 256; i++) {
            uint32_t c = i;
            for (int k = 0; k < 8; k++) c = (c & 1) ? 0xEDB88320u ^ (c >> 1) : c >> 1;
            table[i] = c;
        }
    uint32_t c = 0xFFFFFFFFu;
    const unsigned char *p = (const unsigned char *)data;
    for (size_t i = 0; i < n; i++) c = table[(c ^ p[i]) & 0xFF] ^ (c >> 8);
    return c ^ 0xFFFFFFFFu;
}

template <size_t M>
static int run(const fdtdx_program &prog, unsigned steps, bool jacobi)
{
    using namespace fdtdx;
    typedef uint32_t gpu_size_t;
    typedef int32_t gpu_signed_size_t;
    Stencil<float, nz, ny, nx, M, true, gpu_size_t, gpu_signed_size_t> stencil;  // the declaration of fdtd1.cu:17 as it stands
    stencil.mallocCube("p0", true);
    stencil.mallocCube("p1", true);
    if (jacobi) {
        stencil.initJacobi("p0");
        stencil.initJacobi("p1");
    } else {
        stencil.initMt("p0", "p1");
    }
    stencil.sync();
    auto t0 = std::chrono::steady_clock::now();
    stencil.sweep("p0", "p1", prog, steps);
    stencil.sync();
    double s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    std::printf("GPU time %.6lfs\n", s);  // fdtd1.cu:82
    double cells = double(nz - 2 * prog.radius) * double(ny - 2 * prog.radius) * double(nx - 2 * prog.radius) * steps;
    std::printf("%.1f Gcell/s\n", cells / s * 1e-9);
    std::vector<float> p0(nz * ny * nx), p1(nz * ny * nx);
    stencil.transferCubeToCPU(p0.data(), "p0");
    stencil.transferCubeToCPU(p1.data(), "p1");
    double chk = 0;
    for (size_t i = 0; i < p0.size(); i += 4097) chk += p0[i] + p1[i];
    std::printf("checksum %.9g\n", chk);
    // CRC-32 (the zlib polynomial) over every byte of both cubes, so that a test can compare the run
    // with an independent computation of the same loop bit for bit
    std::printf("crc32 %08x %08x\n", crc32_of(p0.data(), p0.size() * sizeof(float)), crc32_of(p1.data(), p1.size() * sizeof(float)));
    return 0;
}

int main(int argc, char **argv)
{
    std::string v = argc > 1 ? argv[1] : "1";
    unsigned steps = argc > 2 ? (unsigned)atoi(argv[2]) : 1000;
    if (v == "spc") return run<1>(fdtdx::spc13_program(), steps, false);
    if (v == "jacobi") return run<1>(fdtdx::jacobi7_program(), steps, true);
    switch (atoi(v.c_str())) {
    case 1: return run<1>(fdtdx::star_program(1), steps, false);
    case 2: return run<2>(fdtdx::star_program(2), steps, false);
    case 3: return run<3>(fdtdx::star_program(3), steps, false);
    case 4: return run<4>(fdtdx::star_program(4), steps, false);
    case 5: return run<5>(fdtdx::star_program(5), steps, false);
    case 6: return run<6>(fdtdx::star_program(6), steps, false);
    case 7: return run<7>(fdtdx::star_program(7), steps, false);
    case 8: return run<8>(fdtdx::star_program(8), steps, false);
    }
    std::fprintf(stderr, "usage: %s [1..8|spc|jacobi] [steps]\n", argv[0]);
    return 2;
}
