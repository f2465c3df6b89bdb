// fdtd1_loop.cpp -- the multi-rank time loop of the reference's drivers (fdtd1.cu:68-80) on fdtdx::Stencil: the same calls in
// the same order, with the device lambda replaced by a program descriptor and MPI by an RCCL communicator whose id the
// launcher distributes.
//   build: g++ -std=c++17 -O2 -Iinclude examples/fdtd1_loop.cpp -Lfdtd_amd -lfdtdx -Wl,-rpath,$PWD/fdtd_amd -o examples/fdtd1_loop
//   run, one process per slab (what mpirun does for the reference; every rank on its own GPU):
//          ./fdtd1_loop spmd <rank> <nprocs> <id file> [radius] [steps]     rank 0 writes the id file, the others wait for it
//   run, every slab in this one process (a box with one GPU):
//          ./fdtd1_loop local <nslabs> [radius] [steps]
// Prints the CRC-32 of cube p0 as this process holds it afterwards (the whole grid with one process), for a test to compare.
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <memory>
#include <string>
#include <thread>
#include <vector>

#include "fdtdx/stencil.hpp"

constexpr size_t nz = 192, ny = 160, nx = 256;  // compile-time sizes, as in the reference (fdtd1.cu:8-10)
constexpr size_t GHOST = 8;                      // ghost depth of the slabs: enough for every radius

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

typedef fdtdx::Stencil<float, nz, ny, nx, GHOST> Slab;

static fdtdx_comm_id share_id(unsigned rank, const std::string &path)
{
    fdtdx_comm_id id;
    if (rank == 0) {
        id = Slab::commUniqueId();
        std::ofstream(path + ".tmp", std::ios::binary).write(id.bytes, sizeof(id.bytes));
        std::rename((path + ".tmp").c_str(), path.c_str());
    } else {
        for (;;) {
            std::ifstream f(path, std::ios::binary);
            if (f && f.read(id.bytes, sizeof(id.bytes))) break;
            std::this_thread::sleep_for(std::chrono::milliseconds(20));
        }
    }
    return id;
}

int main(int argc, char **argv)
{
    const std::string mode = argc > 1 ? argv[1] : "local";
    std::vector<float> host_p0(nz * ny * nx, 0.f);
    if (mode == "spmd" && argc >= 5) {
        const unsigned rank = (unsigned)atoi(argv[2]), nprocs = (unsigned)atoi(argv[3]);
        const fdtdx_program prop_kernel = fdtdx::star_program(argc > 5 ? (unsigned)atoi(argv[5]) : 1);
        const size_t steps = argc > 6 ? (size_t)atoi(argv[6]) : 1000;
        int ndev = 1;
        (void)ndev;
        Slab stencil(rank, nprocs, -1);
        if (nprocs > 1) stencil.commInit(share_id(rank, argv[4]));
        stencil.mallocCube("p0", true);
        stencil.mallocCube("p1", true);
        stencil.initMt("p0", "p1");
        stencil.barrier();
        auto start = std::chrono::steady_clock::now();
        std::string s0 = "p0", s1 = "p1";
        for (size_t t = 0; t < steps; t++) {  // fdtd1.cu:68-80
            stencil.backupCubeHaloBackup(s0);
            stencil.propagateHaloTopBackup(s0, s1, true, prop_kernel);
            stencil.propagateHaloButtomBackup(s0, s1, true, prop_kernel);
            stencil.sync();
            stencil.propagate(s0, s1, true, prop_kernel);
            stencil.commCubeHaloBackup(s0);
            stencil.sync();
            stencil.restoreCubeHaloBackup(s0);
            stencil.sync();
            std::swap(s0, s1);
        }
        std::printf("GPU time %.6lfs\n", std::chrono::duration<double>(std::chrono::steady_clock::now() - start).count());
        stencil.transferCubeToCPU(host_p0.data(), "p0");
    } else if (mode == "local" && argc >= 3) {
        const int n = atoi(argv[2]);
        const fdtdx_program prop_kernel = fdtdx::star_program(argc > 3 ? (unsigned)atoi(argv[3]) : 1);
        const size_t steps = argc > 4 ? (size_t)atoi(argv[4]) : 1000;
        std::vector<std::unique_ptr<Slab>> slabs;
        std::vector<Slab *> raw;
        for (int r = 0; r < n; r++) {
            slabs.emplace_back(new Slab((unsigned)r, (unsigned)n, 0));
            raw.push_back(slabs.back().get());
        }
        Slab::commInitLocal(raw.data(), n);
        for (auto &s : slabs) {
            s->mallocCube("p0", true);
            s->mallocCube("p1", true);
            s->initMt("p0", "p1");
        }
        auto start = std::chrono::steady_clock::now();
        std::string s0 = "p0", s1 = "p1";
        for (size_t t = 0; t < steps; t++) {
            // one host thread issues every slab: the loop body of fdtd1.cu:70-78 in two phases
            for (auto &s : slabs) {
                s->backupCubeHaloBackup(s0);
                s->propagateHaloTopBackup(s0, s1, true, prop_kernel);
                s->propagateHaloButtomBackup(s0, s1, true, prop_kernel);
                s->propagate(s0, s1, true, prop_kernel);
                s->exchangeStart(s0);
            }
            for (auto &s : slabs) {
                s->exchangeWait();
                s->restoreCubeHaloBackup(s0);
            }
            std::swap(s0, s1);
        }
        for (auto &s : slabs) s->sync();
        std::printf("GPU time %.6lfs\n", std::chrono::duration<double>(std::chrono::steady_clock::now() - start).count());
        slabs[0]->transferCubeToCPU(host_p0.data(), "p0");  // stencil.h:610-637: gathers every slab's planes
    } else {
        std::fprintf(stderr, "usage: %s spmd <rank> <nprocs> <id file> [radius] [steps] | local <nslabs> [radius] [steps]\n", argv[0]);
        return 2;
    }
    std::printf("crc32 %08x\n", crc32_of(host_p0.data(), host_p0.size() * sizeof(float)));
    return 0;
}
