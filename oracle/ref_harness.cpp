// ref_harness.cpp -- thin C-callable shell around the reference's OWN host golden loops.
//
// TEST INFRASTRUCTURE ONLY (never linked into the product).
//
// The reference drivers (fdtdN.cu, fdtdspc.cu, jacobi.cu, main.cu) cannot be built here as
// whole files: they include <cuda.h>, <mpi.h> and device lambdas that need nvcc.  Their CPU
// golden loops, however, are plain C++ statements.  oracle/build_ref.sh cuts those statements
// out of /root/reference by line range AT BUILD TIME into a scratch include directory
// (frag_*.inc, deleted after the build); this file only supplies the surrounding function
// signatures whose parameter names are the identifiers the statements use (p0, p1, nz, ny, nx,
// host_A, host_B, NZ, NY, NX, vel, sum, addr, na).  No reference text lives in this repository.
#include <cstddef>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <utility>

#include "frag_get_macro.inc"      // fdtd1.cu:12   #define GET(z,y,x,nz,ny,nx)
#include "frag_jacobi_consts.inc"  // jacobi.cu:15-21  const float cc..cb

extern "C" {

// fdtd1.cu:21-31 -- std::mt19937_64(19950918+z) initial fields, whole grid.
void ref_init_mt(float *p0, float *p1, size_t nz, size_t ny, size_t nx)
{
#include "frag_init_mt.inc"
}

// main.cu:24-36 -- same generator, third draw is vel, sum = 0.
void ref_init_mt3(float *p0, float *p1, float *vel, float *sum, size_t nz, size_t ny, size_t nx)
{
#include "frag_init_mt3.inc"
}

// fdtdN.cu:42-(46+3N) -- one golden step p0 = f(p1) of the radius-N star.
int ref_star_step(int M, float *p0, float *p1, size_t nz, size_t ny, size_t nx)
{
    switch (M) {
    case 1: {
#include "frag_star1.inc"
    } break;
    case 2: {
#include "frag_star2.inc"
    } break;
    case 3: {
#include "frag_star3.inc"
    } break;
    case 4: {
#include "frag_star4.inc"
    } break;
    case 5: {
#include "frag_star5.inc"
    } break;
    case 6: {
#include "frag_star6.inc"
    } break;
    case 7: {
#include "frag_star7.inc"
    } break;
    case 8: {
#include "frag_star8.inc"
    } break;
    default:
        return -1;
    }
    return 0;
}

// fdtdspc.cu:42-52 -- one golden step of the 13-point stencil.
void ref_spc13_step(float *p0, float *p1, size_t nz, size_t ny, size_t nx)
{
#include "frag_spc13.inc"
}

// jacobi.cu:412-417 -- initial field, both arrays.
void ref_init_jacobi(float *host_A, float *host_B, int64_t NZ, int64_t NY, int64_t NX)
{
#include "frag_jacobi_init.inc"
}

// jacobi.cu:475-485 -- one step host_A = f(host_B) of the check loop.
void ref_jacobi_step(float *host_A, float *host_B, int64_t NZ, int64_t NY, int64_t NX)
{
#include "frag_jacobi_step.inc"
}

// main.cu:54-84 -- one composite step: sweep, inject, filter, product-accumulate.
void ref_main_step(float *p0, float *p1, float *vel, float *sum, size_t *addr, size_t na, size_t nz,
                   size_t ny, size_t nx)
{
#include "frag_main_step.inc"
}

// Ping-pong driver around the step functions, mirroring the drivers' "step; swap" loop
// (fdtd1.cu:40-52).  kind: 0 star, 1 spc13, 2 jacobi.
int ref_run(int kind, int M, float *a, float *b, size_t nz, size_t ny, size_t nx, int steps)
{
    float *out = a, *in = b;
    for (int t = 0; t < steps; t++) {
        if (kind == 0) {
            if (ref_star_step(M, out, in, nz, ny, nx)) return -1;
        } else if (kind == 1) {
            ref_spc13_step(out, in, nz, ny, nx);
        } else {
            ref_jacobi_step(out, in, (int64_t)nz, (int64_t)ny, (int64_t)nx);
        }
        std::swap(out, in);
    }
    return 0;
}

}  // extern "C"

#ifdef REF_BENCH_MAIN
// Stand-alone timing of the golden loop (the CPU baseline of bench.py, kind "reference"):
//   ref_bench <kind> <M> <nz> <ny> <nx> <steps>   ->  one JSON line on stdout   (kind 3: main.cu pipeline)
#include <chrono>
#include <omp.h>
int main(int argc, char **argv)
{
    if (argc < 7) {
        fprintf(stderr, "usage: %s kind M nz ny nx steps\n", argv[0]);
        return 2;
    }
    int kind = atoi(argv[1]), M = atoi(argv[2]);
    size_t nz = strtoull(argv[3], 0, 10), ny = strtoull(argv[4], 0, 10), nx = strtoull(argv[5], 0, 10);
    int steps = atoi(argv[6]);
    float *a = new float[nz * ny * nx], *b = new float[nz * ny * nx];
    double s;
    if (kind == 3) {
        // the composite pipeline of main.cu:52-87: 10000 injection addresses (main.cu:11,37-41), step; swap
        float *vel = new float[nz * ny * nx], *sum = new float[nz * ny * nx];
        const size_t na = 10000;
        size_t *addr = new size_t[na];
        ref_init_mt3(a, b, vel, sum, nz, ny, nx);
        for (size_t i = 0; i < na; i++) {
            size_t x = rand() % nx, y = rand() % ny, z = rand() % nz;
            addr[i] = (z * ny + y) * nx + x;
        }
        ref_main_step(a, b, vel, sum, addr, na, nz, ny, nx);  // warm-up
        auto t0 = std::chrono::steady_clock::now();
        for (int t = 0; t < steps; t++) {
            ref_main_step(a, b, vel, sum, addr, na, nz, ny, nx);
            std::swap(a, b);
        }
        auto t1 = std::chrono::steady_clock::now();
        s = std::chrono::duration<double>(t1 - t0).count();
        M = 4;
    } else {
        if (kind == 2)
            ref_init_jacobi(a, b, (int64_t)nz, (int64_t)ny, (int64_t)nx);
        else
            ref_init_mt(a, b, nz, ny, nx);
        ref_run(kind, M, a, b, nz, ny, nx, 1);  // warm-up: page in both arrays
        auto t0 = std::chrono::steady_clock::now();
        ref_run(kind, M, a, b, nz, ny, nx, steps);
        auto t1 = std::chrono::steady_clock::now();
        s = std::chrono::duration<double>(t1 - t0).count();
    }
    size_t h = (kind == 0 || kind == 3) ? (size_t)M : 1;
    double cells = double(nz - 2 * h) * double(ny - 2 * h) * double(nx - 2 * h) * steps;
    double chk = 0;
    for (size_t i = 0; i < nz * ny * nx; i += 4097) chk += a[i] + b[i];
    printf("{\"gcells_per_s\": %.6f, \"seconds\": %.6f, \"steps\": %d, \"threads\": %d, \"checksum\": %.9g}\n",
           cells / s * 1e-9, s, steps, omp_get_max_threads(), chk);
    return 0;
}
#endif
