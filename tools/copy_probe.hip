// copy_probe.hip -- what a 1:1 read+write stream over two sweep-sized cubes reaches on this box.
//
// The single-step sweeps read one cube and write another once each; this probe measures that traffic
// pattern with nothing else in the kernel, so that "at the speed of a copy" has a hand-written yardstick:
// raw buffer loads of 16 bytes per lane, raw buffer stores (non-temporal or plain), the cube layout of
// the sweeps (512 planes of 512 rows at a pitch of 544 floats).
//
//   flat   grid-stride copy: every workgroup moves `unroll` x 16 B per lane per trip;
//   march  the sweeps' own access pattern: a workgroup owns `rows` full rows of every plane of a z chunk
//          and walks the planes, keeping `pf` planes of loads in flight ahead of the store.
//
//   build: hipcc -O3 --offload-arch=gfx950 tools/copy_probe.hip -o tools/copy_probe
//   run:   tools/copy_probe            (prints one line per configuration, TB/s = bytes read + written / time)
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(e)                                                                        \
    do {                                                                                \
        hipError_t r_ = (e);                                                            \
        if (r_ != hipSuccess) {                                                         \
            std::fprintf(stderr, "%s: %s\n", #e, hipGetErrorString(r_));               \
            std::exit(1);                                                               \
        }                                                                               \
    } while (0)

typedef unsigned int u32x4 __attribute__((__vector_size__(16)));

__device__ __forceinline__ __amdgpu_buffer_rsrc_t rsrc(const void *p, unsigned bytes)
{
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p), 0, bytes, 0x00020000);
}

// flat: segment s of the cube = threads x UNROLL quads; workgroup b takes segments b, b+grid, ...
template <int UNROLL, int ST_AUX, int LD_AUX>
__global__ void flat_copy(const float *in, float *out, long long nquads)
{
    const long long per = (long long)blockDim.x * UNROLL;
    for (long long s = blockIdx.x; s * per < nquads; s += gridDim.x) {
        const long long q0 = s * per;
        const unsigned bytes = (unsigned)((nquads - q0 < per ? nquads - q0 : per) * 16);
        const __amdgpu_buffer_rsrc_t ri = rsrc(in + q0 * 4, bytes), ro = rsrc(out + q0 * 4, bytes);
        u32x4 v[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; u++) v[u] = __builtin_amdgcn_raw_buffer_load_b128(ri, (threadIdx.x + u * blockDim.x) * 16u, 0, LD_AUX);
#pragma unroll
        for (int u = 0; u < UNROLL; u++) __builtin_amdgcn_raw_buffer_store_b128(v[u], ro, (threadIdx.x + u * blockDim.x) * 16u, 0, ST_AUX);
    }
}

// march: workgroup (tile, chunk) owns ROWS rows x 136 quads of every plane in its z chunk; thread t owns
// quads t, t+threads, ... (RPT per plane); plane z+PF is loaded while plane z is stored
template <int RPT, int PF, int ST_AUX>
__global__ void march_copy(const float *in, float *out, long long plane, int rows_per_tile, int ny, int nz, int zchunk, int tiles)
{
    const int tile = blockIdx.x % tiles, chunk = blockIdx.x / tiles;
    const int zs = chunk * zchunk, ze = min(zs + zchunk, nz);
    const int y0 = tile * rows_per_tile;
    const unsigned plane_bytes = (unsigned)(plane * 4);
    const unsigned tile_quads = (unsigned)(min(rows_per_tile, ny - y0) * 136);
    unsigned off[RPT];
#pragma unroll
    for (int r = 0; r < RPT; r++) {
        const unsigned q = threadIdx.x + r * blockDim.x;
        off[r] = q < tile_quads ? (unsigned)(y0 * 136 + q) * 16u : 0x80000000u;
    }
    u32x4 v[PF + 1][RPT];
#pragma unroll
    for (int k = 0; k < PF; k++)
#pragma unroll
        for (int r = 0; r < RPT; r++)
            v[k][r] = __builtin_amdgcn_raw_buffer_load_b128(rsrc(in + (long long)(zs + k) * plane, zs + k < ze ? plane_bytes : 0u), off[r], 0, 0);
    for (int z = zs; z < ze; z += PF + 1) {
#pragma unroll
        for (int ph = 0; ph <= PF; ph++) {
            const int zz = z + ph;
            if (zz < ze) {
                const int slot_new = (ph + PF) % (PF + 1);
#pragma unroll
                for (int r = 0; r < RPT; r++)
                    v[slot_new][r] = __builtin_amdgcn_raw_buffer_load_b128(rsrc(in + (long long)(zz + PF) * plane, zz + PF < ze ? plane_bytes : 0u), off[r], 0, 0);
#pragma unroll
                for (int r = 0; r < RPT; r++)
                    __builtin_amdgcn_raw_buffer_store_b128(v[ph][r], rsrc(out + (long long)zz * plane, plane_bytes), off[r], 0, ST_AUX);
            }
        }
    }
}

// skel: the single-step sweep without its arithmetic.  A workgroup of TQ x LY threads owns a tile of TQ quads x LY*RY rows
// (only the 128 quads that hold the 512 cells of a row are touched, pad columns skipped) of every plane of its z chunk;
// plane z+PF is loaded while plane z is stored.  SYNC adds what the sweep does per plane besides moving bytes: the
// plane goes through LDS with one barrier.  HALO > 0 also loads HALO rows above and below the tile (of the plane being
// loaded: AHEAD = 1, or of the plane after the one being stored: AHEAD = 0, which is when the sweep needs them) and drops them.
// XCD = 1: the sweeps' placement (each XCD, i.e. blocks b, b+8, ..., owns a contiguous run of tiles); 0: tiles in launch order.
template <int TQ, int LY, int RY, int PF, int SYNC, int HALO, int AHEAD, int XCD>
__global__ __launch_bounds__(TQ * LY) void skel_copy(const float *in, float *out, long long plane, int ny, int nz, int zchunk, int ytiles)
{
    constexpr int NT = TQ * LY, CY = LY * RY, XT = 128 / TQ;
    __shared__ u32x4 lds[2][SYNC ? NT * RY : 1];
    const int nb = gridDim.x, bb = blockIdx.x, xcd = bb & 7, slot = bb >> 3, qn = nb >> 3, rn = nb & 7;
    const int lb = XCD ? (xcd < rn ? xcd * (qn + 1) : rn * (qn + 1) + (xcd - rn) * qn) + slot : bb;
    const int bx = lb % XT, tile = (lb / XT) % ytiles, chunk = lb / (XT * ytiles);
    const int zs = chunk * zchunk, ze = min(zs + zchunk, nz);
    const int tx = threadIdx.x % TQ, ty = threadIdx.x / TQ;
    const unsigned plane_bytes = (unsigned)(plane * 4);
    unsigned off[RY];
#pragma unroll
    for (int j = 0; j < RY; j++) {
        const int y = tile * CY + ty + j * LY;
        off[j] = y < ny ? (unsigned)(y * 136 + bx * TQ + tx) * 16u : 0x80000000u;
    }
    constexpr int NH = HALO ? (2 * HALO + LY - 1) / LY : 1;
    unsigned hoff[NH];
#pragma unroll
    for (int k = 0; k < NH; k++) {
        const int r = ty + k * LY;  // 0 .. 2*HALO-1: rows above, then rows below
        const int hy = r < HALO ? tile * CY - HALO + r : tile * CY + CY + (r - HALO);
        hoff[k] = (HALO && r < 2 * HALO && hy >= 0 && hy < ny) ? (unsigned)(hy * 136 + bx * TQ + tx) * 16u : 0x80000000u;
    }
    u32x4 v[PF + 1][RY], h[NH];
    u32x4 acc = {0, 0, 0, 0};
#pragma unroll
    for (int k = 0; k < NH; k++) h[k] = acc;
#pragma unroll
    for (int k = 0; k < PF; k++)
#pragma unroll
        for (int j = 0; j < RY; j++)
            v[k][j] = __builtin_amdgcn_raw_buffer_load_b128(rsrc(in + (long long)(zs + k) * plane, zs + k < ze ? plane_bytes : 0u), off[j], 0, 0);
    int par = 0;
    for (int z = zs; z < ze; z += PF + 1) {
#pragma unroll
        for (int ph = 0; ph <= PF; ph++) {
            const int zz = z + ph;
            if (zz < ze) {
                const int slot_new = (ph + PF) % (PF + 1);
                if (SYNC)
#pragma unroll
                    for (int j = 0; j < RY; j++) lds[par][threadIdx.x + j * NT] = v[ph][j];
#pragma unroll
                for (int j = 0; j < RY; j++)
                    v[slot_new][j] = __builtin_amdgcn_raw_buffer_load_b128(rsrc(in + (long long)(zz + PF) * plane, zz + PF < ze ? plane_bytes : 0u), off[j], 0, 0);
                if (HALO) {
                    const int hz = AHEAD ? zz + PF : zz + 1;
#pragma unroll
                    for (int k = 0; k < NH; k++) {
                        acc |= h[k];
                        h[k] = __builtin_amdgcn_raw_buffer_load_b128(rsrc(in + (long long)hz * plane, hz < ze ? plane_bytes : 0u), hoff[k], 0, 0);
                    }
                }
                if (SYNC) __syncthreads();
#pragma unroll
                for (int j = 0; j < RY; j++) {
                    u32x4 w = v[ph][j];
                    if (SYNC) w ^= lds[par][(threadIdx.x + j * NT + TQ) % (NT * RY)];  // a neighbouring row's quad, as the sweep reads y neighbours
                    __builtin_amdgcn_raw_buffer_store_b128(w, rsrc(out + (long long)zz * plane, plane_bytes), off[j], 0, 2);
                }
                if (SYNC) par ^= 1;
            }
        }
    }
    if (HALO && acc[0] == 0x12345678u) out[0] = 1.f;  // keeps the halo loads alive
}

static float *a, *b;
static const int NZ = 512, NY = 512, PX = 544;
static const long long PLANE = (long long)NY * PX, N = PLANE * NZ;

template <class F>
static double time_us(F launch, int reps = 30)
{
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    for (int i = 0; i < 200; i++) launch();  // the device clock settles over the first tens of milliseconds
    CHECK(hipEventRecord(e0, nullptr));
    for (int i = 0; i < reps; i++) launch();
    CHECK(hipEventRecord(e1, nullptr));
    CHECK(hipEventSynchronize(e1));
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    CHECK(hipGetLastError());
    return ms * 1e3 / reps;
}

template <int UNROLL, int ST_AUX, int LD_AUX>
static void flat(int threads, int grid)
{
    const double us = time_us([&] { hipLaunchKernelGGL((flat_copy<UNROLL, ST_AUX, LD_AUX>), dim3(grid), dim3(threads), 0, nullptr, a, b, N / 4); });
    std::printf("flat  threads=%4d grid=%5d unroll=%d store=%s load=%s : %7.1f us  %5.2f TB/s\n", threads, grid, UNROLL, ST_AUX ? "nt" : "plain",
                LD_AUX ? "nt" : "plain", us, 2.0 * N * 4 / us * 1e-6);
    std::fflush(stdout);
}

template <int RPT, int PF, int ST_AUX>
static void march(int threads, int chunks)
{
    const int rows = threads * RPT / 136 > 0 ? (threads * RPT) / 136 : 1;  // rows a workgroup covers per plane
    const int tiles = (NY + rows - 1) / rows;
    const int zchunk = (NZ + chunks - 1) / chunks;
    const double us = time_us([&] {
        hipLaunchKernelGGL((march_copy<RPT, PF, ST_AUX>), dim3(tiles * chunks), dim3(threads), 0, nullptr, a, b, PLANE, rows, NY, NZ, zchunk, tiles);
    });
    std::printf("march threads=%4d quads/thread=%d rows/tile=%3d tiles=%3d chunks=%2d (grid %4d) planes ahead=%d store=%s : %7.1f us  %5.2f TB/s\n", threads,
                RPT, rows, tiles, chunks, tiles * chunks, PF, ST_AUX ? "nt" : "plain", us, 2.0 * N * 4 / us * 1e-6);
    std::fflush(stdout);
}

template <int TQ, int LY, int RY, int PF, int SYNC, int HALO, int AHEAD, int XCD = 1>
static void skel(int chunks)
{
    const int ytiles = (NY + LY * RY - 1) / (LY * RY), tiles = ytiles * (128 / TQ);
    const int zchunk = (NZ + chunks - 1) / chunks;
    const double us = time_us([&] {
        hipLaunchKernelGGL((skel_copy<TQ, LY, RY, PF, SYNC, HALO, AHEAD, XCD>), dim3(tiles * chunks), dim3(TQ * LY), 0, nullptr, a, b, PLANE, NY, NZ, zchunk,
                           ytiles);
    });
    const double bytes = 2.0 * 512 * 4 * NY * NZ;  // the 512 cells of every row, read and written once
    std::printf("skel  tile=%3dq x %2d rows threads=%4d rows/thread=%d chunks=%2d (grid %4d) ahead=%d %s halo=%d%s%s : %7.1f us  %5.2f TB/s\n", TQ, LY * RY,
                TQ * LY, RY, chunks, tiles * chunks, PF, SYNC ? "LDS+barrier" : "plain      ", HALO,
                HALO ? (AHEAD ? " (of the plane loaded)" : " (of the plane stored)") : "", XCD ? "" : " launch-order tiles", us, bytes / us * 1e-6);
    std::fflush(stdout);
}

int main(int argc, char **argv)
{
    CHECK(hipMalloc((void **)&a, N * 4));
    CHECK(hipMalloc((void **)&b, N * 4));
    CHECK(hipMemset(a, 1, N * 4));
    CHECK(hipMemset(b, 0, N * 4));
    const bool quick = argc > 1;  // any argument: only the sweep skeletons
    if (quick) {
        // full-row tiles, one workgroup per CU
        skel<128, 8, 1, 1, 1, 0, 0>(4);
        skel<128, 4, 1, 1, 1, 0, 0>(2);
        skel<128, 2, 1, 1, 1, 0, 0>(1);
        skel<128, 4, 2, 1, 1, 0, 0>(4);
        skel<128, 2, 2, 1, 1, 0, 0>(2);
        skel<128, 2, 4, 1, 1, 0, 0>(4);
        skel<128, 4, 1, 1, 1, 0, 0, 0>(2);
        skel<128, 8, 1, 1, 1, 0, 0, 0>(4);
        // with halo rows (radius 1, 4, 8)
        skel<128, 4, 1, 1, 1, 1, 0>(2);
        skel<128, 4, 1, 1, 1, 4, 0>(2);
        skel<128, 4, 1, 1, 1, 4, 1>(2);
        skel<128, 4, 1, 5, 1, 4, 0>(2);
        skel<128, 4, 1, 5, 1, 4, 1>(2);
        skel<128, 8, 1, 1, 1, 1, 0>(4);
        skel<128, 8, 1, 5, 1, 4, 0>(4);
        skel<128, 8, 1, 5, 1, 4, 1>(4);
        skel<128, 8, 1, 9, 1, 8, 0>(4);
        skel<128, 8, 1, 9, 1, 8, 1>(4);
        skel<128, 4, 2, 5, 1, 4, 0>(4);
        skel<128, 4, 2, 5, 1, 4, 1>(4);
        // the radius 5-8 tile shapes
        skel<32, 32, 1, 1, 1, 0, 0>(4);
        skel<32, 32, 1, 9, 1, 8, 0>(4);
        skel<32, 32, 1, 9, 1, 8, 1>(4);
        skel<32, 16, 1, 9, 1, 8, 0>(2);
        skel<32, 16, 2, 9, 1, 8, 0>(4);
        skel<64, 16, 1, 9, 1, 8, 0>(4);
        skel<64, 8, 1, 9, 1, 8, 0>(2);
        skel<64, 8, 2, 9, 1, 8, 0>(4);
        skel<64, 8, 2, 9, 1, 8, 1>(4);
        // two workgroups per CU
        skel<128, 4, 1, 1, 1, 0, 0>(4);
        skel<128, 2, 1, 1, 1, 0, 0>(2);
        skel<128, 2, 1, 1, 1, 0, 0>(4);
        CHECK(hipFree(a));
        CHECK(hipFree(b));
        return 0;
    }
    for (int threads : {256, 512, 1024})
        for (int grid : {256, 512, 1024, 2048, 8192}) {
            if ((long long)threads * grid > 4 * 1024 * 2048) continue;
            flat<1, 2, 0>(threads, grid);
            flat<4, 2, 0>(threads, grid);
            flat<8, 2, 0>(threads, grid);
        }
    flat<4, 0, 0>(256, 2048);
    flat<4, 0, 0>(1024, 512);
    flat<4, 2, 2>(256, 2048);
    flat<4, 2, 2>(1024, 512);
    // the sweeps' shape: 1024 threads, one or two quads per thread per plane, 1-4 planes in flight
    march<1, 1, 2>(1024, 4);
    march<1, 2, 2>(1024, 4);
    march<1, 4, 2>(1024, 4);
    march<2, 1, 2>(1024, 8);
    march<2, 2, 2>(1024, 8);
    march<2, 4, 2>(1024, 8);
    march<2, 2, 2>(512, 4);
    march<2, 4, 2>(512, 4);
    march<4, 2, 2>(512, 8);
    march<4, 2, 2>(256, 4);
    march<4, 4, 2>(256, 4);
    march<2, 2, 0>(1024, 8);
    march<1, 2, 2>(1024, 8);
    march<1, 2, 2>(1024, 16);
    CHECK(hipFree(a));
    CHECK(hipFree(b));
    return 0;
}
