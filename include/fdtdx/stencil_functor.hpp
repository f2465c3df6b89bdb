// fdtdx/stencil_functor.hpp -- the reference's generic functor interface, for hipcc (gfx950).
//
// The closed programs of fdtdx.h cover the stencils the reference ships.  User-written stencils
// reach the device through this header instead: it keeps the calling conventions of the
// reference's propagate / inject / filt templates (stencil.h:684-768, 810-823) so that a driver's
// device lambdas compile unchanged:
//
//   sweep   f(z, y, x, addr, out, zl, sz, yl, sy, xl, sx, args...)      (fdtd1.cu:61, stencil.h:90,167,235)
//             z, y, x  global logical coordinates of the cell
//             addr     linear index of the cell in the local padded cube (ghost planes included)
//             out      base of the output cube: the functor writes out[addr]
//             zl/yl/xl the cell seen along each axis: neighbour r is zl[r*sz], yl[r*sy], xl[r*sx]
//           FDTD = true  ("star" mode): the three views live in different memories (registers for
//             z, the LDS plane for y and x), so only pure-axis offsets are legal (stencil.h:127-171);
//           FDTD = false ("box" mode): all three alias the input cube with strides (plane, pitch, 1)
//             and mixed offsets such as yl[sy+sz] are legal (stencil.h:75-93, fdtdspc.cu:70-72).
//   inject  f(i, args...) for i < n                                      (stencil.h:337-343)
//   filt    f(z, y, x, addr, args...) for every cell of the planes this rank owns (stencil.h:345-355)
//
// Cubes, transfers, decomposition and halo regions come from fdtdx::Stencil (the C ABI); this
// header only adds the templated launches, which is why it needs hipcc.  The star-mode kernel is
// the reference's scheme (z queue in registers, centre plane in LDS) laid out for wave64: tiles of
// 128 cells x 16 rows, one 16-byte quad per thread, halo rows and quads fetched by a strided loop,
// the functor called once per cell.  It is the general path (a radius-3 lambda runs at about two thirds of
// the speed of the closed program); the closed programs of fdtdx.h run the tuned kernels.
#pragma once
#include <hip/hip_runtime.h>

#include <algorithm>
#include <limits>

#include "stencil.hpp"

namespace fdtdx {

struct FunctorGeom {
    long long plane;       // floats per plane
    int px, nx, ny;        // row pitch, logical extents
    long long zg0;         // global z of local plane 0
    int z_lo, z_hi;        // local planes to visit
    int zchunk;            // planes per workgroup (star mode)
};

// ---- star mode --------------------------------------------------------------------------------------
// A workgroup owns a tile of TXQ quads (4*TXQ cells, x unit-stride) by TY rows and marches a chunk of z planes.
// A thread owns one quad: it moves 16-byte quads (global loads, LDS image of the current plane with its halo rows
// and halo quads, double-buffered: one barrier per plane), keeps the z neighbours of its four cells in registers,
// and calls the functor once per cell with zl into those registers and yl/xl into the LDS image, as the
// reference's prop_inner/prop_center kernels do (stencil.h:127-289).
template <typename T, int M, int TXQ, int TY, typename gsz, typename gssz, typename F, typename... Args>
__global__ __launch_bounds__(TXQ *TY) void functor_star_kernel(FunctorGeom g, T *out, const T *in, F f, Args... args)
{
    static_assert(sizeof(T) == 4, "quads of four 4-byte cells");
    constexpr int HQ = (M + 3) / 4;                      // halo quads on each side in x
    constexpr int LXQ = TXQ + 2 * HQ, LX = 4 * LXQ;      // LDS row: quads, cells
    constexpr int LY = TY + 2 * M, NT = TXQ * TY;
    constexpr int NH = 2 * M * TXQ + 2 * HQ * TY;        // halo quads: rows above/below, quads left/right (no corners)
    __shared__ float4 tile4[2][LY][LXQ];
    // rows and planes are whole numbers of quads and cubes are 16-byte aligned (fdtdx_geometry): said out loud so
    // that the compiler may merge what the functor stores for the four cells of a quad
    __builtin_assume((g.px & 3) == 0);
    __builtin_assume((g.plane & 3) == 0);
    out = static_cast<T *>(__builtin_assume_aligned(out, 16));
    const int tx = threadIdx.x % TXQ, ty = threadIdx.x / TXQ;
    const int x0 = 4 * ((int)blockIdx.x * TXQ + tx), y = M + (int)blockIdx.y * TY + ty;
    const int zs = g.z_lo + (int)blockIdx.z * g.zchunk;
    const int ze = min(zs + g.zchunk, g.z_hi);
    const bool loads = (x0 < g.nx) && (y < g.ny);
    const bool yin = y < g.ny - M;
    const long long col = (long long)y * g.px + x0;
    const float4 zero = make_float4(0.f, 0.f, 0.f, 0.f);
    T qz[4][2 * M + 1];  // per cell: planes z-M .. z+M
#pragma unroll
    for (int k = 0; k < 2 * M; k++) {
        const float4 v = loads ? *reinterpret_cast<const float4 *>(in + (long long)(zs - M + k) * g.plane + col) : zero;
        qz[0][k + 1] = v.x, qz[1][k + 1] = v.y, qz[2][k + 1] = v.z, qz[3][k + 1] = v.w;
    }
    int par = 0;
    for (int z = zs; z < ze; z++, par ^= 1) {
        const float4 nv = loads ? *reinterpret_cast<const float4 *>(in + (long long)(z + M) * g.plane + col) : zero;
#pragma unroll
        for (int i = 0; i < 4; i++) {
#pragma unroll
            for (int k = 0; k < 2 * M; k++) qz[i][k] = qz[i][k + 1];
        }
        qz[0][2 * M] = nv.x, qz[1][2 * M] = nv.y, qz[2][2 * M] = nv.z, qz[3][2 * M] = nv.w;
        tile4[par][M + ty][HQ + tx] = make_float4(qz[0][M], qz[1][M], qz[2][M], qz[3][M]);
        const T *pc = in + (long long)z * g.plane;
        for (int h = threadIdx.x; h < NH; h += NT) {
            int ly, lq;
            if (h < 2 * M * TXQ) {
                const int r = h / TXQ;
                ly = (r < M) ? r : TY + r;
                lq = HQ + h % TXQ;
            } else {
                const int h2 = h - 2 * M * TXQ, c = h2 % (2 * HQ);
                ly = M + h2 / (2 * HQ);
                lq = (c < HQ) ? c : TXQ + c;
            }
            const int gy = (int)blockIdx.y * TY + ly, gq = (int)blockIdx.x * TXQ + lq - HQ;  // tile origin: row M, quad 0
            tile4[par][ly][lq] = (gy < g.ny && gq >= 0 && 4 * gq < g.nx) ? *reinterpret_cast<const float4 *>(pc + (long long)gy * g.px + 4 * gq) : zero;
        }
        __syncthreads();  // the other image is only rewritten after the next barrier, when everyone has left this plane's reads behind
        T *row = reinterpret_cast<T *>(&tile4[par][M + ty][HQ + tx]);
        const gsz addr0 = (gsz)((long long)z * g.plane + col);
        auto cell = [&](int i) {
            f((gsz)(g.zg0 + z), (gsz)y, (gsz)(x0 + i), (gsz)(addr0 + i), out, &qz[i][M], (gssz)1, row + i, (gssz)LX, row + i, (gssz)1, args...);
        };
        if (yin && x0 >= M && x0 + 3 < g.nx - M) {
            // all four cells are interior: one straight block, so that the functor's four stores can go out as one
            cell(0), cell(1), cell(2), cell(3);
        } else if (yin) {
#pragma unroll
            for (int i = 0; i < 4; i++)
                if (x0 + i >= M && x0 + i < g.nx - M) cell(i);
        }
    }
}

// ---- box mode ---------------------------------------------------------------------------------------
template <typename T, int M, typename gsz, typename gssz, typename F, typename... Args>
__global__ void functor_box_kernel(FunctorGeom g, T *out, const T *in, F f, Args... args)
{
    const int x = M + blockIdx.x * blockDim.x + threadIdx.x;
    const int y = M + blockIdx.y;
    const int z = g.z_lo + blockIdx.z;
    if (x >= g.nx - M) return;
    const long long a = (long long)z * g.plane + (long long)y * g.px + x;
    T *c = const_cast<T *>(in) + a;
    f((gsz)(g.zg0 + z), (gsz)y, (gsz)x, (gsz)a, out, c, (gssz)g.plane, c, (gssz)g.px, c, (gssz)1, args...);
}

template <typename gsz, typename F, typename... Args>
__global__ void functor_inject_kernel(gsz n, F f, Args... args)
{
    const gsz i = (gsz)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) f(i, args...);
}

template <typename gsz, typename F, typename... Args>
__global__ void functor_filt_kernel(FunctorGeom g, F f, Args... args)
{
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    if (x >= g.nx) return;
    const int y = blockIdx.y, z = g.z_lo + blockIdx.z;
    f((gsz)(g.zg0 + z), (gsz)y, (gsz)x, (gsz)((long long)z * g.plane + (long long)y * g.px + x), args...);
}

#define FDTDX_HIP_ASSERT(expr)                                                                   \
    do {                                                                                         \
        hipError_t e_ = (expr);                                                                  \
        if (e_ != hipSuccess) {                                                                  \
            std::fprintf(stderr, "ERROR AT [ %s ] : %s\n", #expr, hipGetErrorString(e_));        \
            std::abort();                                                                        \
        }                                                                                        \
    } while (0)

// The reference's class with its functor-taking methods (template header of stencil.h:357).
template <typename DataType, size_t ZSize, size_t YSize, size_t XSize, size_t M, bool FDTD = true, typename gpu_size_t = uint32_t,
          typename gpu_signed_size_t = int32_t>
class FunctorStencil : public Stencil<DataType, ZSize, YSize, XSize, M, FDTD> {
    using Base = Stencil<DataType, ZSize, YSize, XSize, M, FDTD>;

    FunctorGeom geom(uint64_t z_lo, uint64_t z_hi) const
    {
        const fdtdx_geometry &g = this->geometry();
        // gpu_size_t addresses must reach the whole local cube (the reference's uint32_t limit, fdtd1.cu:13-14)
        if (sizeof(gpu_size_t) < 8 && g.cube_elems > (uint64_t)std::numeric_limits<gpu_size_t>::max()) {
            std::fprintf(stderr, "ERROR: local cube of %llu elements needs a 64-bit gpu_size_t\n", (unsigned long long)g.cube_elems);
            std::abort();
        }
        FunctorGeom f;
        f.plane = (long long)g.plane, f.px = (int)g.pitch_x, f.nx = (int)XSize, f.ny = (int)YSize;
        f.zg0 = (long long)g.first_global, f.z_lo = (int)z_lo, f.z_hi = (int)z_hi, f.zchunk = (int)(z_hi - z_lo);
        return f;
    }

    // the functor overloads stand beside the closed-program methods of the base class
    template <typename F>
    using if_functor = std::enable_if_t<!std::is_convertible<F, std::string>::value && !std::is_same<std::decay_t<F>, fdtdx_program>::value>;

public:
    using Base::Base;
    using Base::filt;
    using Base::inject;
    using Base::propagate;
    using Base::propagatePlanes;

    // stencil.h:684-768 on local planes [z_lo, z_hi); the whole-slab form below is the reference's propagate
    template <typename Function, typename... Arguments, typename = if_functor<Function>>
    void propagatePlanes(const std::string &output, const std::string &input, uint64_t z_lo, uint64_t z_hi, Function kernel,
                         Arguments... args)
    {
        if (z_hi <= z_lo) return;
        DataType *out = this->getCube(output), *in = this->getCube(input);
        FunctorGeom g = geom(z_lo, z_hi);
        if (FDTD) {
            constexpr int TXQ = 32, TY = 16;  // 128 cells x 16 rows, one quad per thread
            const unsigned gx = (unsigned)(((XSize + 3) / 4 + TXQ - 1) / TXQ), gy = (unsigned)((YSize - 2 * M + TY - 1) / TY);
            // about one workgroup of 8 waves per CU and round, chunks of at least 8*M planes (every chunk refills its z queue)
            long long nz = (long long)(z_hi - z_lo), chunks = std::max<long long>(1, 512 / ((long long)gx * gy));
            chunks = std::max<long long>(1, std::min(chunks, nz / (long long)(8 * M)));
            g.zchunk = (int)((nz + chunks - 1) / chunks);
            const unsigned gz = (unsigned)((nz + g.zchunk - 1) / g.zchunk);
            hipLaunchKernelGGL((functor_star_kernel<DataType, (int)M, TXQ, TY, gpu_size_t, gpu_signed_size_t, Function, Arguments...>),
                               dim3(gx, gy, gz), dim3(TXQ * TY), 0, 0, g, out, in, kernel, args...);
        } else {
            const unsigned bx = 256;
            for (uint64_t z = z_lo; z < z_hi; z += 65535) {
                g.z_lo = (int)z;
                hipLaunchKernelGGL((functor_box_kernel<DataType, (int)M, gpu_size_t, gpu_signed_size_t, Function, Arguments...>),
                                   dim3((unsigned)((XSize - 2 * M + bx - 1) / bx), (unsigned)(YSize - 2 * M),
                                        (unsigned)std::min<uint64_t>(65535, z_hi - z)),
                                   dim3(bx), 0, 0, g, out, in, kernel, args...);
            }
        }
        FDTDX_HIP_ASSERT(hipGetLastError());
    }
    template <typename Function, typename... Arguments, typename = if_functor<Function>>
    void propagate(const std::string &output, const std::string &input, bool spill, Function kernel, Arguments... args)
    {
        (void)spill;
        const fdtdx_geometry &g = this->geometry();
        propagatePlanes(output, input, g.update_lo, g.update_hi, kernel, args...);
    }

    // stencil.h:810-815
    template <typename Function, typename... Arguments, typename = if_functor<Function>>
    void inject(size_t length, Function kernel, Arguments... args)
    {
        if (!length) return;
        hipLaunchKernelGGL((functor_inject_kernel<gpu_size_t, Function, Arguments...>), dim3((unsigned)((length + 255) / 256)), dim3(256), 0, 0,
                           (gpu_size_t)length, kernel, args...);
        FDTDX_HIP_ASSERT(hipGetLastError());
    }

    // stencil.h:817-823: every cell (shell included) of the planes this rank owns
    template <typename Function, typename... Arguments, typename = if_functor<Function>>
    void filt(Function kernel, Arguments... args)
    {
        const fdtdx_geometry &gg = this->geometry();
        FunctorGeom g = geom(gg.ghost_lo, gg.ghost_lo + gg.owned_planes);
        for (uint64_t z = gg.ghost_lo; z < gg.ghost_lo + gg.owned_planes; z += 65535) {
            g.z_lo = (int)z;
            hipLaunchKernelGGL((functor_filt_kernel<gpu_size_t, Function, Arguments...>),
                               dim3((unsigned)((XSize + 255) / 256), (unsigned)YSize,
                                    (unsigned)std::min<uint64_t>(65535, gg.ghost_lo + gg.owned_planes - z)),
                               dim3(256), 0, 0, g, kernel, args...);
        }
        FDTDX_HIP_ASSERT(hipGetLastError());
    }
};

}  // namespace fdtdx
