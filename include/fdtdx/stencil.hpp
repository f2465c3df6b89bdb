// fdtdx/stencil.hpp -- C++ face of the C ABI (fdtdx.h) with the reference's `Stencil<>` method names.
//
// Source-level stand-in for the host class of the reference (stencil.h:357-842) on the sweep path:
// the same template header <DataType, ZSize, YSize, XSize, M, FDTD>, the same method names and
// argument order, the same error behaviour (message to stderr, then abort -- stencil.h:16-22).
// What changes at the call site: the device lambda passed to propagate() becomes an
// `fdtdx_program` (a C ABI cannot carry a lambda), and rank/nprocs come from the constructor
// instead of MPI::COMM_WORLD (the launcher owns the communicator here).
#pragma once
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <type_traits>
#include <utility>

#include "../fdtdx.h"

#define FDTDX_ASSERT(expr)                                                                 \
    do {                                                                                   \
        int rc_ = (expr);                                                                  \
        if (rc_ != FDTDX_OK) {                                                             \
            std::fprintf(stderr, "ERROR AT [ %s ] : %s\n", #expr, fdtdx_last_error());     \
            std::abort();                                                                  \
        }                                                                                  \
    } while (0)

namespace fdtdx {

// Programs of the shipped drivers.
inline fdtdx_program star_program(unsigned radius, unsigned order = FDTDX_ORDER_CHAIN)
{  // fdtdN.cu:61-64 .. fdtd8.cu:82-93
    fdtdx_program p{};
    p.kind = FDTDX_STAR, p.radius = radius, p.order = order;
    const float c[7] = {0.01f, 0.02f, 0.03f, 0.04f, 0.05f, 0.06f, 0.07f};
    for (int i = 0; i < 7; i++) p.coef[i] = c[i];
    return p;
}
inline fdtdx_program spc13_program(unsigned order = FDTDX_ORDER_CHAIN)
{  // fdtdspc.cu:64-73
    fdtdx_program p{};
    p.kind = FDTDX_SPC13, p.radius = 1, p.order = order;
    const float c[13] = {0.01f, 0.02f, 0.03f, 0.04f, 0.05f, 0.06f, 0.07f, 0.02f, 0.03f, 0.04f, 0.05f, 0.06f, 0.07f};
    for (int i = 0; i < 13; i++) p.coef[i] = c[i];
    return p;
}
inline fdtdx_program jacobi7_program(unsigned order = FDTDX_ORDER_CHAIN)
{  // jacobi.cu:15-21,31-38
    fdtdx_program p{};
    p.kind = FDTDX_JACOBI7, p.radius = 1, p.order = order;
    const float c[7] = {0.01f, 0.02f, 0.03f, 0.04f, 0.05f, 0.06f, 0.07f};
    for (int i = 0; i < 7; i++) p.coef[i] = c[i];
    return p;
}
inline fdtdx_composite main_composite(unsigned order = FDTDX_ORDER_CHAIN)
{  // main.cu:102-119 (device), :55-72 (host); scal = 1.0/(6*8+1) as passed at main.cu:160
    fdtdx_composite c{};
    c.radius = 4, c.order = order;
    c.cz = 0.5, c.cy = -0.2, c.cx = -0.3, c.scal_num = 1.0, c.scal_den = 6 * 8 + 1, c.wx = 1.0, c.wy = 2.0, c.wz = 3.0;
    return c;
}

// The reference's full template header (stencil.h:357), so that a driver's declaration compiles as it stands (fdtd1.cu:17:
// Stencil<float, nz, ny, nx, 1, true, gpu_size_t, gpu_signed_size_t>).  The index types matter to the functor kernels of
// stencil_functor.hpp only; BLOCK_SIZE, THREADS, PADDINGL and PADDINGR are accepted and ignored: tile shapes are chosen per
// radius on the device and the row pitch is fixed by the library (fdtdx_geometry.pitch_x; no left padding).
template <typename DataType, size_t ZSize, size_t YSize, size_t XSize, size_t M, bool FDTD = true, typename gpu_size_t = uint32_t,
          typename gpu_signed_size_t = int32_t, size_t BLOCK_SIZE = 32, size_t THREADS = 256, size_t PADDINGL = 0, size_t PADDINGR = 0>
class Stencil {
    static_assert(std::is_same<DataType, float>::value, "the sweep kernels are fp32 (every shipped driver uses float)");
    static_assert(M <= 8 && ZSize >= 2 * M + 1, "stencil.h:387: M <= 8 and Z >= 2M+1");
    fdtdx_ctx *ctx_ = nullptr;
    fdtdx_geometry g_{};
    std::string boundary_of_;                // the cube whose boundary planes have been swept ahead of the interior
    bool done_lo_ = false, done_hi_ = false;

    void halo_sweep(const std::string &output, const std::string &input, bool, const fdtdx_program &prog, void *stream, int side)
    {
        if (boundary_of_ != output) boundary_of_ = output, done_lo_ = done_hi_ = false;
        const uint64_t r = prog.kind == FDTDX_STAR ? prog.radius : 1;
        const uint64_t lo = g_.ghost_lo ? g_.ghost_lo : r, hi = g_.ghost_hi ? g_.local_planes - g_.ghost_hi : g_.local_planes - r;
        if (side == 0 && g_.ghost_lo) {
            FDTDX_ASSERT(fdtdx_propagate_planes(ctx_, output.c_str(), input.c_str(), &prog, lo, std::min<uint64_t>(lo + M, hi), stream));
            done_lo_ = true;
        }
        if (side == 1 && g_.ghost_hi) {
            const uint64_t first = std::max<uint64_t>(hi - M, done_lo_ ? std::min<uint64_t>(lo + M, hi) : lo);
            FDTDX_ASSERT(fdtdx_propagate_planes(ctx_, output.c_str(), input.c_str(), &prog, first, hi, stream));
            done_hi_ = true;
        }
        if (g_.ghost_lo || g_.ghost_hi) FDTDX_ASSERT(fdtdx_comm_boundary_done(ctx_, stream));
    }

public:
    // stencil.h:385-413; rank/nprocs: the z-slab this process owns (MPI rank/size in the reference)
    explicit Stencil(unsigned rank = 0, unsigned nprocs = 1, int device = -1)
    {
        fdtdx_config cfg{};
        cfg.Z = ZSize, cfg.Y = YSize, cfg.X = XSize, cfg.M = (unsigned)M;
        cfg.rank = rank, cfg.nprocs = nprocs, cfg.device = device;
        FDTDX_ASSERT(fdtdx_create(&cfg, &ctx_));
        FDTDX_ASSERT(fdtdx_get_geometry(ctx_, &g_));
    }
    ~Stencil() { fdtdx_destroy(ctx_); }  // stencil.h:415-418 (no MPI::Finalize here)
    Stencil(const Stencil &) = delete;
    Stencil &operator=(const Stencil &) = delete;

    size_t getRank() { return (size_t)fdtdx_rank(ctx_); }     // stencil.h:420-423
    size_t getProcs() { return (size_t)fdtdx_nprocs(ctx_); }  // stencil.h:425-428
    void sync() { FDTDX_ASSERT(fdtdx_sync(ctx_)); }           // stencil.h:430-433
    const fdtdx_geometry &geometry() const { return g_; }
    fdtdx_ctx *handle() { return ctx_; }

    // stencil.h:553-582,468-477.  create_halo is accepted and ignored: ghost planes live in the cube.
    DataType *mallocCube(const std::string &name, bool create_halo = false)
    {
        (void)create_halo;
        FDTDX_ASSERT(fdtdx_malloc_cube(ctx_, name.c_str()));
        return getCube(name);
    }
    void free(const std::string &name) { FDTDX_ASSERT(fdtdx_free(ctx_, name.c_str())); }
    DataType *getCube(const std::string &name)
    {
        float *p = nullptr;
        FDTDX_ASSERT(fdtdx_get_cube(ctx_, name.c_str(), &p));
        return p;
    }
    // the first / last 3M planes of the local cube: what getCubeHalo{Top,Buttom}Backup alias for a cube without side buffers
    // in the reference (stencil.h:584-594).  Here no cube has side buffers (ghost planes live in the cube).
    DataType *getCubeHaloTopBackup(const std::string &name) { return getCube(name); }
    DataType *getCubeHaloButtomBackup(const std::string &name)
    {
        const uint64_t planes = g_.local_planes > 3 * M ? g_.local_planes - 3 * M : 0;
        return getCube(name) + planes * g_.plane;
    }
    // stencil.h:596-637.  transferCubeToCPU: every rank ends with the whole grid (the Recv/Send/Bcast of :623-636 happen
    // inside the library, device to device); transferOwnedToCPU writes only this rank's planes into their place.
    void transferCubeToGPU(const std::string &name, DataType *cpu) { FDTDX_ASSERT(fdtdx_transfer_cube_to_gpu(ctx_, name.c_str(), cpu)); }
    void transferCubeToCPU(DataType *cpu, const std::string &name) { FDTDX_ASSERT(fdtdx_gather_cube(ctx_, cpu, name.c_str())); }
    void transferOwnedToCPU(DataType *cpu, const std::string &name) { FDTDX_ASSERT(fdtdx_transfer_cube_to_cpu(ctx_, cpu, name.c_str())); }

    // stencil.h:684-768.  `spill` only chose launch bounds in the reference; ignored.  Where the planes next to the
    // neighbours have already been swept into this cube (propagateHalo{Top,Buttom}Backup since the last exchange), only
    // the planes between them are left to do.
    void propagate(const std::string &output, const std::string &input, bool spill, const fdtdx_program &prog, void *stream = nullptr)
    {
        (void)spill;
        if (boundary_of_ == output && (done_lo_ || done_hi_)) {
            const uint64_t r = prog.kind == FDTDX_STAR ? prog.radius : 1;
            uint64_t lo = g_.ghost_lo ? g_.ghost_lo : r, hi = g_.ghost_hi ? g_.local_planes - g_.ghost_hi : g_.local_planes - r;
            if (done_lo_) lo = std::min<uint64_t>(lo + M, hi);
            if (done_hi_) hi = std::max<uint64_t>(hi - M, lo);
            FDTDX_ASSERT(fdtdx_propagate_planes(ctx_, output.c_str(), input.c_str(), &prog, lo, hi, stream));
        } else {
            FDTDX_ASSERT(fdtdx_propagate(ctx_, output.c_str(), input.c_str(), &prog, stream));
        }
    }
    // ---- the multi-rank flow of the drivers' loop (fdtd1.cu:70-78) ---------------------------------------------------
    // stencil.h:435-438
    void barrier() { FDTDX_ASSERT(fdtdx_comm_barrier(ctx_)); }
    // The communicator replaces MPI::Init / COMM_WORLD of stencil.h:389-392: rank 0 makes the id, the launcher hands it on.
    static fdtdx_comm_id commUniqueId()
    {
        fdtdx_comm_id id;
        FDTDX_ASSERT(fdtdx_comm_unique_id(&id));
        return id;
    }
    void commInit(const fdtdx_comm_id &id) { FDTDX_ASSERT(fdtdx_comm_init(ctx_, &id)); }
    // all slabs in this process (ranks 0..n-1 in order): copies between them instead of RCCL; two-phase exchanges
    static void commInitLocal(Stencil *const *slabs, int n)
    {
        fdtdx_ctx *h[64];
        for (int i = 0; i < n && i < 64; i++) h[i] = slabs[i]->ctx_;
        FDTDX_ASSERT(fdtdx_comm_init_local(h, n < 64 ? n : 64));
    }
    // backupCubeHaloBackup / restoreCubeHaloBackup (stencil.h:658-682) copied boundary planes into side buffers and the
    // received ghost planes back; here ghost planes live in the cube and are received in place: nothing to do.
    void backupCubeHaloBackup(const std::string &) {}
    void restoreCubeHaloBackup(const std::string &) {}
    // propagateHaloTopBackup / ButtomBackup (stencil.h:770-808): the M planes next to the neighbour below (rank-1: "top",
    // the first planes of the slab) / above.  They are swept straight into the cube, and the point behind them is marked
    // for the exchange, which therefore does not wait for an interior sweep queued after them.
    void propagateHaloTopBackup(const std::string &output, const std::string &input, bool spill, const fdtdx_program &prog, void *stream = nullptr)
    {
        halo_sweep(output, input, spill, prog, stream, 0);
    }
    void propagateHaloButtomBackup(const std::string &output, const std::string &input, bool spill, const fdtdx_program &prog,
                                   void *stream = nullptr)
    {
        halo_sweep(output, input, spill, prog, stream, 1);
    }
    // commCubeHaloBackup (stencil.h:649-656) / commCubeHalo (stencil.h:639-647): boundary planes to the neighbours' ghost
    // planes.  Stream-ordered instead of host-blocking: it returns at once, work queued on `stream` afterwards sees the
    // ghosts, sync() waits for everything.  (Slabs linked with commInitLocal use exchangeStart / exchangeWait in two phases.)
    void commCubeHaloBackup(const std::string &name, void *stream = nullptr)
    {
        FDTDX_ASSERT(fdtdx_comm_cube_halo(ctx_, name.c_str(), stream));
        done_lo_ = done_hi_ = false;
    }
    void commCubeHalo(const std::string &name, void *stream = nullptr) { commCubeHaloBackup(name, stream); }
    void exchangeStart(const std::string &name, void *stream = nullptr)
    {
        FDTDX_ASSERT(fdtdx_comm_exchange_start(ctx_, name.c_str(), stream));
        done_lo_ = done_hi_ = false;
    }
    void exchangeWait(void *stream = nullptr) { FDTDX_ASSERT(fdtdx_comm_exchange_wait(ctx_, stream)); }
    // propagateHalo{Top,Buttom}Backup (stencil.h:770-808): one step on local planes [z_lo, z_hi) only
    void propagatePlanes(const std::string &output, const std::string &input, const fdtdx_program &prog, uint64_t z_lo, uint64_t z_hi,
                         void *stream = nullptr)
    {
        FDTDX_ASSERT(fdtdx_propagate_planes(ctx_, output.c_str(), input.c_str(), &prog, z_lo, z_hi, stream));
    }
    // one fused launch of nsteps = 2 or 3 steps on local planes [z_lo, z_hi) (jacobi.cu:446-454)
    void propagateFusedPlanes(const std::string &output, const std::string &input, const std::string &mid, const fdtdx_program &prog,
                              unsigned nsteps, uint64_t z_lo, uint64_t z_hi, void *stream = nullptr)
    {
        FDTDX_ASSERT(fdtdx_propagate_fused_planes(ctx_, output.c_str(), input.c_str(), mid.c_str(), &prog, nsteps, z_lo, z_hi, stream));
    }
    // ---- main.cu's pipeline: propagate with vel/scal (main.cu:160), address lists (stencil.h:446-466,518-544),
    //      inject (stencil.h:810-815), filt (stencil.h:817-823); plane ranges default to what this rank owns ----
    void propagateComposite(const std::string &output, const std::string &input, const std::string &vel, const fdtdx_composite &comp,
                            void *stream = nullptr)
    {
        FDTDX_ASSERT(fdtdx_propagate_composite(ctx_, output.c_str(), input.c_str(), vel.c_str(), &comp, stream));
    }
    // the four calls of main.cu:160-163 as one, product-accumulate folded into the update (same results)
    void compositeStep(const std::string &output, const std::string &input, const std::string &vel, const std::string &sum,
                       const std::string &addr, const fdtdx_composite &comp, float add, int64_t zsel, int64_t ysel, int64_t xsel, double scale,
                       void *stream = nullptr)
    {
        FDTDX_ASSERT(fdtdx_composite_step(ctx_, output.c_str(), input.c_str(), vel.c_str(), sum.c_str(), addr.c_str(), &comp, add, zsel, ysel,
                                          xsel, scale, stream));
    }
    void mallocAddr(const std::string &name, size_t n) { FDTDX_ASSERT(fdtdx_malloc_addr(ctx_, name.c_str(), n)); }
    void transferAddrToGPU(const std::string &name, const uint64_t *global_addr, size_t n)
    {
        FDTDX_ASSERT(fdtdx_transfer_addr_to_gpu(ctx_, name.c_str(), global_addr, n));
    }
    size_t getAddrSize(const std::string &name)
    {
        uint64_t n = 0;
        FDTDX_ASSERT(fdtdx_addr_size(ctx_, name.c_str(), &n));
        return (size_t)n;
    }
    uint64_t *getAddr(const std::string &name)  // get<size_t>(name) of an address list (stencil.h:479-484)
    {
        uint64_t *p = nullptr;
        FDTDX_ASSERT(fdtdx_get_addr(ctx_, name.c_str(), &p));
        return p;
    }
    // The reference keeps two more lists per address buffer, for replaying inject on its side buffers (stencil.h:486-516,
    // 534-535).  There are no side buffers here, so these lists are empty: an inject over them touches nothing.
    size_t getAddrHaloTopSize(const std::string &) { return 0; }
    size_t getAddrHaloButtomSize(const std::string &) { return 0; }
    void inject(const std::string &cube, const std::string &addr, float add, void *stream = nullptr)
    {
        FDTDX_ASSERT(fdtdx_inject(ctx_, cube.c_str(), addr.c_str(), add, g_.ghost_lo, g_.ghost_lo + g_.owned_planes, stream));
    }
    void filt(const std::string &cube, int64_t zsel, int64_t ysel, int64_t xsel, double scale, unsigned order = FDTDX_ORDER_CHAIN,
              void *stream = nullptr)
    {
        FDTDX_ASSERT(fdtdx_filt_planes(ctx_, cube.c_str(), zsel, ysel, xsel, scale, order, g_.ghost_lo, g_.ghost_lo + g_.owned_planes, stream));
    }
    void mulAcc(const std::string &sum, const std::string &a, const std::string &b, unsigned radius, unsigned order = FDTDX_ORDER_CHAIN,
                void *stream = nullptr)
    {
        const uint64_t lo = g_.ghost_lo ? g_.ghost_lo : radius, hi = g_.local_planes - (g_.ghost_hi ? g_.ghost_hi : radius);
        FDTDX_ASSERT(fdtdx_mul_acc(ctx_, sum.c_str(), a.c_str(), b.c_str(), radius, order, lo, hi, stream));
    }
    // the drivers' whole loop on one rank (fdtd1.cu:68-80): `steps` ping-pong steps, step 0 writes a = f(b)
    void sweep(const std::string &a, const std::string &b, const fdtdx_program &prog, unsigned steps, void *stream = nullptr)
    {
        FDTDX_ASSERT(fdtdx_sweep(ctx_, a.c_str(), b.c_str(), &prog, steps, stream));
    }
    // the same without any allocation inside the library: `work` names the third cube the radius-2 two-step loop hops
    // through ("" = none: such programs run single steps)
    void sweepWork(const std::string &a, const std::string &b, const std::string &work, const fdtdx_program &prog, unsigned steps,
                   void *stream = nullptr)
    {
        FDTDX_ASSERT(fdtdx_sweep_work(ctx_, a.c_str(), b.c_str(), work.empty() ? nullptr : work.c_str(), &prog, steps, stream));
    }
    // fused launch with both shell sources named (in-place two-step loops between two cubes; see fdtdx.h)
    void propagateFusedShells(const std::string &output, const std::string &input, const std::string &shell_even, const std::string &shell_odd,
                              const fdtdx_program &prog, unsigned nsteps, uint64_t z_lo, uint64_t z_hi, void *stream = nullptr)
    {
        FDTDX_ASSERT(fdtdx_propagate_fused_shells(ctx_, output.c_str(), input.c_str(), shell_even.c_str(), shell_odd.c_str(), &prog, nsteps, z_lo,
                                                  z_hi, stream));
    }
    // commCubeHalo (stencil.h:639-647): where the planes to send/receive are; the launcher moves them
    std::pair<DataType *, size_t> haloRegion(const std::string &name, int side, int what)
    {
        float *p = nullptr;
        size_t n = 0;
        FDTDX_ASSERT(fdtdx_halo_region(ctx_, name.c_str(), side, what, &p, &n));
        return {p, n};
    }
    void initMt(const std::string &p0, const std::string &p1) { FDTDX_ASSERT(fdtdx_init_mt(ctx_, p0.c_str(), p1.c_str())); }  // fdtd1.cu:21-31
    void initJacobi(const std::string &name) { FDTDX_ASSERT(fdtdx_init_jacobi(ctx_, name.c_str())); }                      // jacobi.cu:412-417
};

}  // namespace fdtdx
