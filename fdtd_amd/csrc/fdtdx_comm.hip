// fdtdx_comm.hip -- ghost-plane exchange between z neighbours, and the boundary-first time step around it.
//
// Native counterpart of the halo machinery of the reference (stencil.h:639-682: commCubeHalo, commCubeHaloBackup,
// backup/restoreCubeHaloBackup) and of the per-step choreography of its drivers (fdtd1.cu:70-78):
//
//   reference, per step                               here, per step
//   -----------------------------------------------   --------------------------------------------------------------
//   backupCubeHaloBackup      D2D copies 3M planes    --  (ghost planes live inside the cube)
//   propagateHalo{Top,Buttom}Backup   M planes each   boundary sweeps on the caller's stream, event `ready` behind them
//   sync; propagate (async)                           exchange on its own stream once `ready` has fired:
//   commCubeHaloBackup   host-blocking MPI Sendrecv       RCCL  ncclGroupStart; ncclSend/ncclRecv per neighbour; ncclGroupEnd
//   sync; restoreCubeHaloBackup  D2D copies; sync         local peer copies (copy engines) straight into the neighbour's ghosts
//                                                     interior sweep on the caller's stream, beside the exchange
//                                                     the caller's stream waits for event `done`
//
// Boundary planes are contiguous in the cube and are received straight into the neighbour's ghost planes of the same
// cube: no packing, no staging copies.  Two transports:
//   * RCCL (one process per GPU, the layout the reference runs in: one MPI rank per GPU): send/recv pairs over xGMI.
//     librccl is opened at run time, so the rest of the library works where it is absent.
//   * local (all slabs in one process): hipMemcpyPeerAsync / device-to-device copies on the exchange stream.  The
//     host issues the slabs one after the other, so a step has two phases: begin on every slab, then end on every slab.
//   * IPC (one process per GPU, no kernel of the exchange on any CU): every rank maps its neighbours' cubes
//     (hipIpcGetMemHandle / hipIpcOpenMemHandle) and PUSHES its boundary planes into their ghost planes with copies on
//     the exchange stream, beside its interior sweep.  The hosts pace the exchange, as the reference's do with their
//     blocking MPI calls while the kernels run (fdtd1.cu:74-76): a rank's host waits for its own copies to land, says so in
//     a word of shared memory, and waits for its neighbours' word before it queues the next step.  The interior sweep
//     outlasts the copies, so the next step is queued while it still runs and the device does not idle.  The sweeps fill
//     every CU with one workgroup, which is what starves RCCL's send/recv kernels beside them; copies need none.
#include "fdtdx_ctx.h"

#include <dlfcn.h>
#include <fcntl.h>
#include <rccl/rccl.h>
#include <sys/mman.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

namespace {

struct RcclApi {
    void *handle = nullptr;
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclSend) Send = nullptr;
    decltype(&ncclRecv) Recv = nullptr;
    decltype(&ncclBroadcast) Broadcast = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
};

int rccl(RcclApi **out)
{
    static RcclApi api;
    static std::mutex m;
    std::lock_guard<std::mutex> lock(m);
    if (!api.handle) {
        // a copy already loaded by the host program (PyTorch ships one under the same soname) is reused by the loader
        const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
        void *h = nullptr;
        for (const char *n : names)
            if ((h = dlopen(n, RTLD_NOW | RTLD_GLOBAL))) break;
        if (!h) return fail(FDTDX_ERR_UNSUPPORTED, "RCCL not found (librccl.so.1): %s", dlerror());
#define SYM(field, name)                                                                                  \
    if (!(api.field = (decltype(api.field))dlsym(h, name))) {                                             \
        dlclose(h);                                                                                       \
        return fail(FDTDX_ERR_UNSUPPORTED, "RCCL lacks %s", name);                                        \
    }
        SYM(GetUniqueId, "ncclGetUniqueId")
        SYM(CommInitRank, "ncclCommInitRank")
        SYM(CommDestroy, "ncclCommDestroy")
        SYM(GroupStart, "ncclGroupStart")
        SYM(GroupEnd, "ncclGroupEnd")
        SYM(Send, "ncclSend")
        SYM(Recv, "ncclRecv")
        SYM(Broadcast, "ncclBroadcast")
        SYM(GetErrorString, "ncclGetErrorString")
#undef SYM
        api.handle = h;
    }
    *out = &api;
    return FDTDX_OK;
}

#define NCCL_TRY(api, expr)                                                                              \
    do {                                                                                                 \
        ncclResult_t r_ = (expr);                                                                        \
        if (r_ != ncclSuccess) return fail(FDTDX_ERR_HIP, "%s: %s", #expr, (api)->GetErrorString(r_));   \
    } while (0)

}  // namespace

enum { COMM_LOCAL = 0, COMM_RCCL = 1, COMM_IPC = 2 };

// what a rank in IPC mode shares with its neighbours through POSIX shared memory
struct IpcShared {
    std::atomic<uint64_t> landed[2];  // exchanges whose copies towards the neighbour below / above have completed
    std::atomic<uint64_t> arrived;    // barriers this rank has entered
};

struct fdtdx_comm {
    int mode = COMM_LOCAL;
    hipStream_t xs = nullptr;       // the exchange stream
    hipStream_t xs2 = nullptr;      // IPC: the copy towards the neighbour above (the two directions use different links and copy engines)
    hipStream_t bs = nullptr;       // the boundary stream: a step's boundary-plane sweeps, beside its interior sweep
    hipEvent_t fork = nullptr;      // recorded on the caller's stream when a step begins: what the boundary sweeps wait for
    hipEvent_t ready = nullptr;     // recorded on the caller's stream: the boundary planes of this step are written
    hipEvent_t done = nullptr;      // recorded on xs: this context's part of the exchange has finished
    hipEvent_t guard[2] = {nullptr, nullptr};  // local: recorded on THIS context's stream for the neighbour on that side
    bool marked = false;            // `ready` was recorded explicitly (fdtdx_comm_boundary_done) for the next exchange
    bool broken = false;            // an RCCL call failed inside a send/recv group: the communicator's state is unknown, every later exchange fails fast
    uint64_t epoch = 0;             // exchanges started
    uint64_t waited = 0;            // exchanges waited for
    hipStream_t last_stream = nullptr;
    float *token = nullptr;         // two floats of device memory passed along the chain by fdtdx_comm_barrier
    // RCCL
    ncclComm_t comm = nullptr;
    int peer[2] = {-1, -1};         // communicator ranks of the neighbours below / above
    int comm_rank = -1, comm_size = 0;
    // local
    fdtdx_ctx *nb[2] = {nullptr, nullptr};
    // IPC
    char shm_name[64] = "";
    IpcShared *self_sh = nullptr, *peer_sh[2] = {nullptr, nullptr};
    std::map<std::string, float *> peer_cube[2];  // the neighbours' cubes as mapped into this process
    std::vector<void *> opened;                    // what hipIpcOpenMemHandle returned (closed on release)
    fdtdx_geometry peer_g[2];
    uint64_t barriers = 0;
};

// Streams share a few hardware queues, and a wait blocks everything queued behind it in its hardware queue.  The runtime
// hands a stream's commands to the hardware in batches of its own timing, so a wait can reach a hardware queue before the
// commands (of another stream on the same queue) that will produce the event it waits for, and then nothing behind it
// ever runs.  Every event another stream is going to wait on is therefore pushed to the hardware at once, in the order
// the host produced it: a query of the stream submits what the stream holds.
static inline void submit_now(hipStream_t s)
{
    if (hipStreamQuery(s) != hipSuccess) (void)hipGetLastError();  // "not ready" is the expected answer
}

static int comm_streams(fdtdx_ctx *ctx, fdtdx_comm *c)
{
    USE_DEVICE(ctx);
    // An ordinary stream.  (A high-priority one, meant to get the exchange's few workgroups ahead of an interior sweep
    // that fills every CU, left the device waiting for ever once a process had created a handful of streams: the second
    // two-slab run with a stream per slab never finished.  Room for the exchange is made with the "reserve_cus" option.)
    if (ctx->opt_exchange_priority) {  // "exchange_priority" option: the exchange stream at the device's highest priority
        int least = 0, greatest = 0;
        HIP_TRY(hipDeviceGetStreamPriorityRange(&least, &greatest));
        HIP_TRY(hipStreamCreateWithPriority(&c->xs, hipStreamNonBlocking, greatest));
    } else {
        HIP_TRY(hipStreamCreateWithFlags(&c->xs, hipStreamNonBlocking));
    }
    HIP_TRY(hipStreamCreateWithFlags(&c->bs, hipStreamNonBlocking));
    if (c->mode == COMM_IPC) HIP_TRY(hipStreamCreateWithFlags(&c->xs2, hipStreamNonBlocking));
    HIP_TRY(hipEventCreateWithFlags(&c->fork, hipEventDisableTiming));
    HIP_TRY(hipEventCreateWithFlags(&c->ready, hipEventDisableTiming));
    HIP_TRY(hipEventCreateWithFlags(&c->done, hipEventDisableTiming));
    for (int s = 0; s < 2; s++) HIP_TRY(hipEventCreateWithFlags(&c->guard[s], hipEventDisableTiming));
    HIP_TRY(hipMalloc((void **)&c->token, 256));
    return FDTDX_OK;
}

void fdtdx_comm_release(fdtdx_ctx *ctx)
{
    fdtdx_comm *c = ctx->comm;
    if (!c) return;
    (void)hipSetDevice(ctx->device);
    if (c->bs) (void)hipStreamSynchronize(c->bs);
    if (c->xs) (void)hipStreamSynchronize(c->xs);
    if (c->xs2) (void)hipStreamSynchronize(c->xs2);
    if (c->comm) {
        RcclApi *api = nullptr;
        if (rccl(&api) == FDTDX_OK) (void)api->CommDestroy(c->comm);
    }
    if (c->ready) (void)hipEventDestroy(c->ready);
    if (c->done) (void)hipEventDestroy(c->done);
    for (int s = 0; s < 2; s++)
        if (c->guard[s]) (void)hipEventDestroy(c->guard[s]);
    if (c->fork) (void)hipEventDestroy(c->fork);
    if (c->bs) (void)hipStreamDestroy(c->bs);
    if (c->xs) (void)hipStreamDestroy(c->xs);
    if (c->xs2) (void)hipStreamDestroy(c->xs2);
    if (c->token) (void)hipFree(c->token);
    for (void *p : c->opened) (void)hipIpcCloseMemHandle(p);
    for (int s = 0; s < 2; s++)
        if (c->peer_sh[s]) munmap(c->peer_sh[s], sizeof(IpcShared));
    if (c->self_sh) {
        munmap(c->self_sh, sizeof(IpcShared));
        shm_unlink(c->shm_name);
    }
    // a neighbour in this process must not keep pointing here
    for (int s = 0; s < 2; s++)
        if (c->nb[s] && c->nb[s]->comm && c->nb[s]->comm->nb[1 - s] == ctx) c->nb[s]->comm->nb[1 - s] = nullptr;
    delete c;
    ctx->comm = nullptr;
}

// What of this slab's exchange machinery has completed on the device, as text: for finding the wait that never fires when
// a multi-stream run stops making progress.  Queries only; nothing is queued or waited for.
extern "C" int fdtdx_comm_state(fdtdx_ctx *ctx, char *buf, size_t len)
{
    if (!ctx || !buf || !len) return fail(FDTDX_ERR_INVALID, "null argument");
    fdtdx_comm *c = ctx->comm;
    if (!c) {
        snprintf(buf, len, "no communicator");
        return FDTDX_OK;
    }
    USE_DEVICE(ctx);
    auto ev = [](hipEvent_t e) {
        if (!e) return "none";
        const hipError_t r = hipEventQuery(e);
        if (r != hipSuccess) (void)hipGetLastError();
        return r == hipSuccess ? "fired" : r == hipErrorNotReady ? "PENDING" : "error";
    };
    auto st = [](hipStream_t s) {
        const hipError_t r = hipStreamQuery(s);
        if (r != hipSuccess) (void)hipGetLastError();
        return r == hipSuccess ? "idle" : r == hipErrorNotReady ? "BUSY" : "error";
    };
    snprintf(buf, len, "rank %u mode=%s epoch=%llu waited=%llu | events: fork=%s ready=%s done=%s guard_lo=%s guard_hi=%s | streams: exchange=%s boundary=%s caller(last)=%s",
             ctx->cfg.rank, c->mode == COMM_RCCL ? "rccl" : c->mode == COMM_IPC ? "ipc" : "local", (unsigned long long)c->epoch, (unsigned long long)c->waited, ev(c->fork), ev(c->ready),
             ev(c->done), ev(c->guard[0]), ev(c->guard[1]), st(c->xs), st(c->bs), st(c->last_stream));
    return FDTDX_OK;
}

extern "C" int fdtdx_comm_destroy(fdtdx_ctx *ctx)
{
    if (!ctx) return fail(FDTDX_ERR_INVALID, "null context");
    fdtdx_comm_release(ctx);
    return FDTDX_OK;
}

extern "C" int fdtdx_comm_unique_id(fdtdx_comm_id *id)
{
    if (!id) return fail(FDTDX_ERR_INVALID, "null argument");
    static_assert(sizeof(fdtdx_comm_id) == sizeof(ncclUniqueId), "fdtdx_comm_id carries an ncclUniqueId");
    RcclApi *api;
    int rc = rccl(&api);
    if (rc) return rc;
    ncclUniqueId u;
    NCCL_TRY(api, api->GetUniqueId(&u));
    memcpy(id->bytes, &u, sizeof(u));
    return FDTDX_OK;
}

extern "C" int fdtdx_comm_init_ranks(fdtdx_ctx *ctx, const fdtdx_comm_id *id, int comm_rank, int comm_size, int peer_lo, int peer_hi)
{
    if (!ctx || !id) return fail(FDTDX_ERR_INVALID, "null argument");
    if (ctx->comm) return fail(FDTDX_ERR_EXISTS, "the context already has a communicator");
    if (comm_rank < 0 || comm_rank >= comm_size) return fail(FDTDX_ERR_INVALID, "communicator rank %d of %d", comm_rank, comm_size);
    const fdtdx_geometry &g = ctx->g;
    if ((g.ghost_lo && (peer_lo < 0 || peer_lo >= comm_size)) || (g.ghost_hi && (peer_hi < 0 || peer_hi >= comm_size)))
        return fail(FDTDX_ERR_INVALID, "neighbour ranks %d / %d outside the communicator of %d", peer_lo, peer_hi, comm_size);
    RcclApi *api;
    int rc = rccl(&api);
    if (rc) return rc;
    fdtdx_comm *c = new (std::nothrow) fdtdx_comm;
    if (!c) return fail(FDTDX_ERR_NOMEM, "out of host memory");
    c->mode = COMM_RCCL;
    c->peer[0] = g.ghost_lo ? peer_lo : -1, c->peer[1] = g.ghost_hi ? peer_hi : -1;
    c->comm_rank = comm_rank, c->comm_size = comm_size;
    ctx->comm = c;
    if ((rc = comm_streams(ctx, c))) {
        fdtdx_comm_release(ctx);
        return rc;
    }
    ncclUniqueId u;
    memcpy(&u, id->bytes, sizeof(u));
    const ncclResult_t r = api->CommInitRank(&c->comm, comm_size, u, comm_rank);
    if (r != ncclSuccess) {
        c->comm = nullptr;
        fdtdx_comm_release(ctx);
        return fail(FDTDX_ERR_HIP, "ncclCommInitRank: %s", api->GetErrorString(r));
    }
    return FDTDX_OK;
}

extern "C" int fdtdx_comm_init(fdtdx_ctx *ctx, const fdtdx_comm_id *id)
{
    if (!ctx) return fail(FDTDX_ERR_INVALID, "null argument");
    const int r = (int)ctx->cfg.rank, n = (int)ctx->cfg.nprocs;
    return fdtdx_comm_init_ranks(ctx, id, r, n, r - 1, r + 1);
}

extern "C" int fdtdx_comm_init_local(fdtdx_ctx *const *ctxs, int n)
{
    if (!ctxs || n < 1) return fail(FDTDX_ERR_INVALID, "null argument");
    for (int i = 0; i < n; i++) {
        if (!ctxs[i]) return fail(FDTDX_ERR_INVALID, "null context");
        if (ctxs[i]->comm) return fail(FDTDX_ERR_EXISTS, "slab %d already has a communicator", i);
        const fdtdx_config &a = ctxs[i]->cfg, &b = ctxs[0]->cfg;
        if ((int)a.rank != i || (int)a.nprocs != n || a.Z != b.Z || a.Y != b.Y || a.X != b.X || a.M != b.M || ctxs[i]->g.pitch_x != ctxs[0]->g.pitch_x)
            return fail(FDTDX_ERR_INVALID, "slab %d: the contexts must be ranks 0..%d of one %d-way split of one grid, in order", i, n - 1, n);
    }
    for (int i = 0; i < n; i++) {
        fdtdx_comm *c = new (std::nothrow) fdtdx_comm;
        if (!c) return fail(FDTDX_ERR_NOMEM, "out of host memory");
        c->mode = COMM_LOCAL;
        ctxs[i]->comm = c;
        int rc = comm_streams(ctxs[i], c);
        if (rc) {
            for (int k = 0; k <= i; k++) fdtdx_comm_release(ctxs[k]);
            return rc;
        }
    }
    for (int i = 0; i < n; i++) {
        ctxs[i]->comm->nb[0] = i > 0 ? ctxs[i - 1] : nullptr;
        ctxs[i]->comm->nb[1] = i < n - 1 ? ctxs[i + 1] : nullptr;
        for (int s = 0; s < 2; s++) {  // let the copy engines reach the neighbour's memory directly
            fdtdx_ctx *o = ctxs[i]->comm->nb[s];
            if (o && o->device != ctxs[i]->device) {
                (void)hipSetDevice(ctxs[i]->device);
                int can = 0;
                if (hipDeviceCanAccessPeer(&can, ctxs[i]->device, o->device) == hipSuccess && can) {
                    const hipError_t e = hipDeviceEnablePeerAccess(o->device, 0);
                    if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) (void)hipGetLastError();
                }
            }
        }
    }
    return FDTDX_OK;
}

// ---- IPC transport: set-up ---------------------------------------------------------------------------------------
static_assert(sizeof(hipIpcMemHandle_t) <= sizeof(((fdtdx_ipc_desc *)nullptr)->cube[0].handle), "fdtdx_ipc_desc carries a hipIpcMemHandle_t per cube");
static_assert(sizeof(fdtdx_ipc_desc) == 976, "launchers pass descriptors around as 976 plain bytes");

extern "C" int fdtdx_comm_ipc_export(fdtdx_ctx *ctx, const char *const *names, int n, fdtdx_ipc_desc *out)
{
    if (!ctx || !names || !out || n < 1) return fail(FDTDX_ERR_INVALID, "null argument");
    if (n > FDTDX_IPC_MAX_CUBES) return fail(FDTDX_ERR_UNSUPPORTED, "at most %d cubes per descriptor", FDTDX_IPC_MAX_CUBES);
    if (ctx->comm) return fail(FDTDX_ERR_EXISTS, "the context already has a communicator");
    memset(out, 0, sizeof(*out));
    fdtdx_comm *c = new (std::nothrow) fdtdx_comm;
    if (!c) return fail(FDTDX_ERR_NOMEM, "out of host memory");
    c->mode = COMM_IPC;
    ctx->comm = c;
    int rc = comm_streams(ctx, c);
    if (rc) {
        fdtdx_comm_release(ctx);
        return rc;
    }
    static std::atomic<unsigned> serial{0};
    snprintf(c->shm_name, sizeof(c->shm_name), "/fdtdx_%d_%u_%u", (int)getpid(), ctx->cfg.rank, serial++);
    const int fd = shm_open(c->shm_name, O_CREAT | O_EXCL | O_RDWR, 0600);
    if (fd < 0 || ftruncate(fd, sizeof(IpcShared)) != 0) {
        if (fd >= 0) close(fd), shm_unlink(c->shm_name);
        c->shm_name[0] = 0;
        fdtdx_comm_release(ctx);
        return fail(FDTDX_ERR_HIP, "shared memory segment for the exchange could not be created");
    }
    void *m = mmap(nullptr, sizeof(IpcShared), PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (m == MAP_FAILED) {
        shm_unlink(c->shm_name);
        fdtdx_comm_release(ctx);
        return fail(FDTDX_ERR_HIP, "mmap of the exchange's shared memory failed");
    }
    c->self_sh = new (m) IpcShared;
    c->self_sh->landed[0] = 0, c->self_sh->landed[1] = 0, c->self_sh->arrived = 0;
    out->rank = ctx->cfg.rank, out->ncubes = (uint32_t)n, out->pid = (int32_t)getpid();
    memcpy(out->shm_name, c->shm_name, sizeof(out->shm_name));
    if (hipSetDevice(ctx->device) != hipSuccess) {
        fdtdx_comm_release(ctx);
        return fail(FDTDX_ERR_HIP, "hipSetDevice(%d) failed", ctx->device);
    }
    for (int i = 0; i < n; i++) {
        float *p;
        if ((rc = fdtdx_find_cube(ctx, names[i], &p))) {
            fdtdx_comm_release(ctx);
            return rc;
        }
        if (strlen(names[i]) >= sizeof(out->cube[i].name)) {
            fdtdx_comm_release(ctx);
            return fail(FDTDX_ERR_UNSUPPORTED, "cube name '%s' too long for a descriptor", names[i]);
        }
        // the handle names the whole allocation the cube lies in (a cube attached from a caching allocator is a piece of one)
        hipDeviceptr_t base = nullptr;
        size_t span = 0;
        hipIpcMemHandle_t h;
        hipError_t e = hipMemGetAddressRange(&base, &span, (hipDeviceptr_t)p);
        if (e == hipSuccess) e = hipIpcGetMemHandle(&h, (void *)base);
        if (e != hipSuccess) {
            fdtdx_comm_release(ctx);
            return fail(FDTDX_ERR_HIP, "no inter-process handle for cube '%s': %s", names[i], hipGetErrorString(e));
        }
        strcpy(out->cube[i].name, names[i]);
        memcpy(out->cube[i].handle, &h, sizeof(h));
        out->cube[i].offset = (uint64_t)((char *)p - (char *)base);
        out->cube[i].bytes = ctx->g.cube_elems * sizeof(float);
    }
    return FDTDX_OK;
}

extern "C" int fdtdx_comm_ipc_connect(fdtdx_ctx *ctx, const fdtdx_ipc_desc *lo, const fdtdx_ipc_desc *hi)
{
    if (!ctx || !ctx->comm || ctx->comm->mode != COMM_IPC) return fail(FDTDX_ERR_INVALID, "call fdtdx_comm_ipc_export on this context first");
    fdtdx_comm *c = ctx->comm;
    const fdtdx_ipc_desc *d[2] = {lo, hi};
    const bool need[2] = {ctx->g.ghost_lo != 0, ctx->g.ghost_hi != 0};
    USE_DEVICE(ctx);
    for (int s = 0; s < 2; s++) {
        if (!need[s]) continue;
        if (!d[s]) return fail(FDTDX_ERR_INVALID, "no descriptor of the neighbour %s", s ? "above" : "below");
        const uint32_t want = s ? ctx->cfg.rank + 1 : ctx->cfg.rank - 1;
        if (d[s]->rank != want || d[s]->ncubes > FDTDX_IPC_MAX_CUBES) return fail(FDTDX_ERR_INVALID, "descriptor of rank %u where rank %u's is needed", d[s]->rank, want);
        if (c->peer_sh[s]) return fail(FDTDX_ERR_EXISTS, "already connected");
        fdtdx_config cf = ctx->cfg;
        cf.rank = want;
        int rc = fdtdx_decompose(&cf, &c->peer_g[s]);
        if (rc) return rc;
        char name[sizeof(d[s]->shm_name) + 1] = "";
        memcpy(name, d[s]->shm_name, sizeof(d[s]->shm_name));
        const int fd = shm_open(name, O_RDWR, 0600);
        if (fd < 0) return fail(FDTDX_ERR_HIP, "cannot open the neighbour's shared memory '%s'", name);
        void *m = mmap(nullptr, sizeof(IpcShared), PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
        close(fd);
        if (m == MAP_FAILED) return fail(FDTDX_ERR_HIP, "mmap of the neighbour's shared memory failed");
        c->peer_sh[s] = (IpcShared *)m;
        for (uint32_t i = 0; i < d[s]->ncubes; i++) {
            if (d[s]->cube[i].bytes != c->peer_g[s].cube_elems * sizeof(float)) return fail(FDTDX_ERR_INVALID, "the neighbour's cube '%.31s' has another geometry", d[s]->cube[i].name);
            hipIpcMemHandle_t h;
            memcpy(&h, d[s]->cube[i].handle, sizeof(h));
            void *base = nullptr;
            const hipError_t e = hipIpcOpenMemHandle(&base, h, hipIpcMemLazyEnablePeerAccess);
            if (e != hipSuccess) return fail(FDTDX_ERR_HIP, "hipIpcOpenMemHandle for the neighbour's cube '%.31s': %s", d[s]->cube[i].name, hipGetErrorString(e));
            c->opened.push_back(base);
            c->peer_cube[s][std::string(d[s]->cube[i].name, strnlen(d[s]->cube[i].name, sizeof(d[s]->cube[i].name)))] = (float *)((char *)base + d[s]->cube[i].offset);
        }
    }
    return FDTDX_OK;
}

// host side of an IPC-mode wait: until `word` reaches `want` (the neighbour's host says so); false after `limit_s` seconds
static bool ipc_await(const std::atomic<uint64_t> &word, uint64_t want, double limit_s)
{
    const auto t0 = std::chrono::steady_clock::now();
    for (unsigned spins = 0; word.load(std::memory_order_acquire) < want; spins++) {
        if ((spins & 0x3ff) == 0x3ff) {
            if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > limit_s) return false;
            std::this_thread::yield();
        }
    }
    return true;
}

extern "C" int fdtdx_comm_boundary_done(fdtdx_ctx *ctx, void *stream)
{
    if (!ctx || !ctx->comm) return fail(FDTDX_ERR_INVALID, "the context has no communicator");
    USE_DEVICE(ctx);
    HIP_TRY(hipEventRecord(ctx->comm->ready, (hipStream_t)stream));
    submit_now((hipStream_t)stream);
    ctx->comm->marked = true;
    return FDTDX_OK;
}

extern "C" int fdtdx_comm_exchange_start(fdtdx_ctx *ctx, const char *name, void *stream)
{
    if (!ctx || !name) return fail(FDTDX_ERR_INVALID, "null argument");
    fdtdx_comm *c = ctx->comm;
    if (!c) return ctx->cfg.nprocs == 1 ? FDTDX_OK : fail(FDTDX_ERR_INVALID, "no communicator: call fdtdx_comm_init or fdtdx_comm_init_local first");
    if (c->broken) return fail(FDTDX_ERR_HIP, "an earlier RCCL call of this communicator failed: destroy it (fdtdx_comm_destroy) and set up a new one");
    if (c->epoch != c->waited) return fail(FDTDX_ERR_INVALID, "the previous exchange has not been waited for (fdtdx_comm_exchange_wait)");
    float *snd[2], *rcv[2];
    size_t cnt[2], cnt2;
    int rc;
    for (int s = 0; s < 2; s++) {
        if ((rc = fdtdx_halo_region(ctx, name, s, 0, &snd[s], &cnt[s])) || (rc = fdtdx_halo_region(ctx, name, s, 1, &rcv[s], &cnt2))) return rc;
    }
    USE_DEVICE(ctx);
    hipStream_t st = (hipStream_t)stream;
    if (!c->marked) {
        HIP_TRY(hipEventRecord(c->ready, st));  // everything queued so far, the boundary sweeps included
        submit_now(st);
    }
    c->marked = false;
    HIP_TRY(hipStreamWaitEvent(c->xs, c->ready, 0));
    if (c->xs2) HIP_TRY(hipStreamWaitEvent(c->xs2, c->ready, 0));
    if (c->mode == COMM_RCCL) {
        RcclApi *api;
        if ((rc = rccl(&api))) return rc;
        // a failure between GroupStart and GroupEnd must not leave the thread's group open (every later RCCL call of this
        // thread, the host program's included, would queue into it and never launch): always close it, keep the first error
        ncclResult_t first = api->GroupStart();
        if (first == ncclSuccess) {
            for (int s = 0; s < 2 && first == ncclSuccess; s++)
                if (cnt[s]) {
                    first = api->Send(snd[s], cnt[s], ncclFloat, c->peer[s], c->comm, c->xs);
                    if (first == ncclSuccess) first = api->Recv(rcv[s], cnt[s], ncclFloat, c->peer[s], c->comm, c->xs);
                }
            const ncclResult_t end = api->GroupEnd();
            if (first == ncclSuccess) first = end;
        }
        if (first != ncclSuccess) {
            c->broken = true;
            return fail(FDTDX_ERR_HIP, "RCCL send/recv group of the ghost-plane exchange: %s", api->GetErrorString(first));
        }
    } else if (c->mode == COMM_IPC) {
        // push: this rank's boundary planes into the neighbour's ghost planes on the side that faces this rank.  What the
        // copy overwrites there the neighbour last read in the boundary sweep of its previous step, and that sweep had
        // finished before the neighbour pushed the planes this rank's own boundary sweep (queued ahead of this copy) has
        // just read: nothing more to wait for.  One stream per direction: the two copies leave over different links.
        for (int s = 0; s < 2; s++) {
            if (!cnt[s]) continue;
            auto it = c->peer_cube[s].find(name);
            if (!c->peer_sh[s] || it == c->peer_cube[s].end()) return fail(FDTDX_ERR_INVALID, "cube '%s' of the neighbour %s is not mapped (fdtdx_comm_ipc_export / _connect)", name, s ? "above" : "below");
            const fdtdx_geometry &pg = c->peer_g[s];
            const uint64_t depth = s ? pg.ghost_lo : pg.ghost_hi;  // its ghost planes towards this rank
            if (depth * pg.plane != cnt[s]) return fail(FDTDX_ERR_INVALID, "the neighbouring slab has another geometry");
            float *dst = it->second + (s ? 0 : (pg.local_planes - pg.ghost_hi) * pg.plane);
            HIP_TRY(hipMemcpyAsync(dst, snd[s], cnt[s] * sizeof(float), hipMemcpyDeviceToDevice, s ? c->xs2 : c->xs));
        }
        submit_now(c->xs2);
    } else {
        for (int s = 0; s < 2; s++) {
            fdtdx_ctx *o = c->nb[s];
            if (!cnt[s]) continue;
            if (!o || !o->comm) return fail(FDTDX_ERR_INVALID, "the neighbouring slab is gone");
            // The copy lands in the neighbour's ghost planes of this cube, which the neighbour's sweeps queued so far may
            // still read (as the input of its previous step): wait for all it has queued.
            // (always: a neighbour that has not exchanged yet may have queued sweeps all the same, on the default stream too,
            // which `last_stream == nullptr` stands for, and the exchange stream does not synchronise with that stream by itself)
            {
                HIP_TRY(hipSetDevice(o->device));
                HIP_TRY(hipEventRecord(o->comm->guard[1 - s], o->comm->last_stream));
                submit_now(o->comm->last_stream);
                HIP_TRY(hipSetDevice(ctx->device));
                HIP_TRY(hipStreamWaitEvent(c->xs, o->comm->guard[1 - s], 0));
            }
            float *dst;
            size_t dcnt;
            if ((rc = fdtdx_halo_region(o, name, 1 - s, 1, &dst, &dcnt))) return rc;
            if (dcnt != cnt[s]) return fail(FDTDX_ERR_INVALID, "the neighbouring slab has another geometry");
            USE_DEVICE(ctx);
            if (o->device == ctx->device)
                HIP_TRY(hipMemcpyAsync(dst, snd[s], cnt[s] * sizeof(float), hipMemcpyDeviceToDevice, c->xs));
            else
                HIP_TRY(hipMemcpyPeerAsync(dst, o->device, snd[s], ctx->device, cnt[s] * sizeof(float), c->xs));
        }
    }
    HIP_TRY(hipEventRecord(c->done, c->xs));
    submit_now(c->xs);
    c->last_stream = st;
    c->epoch++;
    return FDTDX_OK;
}

extern "C" int fdtdx_comm_exchange_wait(fdtdx_ctx *ctx, void *stream)
{
    if (!ctx) return fail(FDTDX_ERR_INVALID, "null argument");
    fdtdx_comm *c = ctx->comm;
    if (!c) return ctx->cfg.nprocs == 1 ? FDTDX_OK : fail(FDTDX_ERR_INVALID, "no communicator");
    if (c->waited == c->epoch) return FDTDX_OK;  // nothing in flight
    USE_DEVICE(ctx);
    hipStream_t st = (hipStream_t)stream;
    if (c->mode == COMM_IPC) {
        // the hosts pace the exchange: this rank's copies have landed (it says so), then its neighbours' (they say so);
        // what the caller queues after this call is queued after both
        if (c->broken) return fail(FDTDX_ERR_HIP, "an earlier exchange of this communicator failed");
        for (int s = 0; s < 2; s++) {
            HIP_TRY(hipStreamSynchronize(s ? c->xs2 : c->xs));
            if (c->peer_sh[s]) c->self_sh->landed[s].store(c->epoch, std::memory_order_release);
        }
        for (int s = 0; s < 2; s++)
            if (c->peer_sh[s] && !ipc_await(c->peer_sh[s]->landed[1 - s], c->epoch, 60.0)) {
                c->broken = true;
                return fail(FDTDX_ERR_HIP, "the neighbour %s has not delivered exchange %llu within 60 s", s ? "above" : "below", (unsigned long long)c->epoch);
            }
        c->waited = c->epoch;
        c->last_stream = st;
        return FDTDX_OK;
    }
    HIP_TRY(hipStreamWaitEvent(st, c->done, 0));  // RCCL: sends and receives; local: the copies out of this cube
    if (c->mode == COMM_LOCAL)
        for (int s = 0; s < 2; s++) {  // the copies INTO this cube are the neighbours' work
            fdtdx_ctx *o = c->nb[s];
            if (!o || !o->comm) continue;
            if (o->comm->epoch != c->epoch)
                return fail(FDTDX_ERR_INVALID, "slabs in one process exchange in two phases: start the exchange on every slab, then wait on every slab");
            HIP_TRY(hipStreamWaitEvent(st, o->comm->done, 0));
        }
    c->waited = c->epoch;
    c->last_stream = st;
    return FDTDX_OK;
}

extern "C" int fdtdx_comm_cube_halo(fdtdx_ctx *ctx, const char *name, void *stream)
{
    if (ctx && ctx->comm && ctx->comm->mode == COMM_LOCAL && ctx->cfg.nprocs > 1)
        return fail(FDTDX_ERR_UNSUPPORTED, "slabs in one process exchange in two phases: fdtdx_comm_exchange_start on every slab, then fdtdx_comm_exchange_wait");
    int rc = fdtdx_comm_exchange_start(ctx, name, stream);
    return rc ? rc : fdtdx_comm_exchange_wait(ctx, stream);
}

// Stencil::barrier (stencil.h:435-438, MPI Barrier): every slab's queued work has finished and every rank has arrived.
// RCCL: a token travels up the chain of slabs and back down (each rank receives from below, sends up, receives from
// above, sends down), then the exchange stream is drained; local: the devices of this slab and its neighbours are drained.
extern "C" int fdtdx_comm_barrier(fdtdx_ctx *ctx)
{
    if (!ctx) return fail(FDTDX_ERR_INVALID, "null context");
    USE_DEVICE(ctx);
    HIP_TRY(hipDeviceSynchronize());
    fdtdx_comm *c = ctx->comm;
    if (!c) return ctx->cfg.nprocs == 1 ? FDTDX_OK : fail(FDTDX_ERR_INVALID, "no communicator");
    if (c->mode == COMM_LOCAL) {
        for (int s = 0; s < 2; s++)
            for (fdtdx_ctx *o = c->nb[s]; o; o = o->comm ? o->comm->nb[s] : nullptr) {
                HIP_TRY(hipSetDevice(o->device));
                HIP_TRY(hipDeviceSynchronize());
            }
        HIP_TRY(hipSetDevice(ctx->device));
        return FDTDX_OK;
    }
    if (c->mode == COMM_IPC) {  // this rank and both its neighbours have drained their devices and arrived
        const uint64_t n = ++c->barriers;
        c->self_sh->arrived.store(n, std::memory_order_release);
        for (int s = 0; s < 2; s++)
            if (c->peer_sh[s] && !ipc_await(c->peer_sh[s]->arrived, n, 120.0)) return fail(FDTDX_ERR_HIP, "the neighbour %s has not reached barrier %llu within 120 s", s ? "above" : "below", (unsigned long long)n);
        return FDTDX_OK;
    }
    RcclApi *api;
    int rc = rccl(&api);
    if (rc) return rc;
    if (c->peer[0] >= 0) NCCL_TRY(api, api->Recv(c->token, 1, ncclFloat, c->peer[0], c->comm, c->xs));
    if (c->peer[1] >= 0) {
        NCCL_TRY(api, api->Send(c->token, 1, ncclFloat, c->peer[1], c->comm, c->xs));
        NCCL_TRY(api, api->Recv(c->token + 1, 1, ncclFloat, c->peer[1], c->comm, c->xs));
    }
    if (c->peer[0] >= 0) NCCL_TRY(api, api->Send(c->token + 1, 1, ncclFloat, c->peer[0], c->comm, c->xs));
    HIP_TRY(hipStreamSynchronize(c->xs));
    return FDTDX_OK;
}

// transferCubeToCPU's gather (stencil.h:623-636: Recv of every rank's slab on rank 0, Send elsewhere, then a chunked
// Bcast): every rank ends with the whole unpadded [Z][Y][X] grid in `host`.  Here every rank's owned planes travel once,
// device to device (one ncclBroadcast per slab into a staging buffer, no host-side Send/Recv/Bcast), and each rank unpads
// what arrives straight into its place in `host`.  Slabs of one process: the chain of slabs is walked and each slab's
// planes are copied out by its own device.  One rank: the plain download.
extern "C" int fdtdx_gather_cube(fdtdx_ctx *ctx, float *host, const char *name)
{
    if (!ctx || !host || !name) return fail(FDTDX_ERR_INVALID, "null argument");
    const uint64_t plane_host = (uint64_t)ctx->cfg.Y * ctx->cfg.X;
    auto own_part = [&](fdtdx_ctx *o) -> int {
        float *d;
        int rc = fdtdx_find_cube(o, name, &d);
        if (rc) return rc;
        const fdtdx_geometry &g = o->g;
        return fdtdx_copy_planes_d2h(o, host + g.owned_begin * plane_host, d + g.ghost_lo * g.plane, g.owned_planes);
    };
    fdtdx_comm *c = ctx->comm;
    if (!c) return ctx->cfg.nprocs == 1 ? own_part(ctx) : fail(FDTDX_ERR_INVALID, "no communicator: call fdtdx_comm_init or fdtdx_comm_init_local first");
    if (c->mode == COMM_LOCAL) {
        int rc = own_part(ctx);
        for (int s = 0; s < 2 && !rc; s++)
            for (fdtdx_ctx *o = c->nb[s]; o && !rc; o = o->comm ? o->comm->nb[s] : nullptr) rc = own_part(o);
        if (hipSetDevice(ctx->device) != hipSuccess) (void)hipGetLastError();
        return rc;
    }
    if (c->mode == COMM_IPC) return fail(FDTDX_ERR_UNSUPPORTED, "fdtdx_gather_cube: ranks in IPC mode map their neighbours only; gather with the launcher's communicator");
    // RCCL: the communicator must be the one fdtdx_comm_init makes (communicator rank == slab rank)
    const int n = (int)ctx->cfg.nprocs;
    if (c->comm_size != n || c->comm_rank != (int)ctx->cfg.rank)
        return fail(FDTDX_ERR_UNSUPPORTED, "fdtdx_gather_cube needs a communicator whose ranks are the slab ranks (fdtdx_comm_init)");
    if (c->broken) return fail(FDTDX_ERR_HIP, "an earlier RCCL call of this communicator failed");
    RcclApi *api;
    int rc = rccl(&api);
    if (rc) return rc;
    float *mine;
    if ((rc = fdtdx_find_cube(ctx, name, &mine))) return rc;
    USE_DEVICE(ctx);
    HIP_TRY(hipDeviceSynchronize());  // stencil.h:615: everything queued has finished before the cube is read
    uint64_t most = 0;
    for (int r = 0; r < n; r++) {
        fdtdx_config cf = ctx->cfg;
        cf.rank = (uint32_t)r;
        fdtdx_geometry gr;
        if ((rc = fdtdx_decompose(&cf, &gr))) return rc;
        most = std::max<uint64_t>(most, gr.owned_planes);
    }
    float *stage = nullptr;
    HIP_TRY(hipMalloc((void **)&stage, most * ctx->g.plane * sizeof(float)));
    for (int r = 0; r < n && !rc; r++) {
        fdtdx_config cf = ctx->cfg;
        cf.rank = (uint32_t)r;
        fdtdx_geometry gr;
        if ((rc = fdtdx_decompose(&cf, &gr))) break;
        const size_t cnt = (size_t)(gr.owned_planes * gr.plane);
        const float *src = (r == c->comm_rank) ? mine + ctx->g.ghost_lo * ctx->g.plane : stage;
        const ncclResult_t e = api->Broadcast(src, stage, cnt, ncclFloat, r, c->comm, c->xs);
        if (e != ncclSuccess) {
            c->broken = true;
            rc = fail(FDTDX_ERR_HIP, "ncclBroadcast of slab %d: %s", r, api->GetErrorString(e));
            break;
        }
        if (hipStreamSynchronize(c->xs) != hipSuccess) {
            rc = fail(FDTDX_ERR_HIP, "hipStreamSynchronize after the broadcast of slab %d failed", r);
            break;
        }
        rc = fdtdx_copy_planes_d2h(ctx, host + gr.owned_begin * plane_host, stage, gr.owned_planes);
    }
    if (hipFree(stage) != hipSuccess && !rc) rc = fail(FDTDX_ERR_HIP, "hipFree(staging) failed");
    return rc;
}

// ---- the time step of a multi-rank driver (fdtd1.cu:70-78) ------------------------------------------------------
static void update_range(const fdtdx_ctx *ctx, uint64_t r, uint64_t *lo, uint64_t *hi)
{
    const fdtdx_geometry &g = ctx->g;
    *lo = g.ghost_lo ? g.ghost_lo : r;
    *hi = g.ghost_hi ? g.local_planes - g.ghost_hi : g.local_planes - r;
}

// One step of a slab with neighbours.  The planes a neighbour needs (the G owned planes next to it, both sides in one
// launch) are swept first, the exchange of `out` starts behind them, and the interior sweep runs beside the exchange: it
// reads owned planes only.  With the "boundary_stream" option the boundary sweep goes to a stream of its own as well
// (it waits for what the caller's stream holds when the step begins, the exchange waits for it, and the caller's stream
// waits for the exchange when the step ends, which covers it).  That is off by default: an interior sweep that has
// started first holds every CU (its workgroups leave no room for a second one), so the boundary sweep and the exchange
// behind it finish only after it, and the next step waits for them -- measured 3-8 % slower per step than sweeping the
// boundary planes ahead of the interior in the caller's stream.
template <class Sweep>
static int split_step(fdtdx_ctx *ctx, const char *out, uint64_t r, void *stream, Sweep &&sweep)
{
    const fdtdx_geometry &g = ctx->g;
    const uint64_t G = ctx->cfg.M;
    uint64_t lo, hi;
    update_range(ctx, r, &lo, &hi);
    uint64_t ilo = lo, ihi = hi;
    int rc;
    fdtdx_comm *c = ctx->comm;
    void *bstream = stream;
    if (c && (g.ghost_lo || g.ghost_hi) && ctx->opt_boundary_stream) {
        USE_DEVICE(ctx);
        HIP_TRY(hipEventRecord(c->fork, (hipStream_t)stream));
        submit_now((hipStream_t)stream);
        HIP_TRY(hipStreamWaitEvent(c->bs, c->fork, 0));
        bstream = c->bs;
    }
    if (g.ghost_lo) ilo = std::min(lo + G, hi);
    if (g.ghost_hi) ihi = std::max(hi - G, ilo);
    // the planes next to the neighbours, both sides in ONE launch (two small launches in a row would each leave most of the device idle)
    if ((rc = sweep(lo, ilo, ihi, hi, bstream))) return rc;
    if ((rc = fdtdx_comm_exchange_start(ctx, out, bstream))) return rc;
    if (c) c->last_stream = (hipStream_t)stream;  // the stream that stands for this slab's queued work: it joins the others when the step ends
    return sweep(ilo, ihi, ihi, ihi, stream);
}

static int program_radius(const fdtdx_program *prog, uint64_t *r)
{
    if (!prog) return fail(FDTDX_ERR_INVALID, "null program");
    *r = prog->kind == FDTDX_STAR ? prog->radius : 1;
    return FDTDX_OK;
}

extern "C" int fdtdx_step_begin(fdtdx_ctx *ctx, const char *out, const char *in, const fdtdx_program *prog, void *stream)
{
    if (!ctx) return fail(FDTDX_ERR_INVALID, "null context");
    uint64_t r;
    int rc = program_radius(prog, &r);
    if (rc) return rc;
    return split_step(ctx, out, r, stream, [&](uint64_t z0, uint64_t z1, uint64_t z2, uint64_t z3, void *s) {
        return (z1 > z0 || z3 > z2) ? fdtdx_propagate_two_ranges(ctx, out, in, prog, z0, z1, z2, z3, s) : FDTDX_OK;
    });
}

extern "C" int fdtdx_hop_begin(fdtdx_ctx *ctx, const char *out, const char *in, const char *shell_even, const char *shell_odd,
                               const fdtdx_program *prog, uint32_t nsteps, void *stream)
{
    if (!ctx) return fail(FDTDX_ERR_INVALID, "null context");
    uint64_t r;
    int rc = program_radius(prog, &r);
    if (rc) return rc;
    return split_step(ctx, out, r, stream, [&](uint64_t z0, uint64_t z1, uint64_t z2, uint64_t z3, void *s) {
        return (z1 > z0 || z3 > z2) ? fdtdx_propagate_fused_two_ranges(ctx, out, in, shell_even, shell_odd, prog, nsteps, z0, z1, z2, z3, s)
                                    : FDTDX_OK;
    });
}

extern "C" int fdtdx_step_end(fdtdx_ctx *ctx, void *stream) { return fdtdx_comm_exchange_wait(ctx, stream); }

// ---- extension (no counterpart in the reference): one half step of the six-field E/H update on a slab with neighbours --
// The H half differences Ex, Ey forwards along z and the E half differences Hx, Hy backwards, so what a neighbour needs of a
// half step's result is ONE plane of TWO fields, in ONE direction: Hx, Hy of the last owned plane go up, Ex, Ey of the first
// owned plane go down.  That plane is updated first, the two copies (or sends) start behind it on the exchange stream and
// the remaining planes are updated beside them; fdtdx_step_end makes the caller's stream wait for the planes received.
// Ez and Hz ghost planes are never read and stay as they are.
extern "C" int fdtdx_yee3d_slab_half_begin(fdtdx_ctx *ctx, const char *const fields[6], int which, float coef, uint32_t order, void *stream)
{
    if (!ctx || !fields) return fail(FDTDX_ERR_INVALID, "null argument");
    if (which != 0 && which != 1) return fail(FDTDX_ERR_INVALID, "which must be 0 (H half step) or 1 (E half step)");
    const fdtdx_geometry &g = ctx->g;
    if (g.ghost_lo > 1 || g.ghost_hi > 1) return fail(FDTDX_ERR_UNSUPPORTED, "the E/H step runs on slabs with one ghost plane per neighbour (context created with M = 1)");
    if (g.local_planes < 3) return FDTDX_OK;
    const uint64_t lo = 1, hi = g.local_planes - 1;
    fdtdx_comm *c = ctx->comm;
    if (!c) {
        if (ctx->cfg.nprocs != 1) return fail(FDTDX_ERR_INVALID, "no communicator: call fdtdx_comm_init or fdtdx_comm_init_local first");
        return fdtdx_yee3d_half(ctx, fields, which, coef, order, lo, hi, stream);
    }
    if (c->broken) return fail(FDTDX_ERR_HIP, "an earlier RCCL call of this communicator failed: destroy it (fdtdx_comm_destroy) and set up a new one");
    if (c->epoch != c->waited) return fail(FDTDX_ERR_INVALID, "the previous exchange has not been waited for (fdtdx_step_end)");
    const int s = which == 0 ? 1 : 0;                                   // the side the plane travels to: up after the H half, down after the E half
    const bool snd = s ? g.ghost_hi != 0 : g.ghost_lo != 0, rcv = s ? g.ghost_lo != 0 : g.ghost_hi != 0;
    const uint64_t zs = s ? g.local_planes - 2 : 1;                     // the owned plane next to that neighbour
    const uint64_t zr = s ? 0 : g.local_planes - 1;                     // the ghost plane the other neighbour fills
    float *cube[2];
    int rc;
    for (int k = 0; k < 2; k++)
        if ((rc = fdtdx_find_cube(ctx, fields[(which == 0 ? 3 : 0) + k], &cube[k]))) return rc;
    if (snd && (rc = fdtdx_yee3d_half(ctx, fields, which, coef, order, zs, zs + 1, stream))) return rc;
    USE_DEVICE(ctx);
    hipStream_t st = (hipStream_t)stream;
    HIP_TRY(hipEventRecord(c->ready, st));
    submit_now(st);
    c->marked = false;
    HIP_TRY(hipStreamWaitEvent(c->xs, c->ready, 0));
    const size_t cnt = (size_t)g.plane;
    if (c->mode == COMM_IPC) return fail(FDTDX_ERR_UNSUPPORTED, "the E/H half step moves single planes of two fields: RCCL or local transport only");
    if (c->mode == COMM_RCCL) {
        RcclApi *api;
        if ((rc = rccl(&api))) return rc;
        ncclResult_t first = api->GroupStart();  // as in fdtdx_comm_exchange_start: the group is closed whatever happens inside it
        if (first == ncclSuccess) {
            for (int k = 0; k < 2 && first == ncclSuccess; k++) {
                if (snd) first = api->Send(cube[k] + zs * g.plane, cnt, ncclFloat, c->peer[s], c->comm, c->xs);
                if (rcv && first == ncclSuccess) first = api->Recv(cube[k] + zr * g.plane, cnt, ncclFloat, c->peer[1 - s], c->comm, c->xs);
            }
            const ncclResult_t end = api->GroupEnd();
            if (first == ncclSuccess) first = end;
        }
        if (first != ncclSuccess) {
            c->broken = true;
            return fail(FDTDX_ERR_HIP, "RCCL send/recv group of the E/H half step: %s", api->GetErrorString(first));
        }
    } else if (snd) {
        fdtdx_ctx *o = c->nb[s];
        if (!o || !o->comm) return fail(FDTDX_ERR_INVALID, "the neighbouring slab is gone");
        if (o->g.plane != g.plane) return fail(FDTDX_ERR_INVALID, "the neighbouring slab has another geometry");
        // the neighbour's launches queued so far may still read the ghost plane the copies land in
        {  // always, the default stream included (see fdtdx_comm_exchange_start)
            HIP_TRY(hipSetDevice(o->device));
            HIP_TRY(hipEventRecord(o->comm->guard[1 - s], o->comm->last_stream));
            submit_now(o->comm->last_stream);
            HIP_TRY(hipSetDevice(ctx->device));
            HIP_TRY(hipStreamWaitEvent(c->xs, o->comm->guard[1 - s], 0));
        }
        const uint64_t zd = s ? 0 : o->g.local_planes - 1;              // its ghost plane on the side that faces this slab
        for (int k = 0; k < 2; k++) {
            float *dst;
            if ((rc = fdtdx_find_cube(o, fields[(which == 0 ? 3 : 0) + k], &dst))) return rc;
            if (o->device == ctx->device)
                HIP_TRY(hipMemcpyAsync(dst + zd * g.plane, cube[k] + zs * g.plane, cnt * sizeof(float), hipMemcpyDeviceToDevice, c->xs));
            else
                HIP_TRY(hipMemcpyPeerAsync(dst + zd * g.plane, o->device, cube[k] + zs * g.plane, ctx->device, cnt * sizeof(float), c->xs));
        }
    }
    HIP_TRY(hipEventRecord(c->done, c->xs));
    submit_now(c->xs);
    c->last_stream = st;
    c->epoch++;
    const uint64_t rlo = (snd && !s) ? lo + 1 : lo, rhi = (snd && s) ? hi - 1 : hi;
    return rhi > rlo ? fdtdx_yee3d_half(ctx, fields, which, coef, order, rlo, rhi, stream) : FDTDX_OK;
}

// The drivers' whole loop on n slabs (fdtd1.cu:68-80): n = 1 for a process that owns one slab and talks to the others
// through its RCCL communicator, n = all slabs for a process that owns them all.  Level t lives in b for even t and in a
// for odd t; fused launches where the ghost depth allows them; the last steps are single steps, so that a and b finish
// holding the two newest levels.
extern "C" int fdtdx_run(fdtdx_ctx *const *ctxs, int n, const char *a, const char *b, const fdtdx_program *prog, uint32_t steps,
                         void *const *streams)
{
    return fdtdx_run_work(ctxs, n, a, b, nullptr, prog, steps, streams);
}

// `work` (may be null): a third cube of every slab's pool for the two-step loops that hop through one (the radius-2 star:
// its fused kernel takes the even levels' shell from its input cube, so the loop runs b -> work -> b with `a` lending its
// shell to the odd levels, as fdtdx_sweep_work does on one rank).  Nothing is allocated here.
extern "C" int fdtdx_run_work(fdtdx_ctx *const *ctxs, int n, const char *a, const char *b, const char *work, const fdtdx_program *prog,
                              uint32_t steps, void *const *streams)
{
    if (!ctxs || n < 1 || !a || !b) return fail(FDTDX_ERR_INVALID, "null argument");
    uint64_t r;
    int rc = program_radius(prog, &r);
    if (rc) return rc;
    for (int i = 0; i < n; i++)
        if (!ctxs[i]) return fail(FDTDX_ERR_INVALID, "null context");
    auto stream_of = [&](int i) { return streams ? streams[i] : nullptr; };
    // how many steps a launch may fuse: the deepest kernel the library has for the program that the ghost depth supports
    // towards a neighbour (k steps of radius r reach k*r planes) and that can run between the two cubes alone
    int kt = 0;
    {
        int64_t want = ctxs[0]->opt_temporal;
        const uint64_t G = ctxs[0]->cfg.M;
        const bool multi = ctxs[0]->cfg.nprocs > 1;
        const bool r1 = (prog->kind == FDTDX_STAR && prog->radius == 1) || prog->kind == FDTDX_JACOBI7;
        if (steps >= 4 && want != 0) {
            if (r1 && want != 2 && (!multi || G >= 3 * r) && fdtdx_has_fused(ctxs[0], prog, 3, false))
                kt = 3;  // straight between the two cubes
            else if (want != 3 && (!multi || G >= 2 * r) && fdtdx_has_fused(ctxs[0], prog, 2, true))
                kt = 2;  // in place via the named shell sources (13-point)
            else if (want != 3 && work && (!multi || G >= 2 * r) && fdtdx_has_fused(ctxs[0], prog, 2, false))
                kt = -2;  // two steps per launch through the third cube (radius-2 star)
        }
    }
    if (kt == -2)
        for (int i = 0; i < n; i++) {
            float *pa, *pb, *pw;
            if ((rc = fdtdx_find_cube(ctxs[i], a, &pa)) || (rc = fdtdx_find_cube(ctxs[i], b, &pb)) || (rc = fdtdx_find_cube(ctxs[i], work, &pw))) return rc;
            if (pw == pa || pw == pb) return fail(FDTDX_ERR_INVALID, "the work cube must be a third cube");
        }
    auto each = [&](auto &&f) {
        for (int i = 0; i < n; i++)
            if (int e = f(i)) return e;
        return 0;
    };
    uint32_t done = 0;
    if (kt == 3) {
        for (uint32_t k = 0; k < (steps - 1) / 3; k++) {
            const char *out = (done % 2 == 0) ? a : b, *in = (done % 2 == 0) ? b : a;
            if ((rc = each([&](int i) { return fdtdx_hop_begin(ctxs[i], out, in, in, out, prog, 3, stream_of(i)); }))) return rc;
            if ((rc = each([&](int i) { return fdtdx_step_end(ctxs[i], stream_of(i)); }))) return rc;
            done += 3;
        }
    } else if (kt == 2) {
        for (uint32_t k = 0; k < 2 * ((steps - 1) / 4); k++) {
            const char *out = (k % 2 == 0) ? a : b, *in = (k % 2 == 0) ? b : a;
            if ((rc = each([&](int i) { return fdtdx_hop_begin(ctxs[i], out, in, b, a, prog, 2, stream_of(i)); }))) return rc;
            if ((rc = each([&](int i) { return fdtdx_step_end(ctxs[i], stream_of(i)); }))) return rc;
            done += 2;
        }
    } else if (kt == -2) {
        const uint32_t pairs = (steps % 2 == 0) ? steps / 2 - 1 : ((steps - 1) / 2) & ~1u;
        if (pairs > 0) {
            // work starts as a copy of b (ghost planes and shell included): it carries b's shell through the loop
            if ((rc = each([&](int i) {
                    float *pb, *pw;
                    int e;
                    if ((e = fdtdx_find_cube(ctxs[i], b, &pb)) || (e = fdtdx_find_cube(ctxs[i], work, &pw))) return e;
                    if (hipSetDevice(ctxs[i]->device) != hipSuccess) return fail(FDTDX_ERR_HIP, "hipSetDevice(%d) failed", ctxs[i]->device);
                    if (hipMemcpyAsync(pw, pb, ctxs[i]->g.cube_elems * sizeof(float), hipMemcpyDeviceToDevice, (hipStream_t)stream_of(i)) != hipSuccess)
                        return fail(FDTDX_ERR_HIP, "copy of cube '%s' into the work cube failed", b);
                    return 0;
                })))
                return rc;
            const char *cur = b, *oth = work;
            for (uint32_t k = 0; k < pairs; k++) {
                if ((rc = each([&](int i) { return fdtdx_hop_begin(ctxs[i], oth, cur, cur, a, prog, 2, stream_of(i)); }))) return rc;
                if ((rc = each([&](int i) { return fdtdx_step_end(ctxs[i], stream_of(i)); }))) return rc;
                std::swap(cur, oth);
            }
            done = 2 * pairs;
            if (cur != b) {  // even step count, odd number of pairs: the newest even level sits in the work cube; step from there
                if ((rc = each([&](int i) { return fdtdx_step_begin(ctxs[i], a, cur, prog, stream_of(i)); }))) return rc;
                if ((rc = each([&](int i) { return fdtdx_step_end(ctxs[i], stream_of(i)); }))) return rc;
                done++;
            }
        }
    }
    // single steps: where there are more than a few, the sweeps take the shell components of partly stored quads from strips
    // gathered once, here (the cubes' shells do not change while the loop runs)
    if (prog->kind != FDTDX_SPC13 && steps - done >= 8)
        for (int i = 0; i < n; i++) {
            float *pa, *pb;
            if (fdtdx_find_cube(ctxs[i], a, &pa) == FDTDX_OK && fdtdx_find_cube(ctxs[i], b, &pb) == FDTDX_OK)
                (void)fdtdx_strips_bind(ctxs[i], pa, pb, (int)r, stream_of(i));
        }
    rc = FDTDX_OK;
    for (uint32_t t = done; t < steps && !rc; t++) {
        const char *out = (t % 2 == 0) ? a : b, *in = (t % 2 == 0) ? b : a;
        rc = each([&](int i) { return fdtdx_step_begin(ctxs[i], out, in, prog, stream_of(i)); });
        if (!rc) rc = each([&](int i) { return fdtdx_step_end(ctxs[i], stream_of(i)); });
    }
    for (int i = 0; i < n; i++) fdtdx_strips_release(ctxs[i]);
    return rc;
}
