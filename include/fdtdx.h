/*
 * fdtdx.h -- C ABI of the MI355X (gfx950) stencil sweep engine.
 *
 * Drop-in boundary for the `Stencil<...>` host class of the reference (stencil.h:357-842) and
 * the per-step sweep it dispatches (stencil.h:684-768).  Device lambdas cannot cross a C ABI,
 * so the per-cell arithmetic is selected from the closed set of programs the reference ships
 * (fdtd1..8.cu, fdtdspc.cu, jacobi.cu) through `fdtdx_program`.
 *
 * Conventions
 *   - every function returns FDTDX_OK (0) or an FDTDX_ERR_* code and never aborts; the text of
 *     the last failure on the calling thread is available from fdtdx_last_error().  (The
 *     reference asserts/aborts instead: stencil.h:16-22; the C++ mirror in fdtdx/stencil.hpp
 *     restores that behaviour on top of these codes.)
 *   - all pointers are plain host or device pointers; `stream` is a hipStream_t passed as
 *     void* (NULL = the default stream).  Launches are asynchronous on that stream.
 *   - device cubes are laid out [local plane][y][PX] fp32, x unit-stride, PX = padded pitch.
 *   - one context = one z-slab = one GPU = one process (one MPI rank in the reference).
 */
#ifndef FDTDX_H
#define FDTDX_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct fdtdx_ctx fdtdx_ctx;

enum {
    FDTDX_OK = 0,
    FDTDX_ERR_INVALID = 1,     /* bad argument (reference: assert, stencil.h:387,403) */
    FDTDX_ERR_NOTFOUND = 2,    /* unknown buffer name (reference: assert, stencil.h:470,482,580) */
    FDTDX_ERR_EXISTS = 3,      /* duplicate buffer name (reference: assert, stencil.h:443,555) */
    FDTDX_ERR_HIP = 4,         /* HIP runtime failure (reference: CUDA_ASSERT, stencil.h:16-22) */
    FDTDX_ERR_UNSUPPORTED = 5, /* program/shape outside what the kernels cover */
    FDTDX_ERR_NOMEM = 6
};

/* Which per-cell program runs (the device lambdas of the shipped drivers). */
enum {
    FDTDX_STAR = 0,    /* fdtdN.cu:61-64 .. fdtd8.cu:82-93: radius-M star, coef[0..6] =
                          {centre, x+, x-, y+, y-, z+, z-} shared by every radius r */
    FDTDX_SPC13 = 1,   /* fdtdspc.cu:64-73: coef[0..12] in lambda order */
    FDTDX_JACOBI7 = 2  /* jacobi.cu:31-38: coef[0..6] = {cc, ce, cw, cs, cn, ct, cb},
                          evaluated ce*E+cw*W+cs*S+cn*N+ct*T+cb*B+cc*C */
};

/* Floating-point association of the sum. */
enum {
    FDTDX_ORDER_CHAIN = 0,  /* device order: flat left-to-right, every "+ c*v" one FMA
                               (nvcc -fmad=true, CMakeLists.txt:5) */
    FDTDX_ORDER_GOLDEN = 1  /* host golden order: parenthesised (+r,-r) pairs, separate IEEE
                               mul/add (fdtd1.cu:46-49, fdtdspc.cu:46-52, jacobi.cu:478-485);
                               bit-identical to the reference CPU loop */
};

typedef struct {
    uint32_t kind;   /* FDTDX_STAR | FDTDX_SPC13 | FDTDX_JACOBI7 */
    uint32_t radius; /* M for FDTDX_STAR (1..8); ignored (1) otherwise */
    uint32_t order;  /* FDTDX_ORDER_* */
    uint32_t reserved;
    float coef[16];
} fdtdx_program;

/* Replaces the template arguments + MPI rank/size of `Stencil` (stencil.h:357,385-413). */
typedef struct {
    uint64_t Z, Y, X;     /* global logical grid (ZSize, YSize, XSize) */
    uint32_t M;           /* ghost/shell width: the radius the cubes are sized for (1..8) */
    uint32_t rank;        /* this slab's index along z (MPI rank) */
    uint32_t nprocs;      /* number of z-slabs (MPI size) */
    int32_t device;       /* HIP device ordinal; -1 = keep the current device */
    uint32_t pitch_x;     /* row pitch in floats, multiple of 32; 0 = (X/32+1)*32 as the reference's
                             PADDINGR default for M<=2 (stencil.h:357) */
    uint32_t flags;       /* reserved, 0 */
} fdtdx_config;

/* z-slab decomposition and padded layout (stencil.h:398-412). */
typedef struct {
    uint64_t pitch_x;       /* PX: floats per row */
    uint64_t plane;         /* Y*PX: floats per plane */
    uint64_t owned_begin;   /* mpiOff[rank]: first global plane owned */
    uint64_t owned_planes;  /* mpiSize[rank] */
    uint64_t ghost_lo;      /* M if rank>0 else 0: ghost planes below the owned range */
    uint64_t ghost_hi;      /* M if rank<nprocs-1 else 0 */
    uint64_t local_planes;  /* ghost_lo + owned_planes + ghost_hi  (size2/plane) */
    uint64_t first_global;  /* global z of local plane 0          (off2/plane) */
    uint64_t update_lo;     /* first local plane a sweep writes: M */
    uint64_t update_hi;     /* one past the last local plane a sweep writes: local_planes-M */
    uint64_t cube_elems;    /* local_planes*plane */
} fdtdx_geometry;

/* Pure host function (no GPU): the decomposition for any rank. stencil.h:398-412. */
int fdtdx_decompose(const fdtdx_config *cfg, fdtdx_geometry *out);

/* Stencil::Stencil / ~Stencil (stencil.h:385-418), minus MPI_Init/Finalize which belong to the
 * launcher (torch.distributed / RCCL) in this design. */
int fdtdx_create(const fdtdx_config *cfg, fdtdx_ctx **out);
int fdtdx_destroy(fdtdx_ctx *ctx);
const char *fdtdx_last_error(void);
int fdtdx_get_geometry(const fdtdx_ctx *ctx, fdtdx_geometry *out);

/* Stencil::getRank / getProcs / sync (stencil.h:420-433). */
int fdtdx_rank(const fdtdx_ctx *ctx);
int fdtdx_nprocs(const fdtdx_ctx *ctx);
int fdtdx_sync(fdtdx_ctx *ctx);

/* Stencil::mallocCube / free / getCube (stencil.h:553-582, 468-477).  Cubes are zero-filled.
 * attach_cube registers caller-owned device memory (>= cube_elems floats) under a name; the
 * context never frees it. */
int fdtdx_malloc_cube(fdtdx_ctx *ctx, const char *name);
int fdtdx_attach_cube(fdtdx_ctx *ctx, const char *name, void *device_ptr, size_t bytes);
int fdtdx_free(fdtdx_ctx *ctx, const char *name);
int fdtdx_get_cube(fdtdx_ctx *ctx, const char *name, float **device_ptr);

/* Stencil::transferCubeToGPU (stencil.h:596-608): `host` is the whole unpadded [Z][Y][X] grid;
 * this slab's planes (ghosts included) are padded and copied to the device. */
int fdtdx_transfer_cube_to_gpu(fdtdx_ctx *ctx, const char *name, const float *host);
/* Stencil::transferCubeToCPU (stencil.h:610-622), local part: writes the planes this rank OWNS
 * into their place in the unpadded global grid `host`.  The cross-rank gather/broadcast of
 * stencil.h:623-636 is done by the caller's communicator (fdtd_amd.slab.gather_cube). */
int fdtdx_transfer_cube_to_cpu(fdtdx_ctx *ctx, float *host, const char *name);
/* Stencil::transferCubeToCPU whole (stencil.h:610-637, the Recv/Send/Bcast of :623-636 included): every rank ends with
 * the WHOLE unpadded [Z][Y][X] grid in `host`.  With an RCCL communicator made by fdtdx_comm_init (every rank must
 * call): each slab's owned planes are broadcast device to device and unpadded on arrival.  Slabs of one process
 * (fdtdx_comm_init_local): one call on any slab fills the whole grid.  One rank: same as fdtdx_transfer_cube_to_cpu. */
int fdtdx_gather_cube(fdtdx_ctx *ctx, float *host, const char *name);
/* Same transfers for a caller that holds only its own planes: `host_local` is
 * [local_planes][Y][X] (upload) / [owned_planes][Y][X] (download). */
int fdtdx_upload_local(fdtdx_ctx *ctx, const char *name, const float *host_local);
int fdtdx_download_owned(fdtdx_ctx *ctx, float *host_owned, const char *name);

/* Initial fields of the fdtdN/fdtdspc drivers generated in place (fdtd1.cu:21-31:
 * std::mt19937_64(19950918+z) per global plane, draws p0 then p1 per cell) and of jacobi.cu
 * (jacobi.cu:412-417: i - j + 1/(k+1)).  Host-generated, then uploaded. */
int fdtdx_init_mt(fdtdx_ctx *ctx, const char *name_p0, const char *name_p1);
int fdtdx_init_jacobi(fdtdx_ctx *ctx, const char *name);

/* Stencil::propagate (stencil.h:684-768): one sweep out = f(in) over every plane this rank
 * updates ([update_lo, update_hi)), interior [h, n-h) in y and x (h = program radius). */
int fdtdx_propagate(fdtdx_ctx *ctx, const char *out, const char *in, const fdtdx_program *prog,
                    void *stream);
/* The same sweep restricted to local planes [z_lo, z_hi): the boundary-planes-first / interior
 * split that replaces propagateHalo{Top,Buttom}Backup (stencil.h:770-808). */
int fdtdx_propagate_planes(fdtdx_ctx *ctx, const char *out, const char *in, const fdtdx_program *prog,
                           uint64_t z_lo, uint64_t z_hi, void *stream);
/* propagateHaloTopBackup and propagateHaloButtomBackup (stencil.h:770-808) as ONE launch: two plane ranges, the second
 * behind the first (either may be empty).  Two launches of a handful of planes in a row each leave most of the device
 * idle; fdtdx_step_begin sweeps the planes next to both neighbours this way. */
int fdtdx_propagate_two_ranges(fdtdx_ctx *ctx, const char *out, const char *in, const fdtdx_program *prog,
                               uint64_t z_lo, uint64_t z_hi, uint64_t z2_lo, uint64_t z2_hi, void *stream);
/* Raw-pointer form for callers that manage their own cubes (same layout/geometry as ctx). */
int fdtdx_propagate_ptr(fdtdx_ctx *ctx, float *out, const float *in, const fdtdx_program *prog,
                        uint64_t z_lo, uint64_t z_hi, void *stream);

/* Fused time steps restricted to local planes [z_lo, z_hi): out = f^nsteps(in), nsteps = 2 or 3,
 * the launch jacobi.cu:446-454 issues (temporal_blocking, jacobi.cu:115-400: three steps),
 * generalised to cubes whose shells differ.  The reference's two cubes keep different shells
 * (fdtd1.cu:28-29) and step t+1 reads the shell step t's output cube holds, so levels of equal
 * parity see equal shells: cells of the skipped odd levels that lie on the global shell are read
 * from `mid` (the cube a single step from `in` would have written), those of the skipped even
 * level from `in`.  `out` must already hold the shell of its level: that of `in` for nsteps = 2
 * (out != mid), that of `mid` for nsteps = 3 (out may be `mid` itself: the launch reads only its
 * shell cells and writes only interior cells).  Radius-1 programs (star, jacobi) with nsteps 2 or 3,
 * the radius-2 star and the 13-point program with nsteps 2 (FDTDX_ERR_UNSUPPORTED otherwise).  Towards a
 * neighbouring slab the range needs nsteps x radius ghost planes (create the context with M that large). */
int fdtdx_propagate_fused_planes(fdtdx_ctx *ctx, const char *out, const char *in, const char *mid,
                                 const fdtdx_program *prog, uint32_t nsteps, uint64_t z_lo, uint64_t z_hi,
                                 void *stream);
/* The same launch with both shell sources named: cells of the global shell are taken from `shell_even` for the even
 * levels of the launch (level 0, the input, included: `in` is read for its interior cells only where `shell_even` is
 * another cube) and from `shell_odd` for the odd ones; the shell of `out` is never written, whatever it holds.  This is
 * what lets a two-step loop run in place between the reference's two cubes (fdtd1.cu:68-80 keeps level t in one cube for
 * even t, in the other for odd t, shells included): out = a, in = b, then out = b, in = a, always with shell_even = b and
 * shell_odd = a; after an even number of launches b holds the newest even level inside its own shell.  No third cube.
 * FDTDX_ERR_UNSUPPORTED for the fused kernels that only take their even shell from `in` (radius-1 and radius-2 stars with
 * nsteps = 2 need out != shell_odd and shell_even == in). */
int fdtdx_propagate_fused_shells(fdtdx_ctx *ctx, const char *out, const char *in, const char *shell_even,
                                 const char *shell_odd, const fdtdx_program *prog, uint32_t nsteps, uint64_t z_lo,
                                 uint64_t z_hi, void *stream);
/* ... over two plane ranges in one launch (see fdtdx_propagate_two_ranges; fdtdx_hop_begin uses it). */
int fdtdx_propagate_fused_two_ranges(fdtdx_ctx *ctx, const char *out, const char *in, const char *shell_even,
                                     const char *shell_odd, const fdtdx_program *prog, uint32_t nsteps, uint64_t z_lo,
                                     uint64_t z_hi, uint64_t z2_lo, uint64_t z2_hi, void *stream);
/* nsteps = 2 */
int fdtdx_propagate2_planes(fdtdx_ctx *ctx, const char *out, const char *in, const char *mid,
                            const fdtdx_program *prog, uint64_t z_lo, uint64_t z_hi, void *stream);

/* The drivers' time loop on one rank (fdtd1.cu:68-80 with nprocs == 1): `steps` ping-pong
 * steps, step 0 writes a = f(b), then the roles swap.  On return cube `a`/`b` hold exactly what
 * the reference's two cubes hold after the same number of steps.
 * Memory: every context holds two strips of 8 floats per row of its local cube (0.4 % of a cube each, allocated by
 * fdtdx_create; the loops do without them if that allocation failed), into which the loops gather, once per call, the
 * shell cells of the quads that straddle the x shell, so that the sweeps need not re-read them from the output cube.
 * The fused loops of the radius-1 programs (three steps per launch) and of the 13-point program (two steps per
 * launch, in place) run between the two cubes alone.  The two-step loop of the radius-2 star (and of the radius-1
 * programs under option "temporal" = 2) hops between b and a THIRD cube of the same size that carries b's shell.
 * fdtdx_sweep allocates that cube on the first such call (hipMalloc, +50 % footprint, kept until fdtdx_destroy) and runs
 * single steps, with the same results, if the allocation fails.  fdtdx_sweep_work never allocates: the caller names the
 * third cube (`work`, any cube of the pool other than a and b; its contents are overwritten), or passes NULL, and the
 * programs that would need it then run single steps. */
int fdtdx_sweep(fdtdx_ctx *ctx, const char *a, const char *b, const fdtdx_program *prog, uint32_t steps,
                void *stream);
int fdtdx_sweep_work(fdtdx_ctx *ctx, const char *a, const char *b, const char *work, const fdtdx_program *prog,
                     uint32_t steps, void *stream);

/* Ghost-plane exchange regions (commCubeHalo, stencil.h:639-647): pointer/count of the planes to
 * SEND to (what = 0) or RECEIVE from (what = 1) the neighbour on `side` (0 = rank-1, 1 = rank+1).
 * count = M*plane floats, or 0 when there is no neighbour on that side.  Planes are contiguous,
 * so the region is handed to RCCL send/recv (or hipMemcpyPeer) without packing; the exchange
 * functions below do exactly that. */
int fdtdx_halo_region(fdtdx_ctx *ctx, const char *name, int side, int what, float **ptr, size_t *count);

/* ---- ghost-plane exchange with the z neighbours and the boundary-first time step ---------------------------------
 * commCubeHalo / commCubeHaloBackup with backup/restoreCubeHaloBackup (stencil.h:639-682) and the per-step sequence of
 * the drivers (fdtd1.cu:70-78).  The first and last M owned planes of a cube go to the neighbours' ghost planes of the
 * same cube (fdtdx_halo_region names them); no side buffers, no staging copies.  The exchange runs on a stream of its
 * own, ordered behind the boundary sweeps by an event, while the interior sweep runs on the caller's stream.
 *
 * Two transports.  RCCL: one process per slab (the reference's layout, one MPI rank per GPU); rank 0 makes an id with
 * fdtdx_comm_unique_id, the launcher hands its 128 bytes to every rank (MPI_Bcast where the reference calls MPI_Init,
 * stencil.h:389; torch.distributed or a file elsewhere), every rank calls fdtdx_comm_init.  Local: one process owns every
 * slab (fdtdx_comm_init_local on the contexts of ranks 0..n-1, each on its own device or several on one); copies go
 * through the copy engines.  A single host thread issues the slabs one after the other, so in local mode every exchange
 * (and every step below) is two phases: start / begin on EVERY slab, then wait / end on EVERY slab. */
typedef struct { char bytes[128]; } fdtdx_comm_id;
int fdtdx_comm_unique_id(fdtdx_comm_id *id);
int fdtdx_comm_init(fdtdx_ctx *ctx, const fdtdx_comm_id *id);
/* The same with the communicator laid out by the caller: this context is rank comm_rank of comm_size, its z neighbours
 * are ranks peer_lo / peer_hi (ignored where the slab has no neighbour). */
int fdtdx_comm_init_ranks(fdtdx_ctx *ctx, const fdtdx_comm_id *id, int comm_rank, int comm_size, int peer_lo, int peer_hi);
int fdtdx_comm_init_local(fdtdx_ctx *const *ctxs, int n);
int fdtdx_comm_destroy(fdtdx_ctx *ctx); /* fdtdx_destroy does it too */
/* A third transport for one process per GPU, with no kernel of the exchange on any CU: every rank maps its neighbours'
 * cubes into its own address space and pushes its boundary planes into their ghost planes with copies on its exchange
 * stream, beside its interior sweep; the hosts pace the exchange through a word of shared memory, as the reference's hosts
 * do with blocking MPI calls while their kernels run (fdtd1.cu:74-76; fdtdx_comm_exchange_wait / fdtdx_step_end block the
 * calling thread in this mode until this rank's copies and its neighbours' have landed).  Set-up: every rank calls
 * fdtdx_comm_ipc_export for the cubes it will exchange (all allocated already), the launcher hands each rank the
 * descriptors of its neighbours (plain bytes: a file, a pipe, torch.distributed), every rank calls
 * fdtdx_comm_ipc_connect (NULL where there is no neighbour).  Ranks must be processes of one machine.  The exchange relies
 * on the boundary-first step order of fdtdx_step_begin / fdtdx_hop_begin / fdtdx_run (a neighbour's ghost planes are
 * overwritten only after the neighbour has read them); fdtdx_gather_cube and the E/H half step are not available. */
#define FDTDX_IPC_MAX_CUBES 8
typedef struct fdtdx_ipc_desc {
    uint32_t rank, ncubes;
    int32_t pid, reserved;
    char shm_name[64];
    struct {
        char name[32];
        char handle[64];
        uint64_t offset, bytes;
    } cube[FDTDX_IPC_MAX_CUBES];
} fdtdx_ipc_desc;
int fdtdx_comm_ipc_export(fdtdx_ctx *ctx, const char *const *names, int n, fdtdx_ipc_desc *out);
int fdtdx_comm_ipc_connect(fdtdx_ctx *ctx, const fdtdx_ipc_desc *lo, const fdtdx_ipc_desc *hi);
/* Stencil::barrier (stencil.h:435-438): this slab's queued work has finished and every slab of the chain has arrived. */
int fdtdx_comm_barrier(fdtdx_ctx *ctx);
/* Diagnostic: one line of text saying which of this slab's exchange events have fired and which of its streams are idle
 * (queries only).  For finding the wait that never completes when a run with several streams stops making progress. */
int fdtdx_comm_state(fdtdx_ctx *ctx, char *buf, size_t len);
/* Start the exchange of cube `name` behind everything queued on `stream` so far (or behind the point last marked with
 * fdtdx_comm_boundary_done, for drivers that queue the interior sweep before they call the exchange, as fdtd1.cu:70-75
 * does) / make work queued on `stream` after the call see the received ghost planes.  cube_halo = start + wait. */
int fdtdx_comm_boundary_done(fdtdx_ctx *ctx, void *stream);
int fdtdx_comm_exchange_start(fdtdx_ctx *ctx, const char *name, void *stream);
int fdtdx_comm_exchange_wait(fdtdx_ctx *ctx, void *stream);
int fdtdx_comm_cube_halo(fdtdx_ctx *ctx, const char *name, void *stream);
/* One time step of a multi-rank driver, fdtd1.cu:70-78: sweeps of the M planes next to each neighbour
 * (propagateHalo{Top,Buttom}Backup), the exchange of `out` started behind them, the interior sweep beside it (begin);
 * the caller's stream waits for the exchange (end).  hop: the same around one fused launch of nsteps steps
 * (fdtdx_propagate_fused_shells; the context's ghost depth M must be nsteps x radius). */
int fdtdx_step_begin(fdtdx_ctx *ctx, const char *out, const char *in, const fdtdx_program *prog, void *stream);
int fdtdx_hop_begin(fdtdx_ctx *ctx, const char *out, const char *in, const char *shell_even, const char *shell_odd,
                    const fdtdx_program *prog, uint32_t nsteps, void *stream);
int fdtdx_step_end(fdtdx_ctx *ctx, void *stream);
/* The drivers' whole loop (fdtd1.cu:68-80) on n slabs: n = 1 for a process that owns one slab and reaches the others
 * through RCCL, n = all slabs of a process that owns them all (streams[i]: the stream of slab i, NULL = default streams).
 * Fuses three steps per launch for radius-1 programs and two for the 13-point program where the ghost depth allows
 * (M >= 3, M >= 2); on return a / b hold what the reference's cubes hold after `steps` steps. */
int fdtdx_run(fdtdx_ctx *const *ctxs, int n, const char *a, const char *b, const fdtdx_program *prog, uint32_t steps,
              void *const *streams);
/* The same loop with a third cube named (`work`, in every slab's pool; NULL = fdtdx_run): the radius-2 star then fuses two
 * steps per launch as well (ghost depth M >= 4), hopping b -> work -> b as fdtdx_sweep_work does on one rank.  Nothing is
 * allocated inside; the contents of `work` are overwritten. */
int fdtdx_run_work(fdtdx_ctx *const *ctxs, int n, const char *a, const char *b, const char *work, const fdtdx_program *prog,
                   uint32_t steps, void *const *streams);

/* ---- the composite pipeline of main.cu (the only caller of inject / filt / address lists) ------------ */

/* The update of main.cu:102-119 (device lambda) / :55-72 (host loop), radius 4 there:
 *   out = cz*z + cy*y + cx*x + vel*(scal_num/scal_den)*(C + wx*SUM_r(X+r + X-r) + wy*SUM_r(Y+r + Y-r)
 *                                                         + wz*SUM_r(Z+r + Z-r)) - out
 * with z, y, x the GLOBAL cell coordinates and C, X.., Y.., Z.. cells of `in`.  FDTDX_ORDER_GOLDEN
 * evaluates it as the host loop does (double constants, double arithmetic on float pair sums, one
 * rounding); FDTDX_ORDER_CHAIN as the device lambda (float constants, scal = (float)(num/den), FMAs).
 * Kernels are compiled for radius 4, the only radius main.cu uses; any other radius returns FDTDX_ERR_UNSUPPORTED. */
typedef struct fdtdx_composite {
    uint32_t radius, order; /* radius: 4 */
    double cz, cy, cx, scal_num, scal_den, wx, wy, wz;
} fdtdx_composite;
/* propagate(out, in, spill, prop_kernel, getCube("vel"), scal) of main.cu:160 on local planes [z_lo, z_hi)
 * (main.cu:153-154 are the same launch on boundary planes) / on all planes this rank updates. */
int fdtdx_propagate_composite_planes(fdtdx_ctx *ctx, const char *out, const char *in, const char *vel,
                                     const fdtdx_composite *comp, uint64_t z_lo, uint64_t z_hi, void *stream);
int fdtdx_propagate_composite(fdtdx_ctx *ctx, const char *out, const char *in, const char *vel,
                              const fdtdx_composite *comp, void *stream);

/* Address lists: malloc<size_t>(name, n, true) (stencil.h:446-466) and transferToGPU(name, addr, n)
 * (stencil.h:518-544): `global_addr` are linear indices z*Y*X + y*X + x into the undecomposed,
 * unpadded grid; the addresses whose plane this rank owns are kept and re-based to the local padded
 * cube.  fdtdx_addr_size = getAddrSize (stencil.h:500-504). */
int fdtdx_malloc_addr(fdtdx_ctx *ctx, const char *name, uint64_t capacity);
int fdtdx_transfer_addr_to_gpu(fdtdx_ctx *ctx, const char *name, const uint64_t *global_addr, uint64_t n);
int fdtdx_addr_size(fdtdx_ctx *ctx, const char *name, uint64_t *count);
/* get<size_t>(name) (stencil.h:479-484) for an address list: the device array of re-based local addresses (borrowed). */
int fdtdx_get_addr(fdtdx_ctx *ctx, const char *name, uint64_t **device_ptr);

/* inject (stencil.h:810-815) with the functor of main.cu:120-123: cube[addr[i]] += add (atomic) for the
 * listed addresses that lie in local planes [z_lo, z_hi) (the reference keeps separate lists for its
 * boundary side buffers, main.cu:155-156; here the range selects them). */
int fdtdx_inject(fdtdx_ctx *ctx, const char *cube, const char *addr_name, float add, uint64_t z_lo, uint64_t z_hi,
                 void *stream);
/* filt (stencil.h:817-823, boundary forms :825-841) with the functor of main.cu:124-128: every cell of
 * local planes [z_lo, z_hi), shell included, whose global z == zsel or y == ysel or x == xsel is
 * multiplied by `scale` (host order: in double, rounded once; device order: by (float)scale).
 * A negative selector selects nothing. */
int fdtdx_filt_planes(fdtdx_ctx *ctx, const char *cube, int64_t zsel, int64_t ysel, int64_t xsel, double scale,
                      uint32_t order, uint64_t z_lo, uint64_t z_hi, void *stream);
/* filt with the functor of main.cu:129-133: sum += a*b on the cells at least `radius` away from the
 * grid faces in y and x, in local planes [z_lo, z_hi) (the caller passes the interior planes). */
int fdtdx_mul_acc(fdtdx_ctx *ctx, const char *sum, const char *a, const char *b, uint32_t radius, uint32_t order,
                  uint64_t z_lo, uint64_t z_hi, void *stream);

/* One whole step of main.cu:160-163 on the planes this rank owns: out = composite(in, vel, out); inject(out);
 * filt(out); sum += out*in -- with the product-accumulate folded into the update kernel, so that a step moves
 * 24 instead of 32 bytes per cell.  The results are exactly those of the four calls above (cells that inject or
 * filt change after the update get their product from the final value).  Allocates a buffer of one float per
 * listed address inside the context on first use. */
int fdtdx_composite_step(fdtdx_ctx *ctx, const char *out, const char *in, const char *vel, const char *sum,
                         const char *addr_name, const fdtdx_composite *comp, float add, int64_t zsel, int64_t ysel,
                         int64_t xsel, double scale, void *stream);

/* fdtdx_composite_step for callers that cannot allocate device memory: code under HIP graph capture, tight serving
 * loops, harnesses that forbid hipMalloc.  No counterpart in the reference.  Same arguments, same results bit for bit;
 * the one buffer the step needs beyond its cubes (one float per listed address) is a file-scope device array of
 * FDTDX_STATIC_ADDR_MAX floats (1 MiB, held on every GPU the library runs on until the process ends) instead of
 * memory allocated inside the context, and the call allocates nothing.  A list with more addresses on this rank is
 * rejected with FDTDX_ERR_UNSUPPORTED.  All calls on one GPU share that GPU's array: calls that overlap in time on
 * one GPU (different streams or contexts) race; calls ordered on one stream do not. */
#define FDTDX_STATIC_ADDR_MAX 262144u
int fdtdx_composite_step_static(fdtdx_ctx *ctx, const char *out, const char *in, const char *vel, const char *sum,
                                const char *addr_name, const fdtdx_composite *comp, float add, int64_t zsel, int64_t ysel,
                                int64_t xsel, double scale, void *stream);

/* ---- extension: staggered-grid E/H updates -----------------------------------------------------------
 * The reference has no E/H kernels (its drivers are scalar ping-pong stencils); these entry points have no
 * counterpart in it and are defined by this library alone (CPU restatement: oracle/fdtd_oracle.c).
 *
 * 3-D Yee, six cubes fields[] = {Ex, Ey, Ez, Hx, Hy, Hz} of the context's geometry, outer shell untouched:
 *   which = 0:  Hx -= c*((Ez[y+1]-Ez) - (Ey[z+1]-Ey)), Hy -= c*((Ex[z+1]-Ex) - (Ez[x+1]-Ez)), Hz -= c*((Ey[x+1]-Ey) - (Ex[y+1]-Ex))
 *   which = 1:  Ex += c*((Hz-Hz[y-1]) - (Hy-Hy[z-1])), Ey += c*((Hx-Hx[z-1]) - (Hz-Hz[x-1])), Ez += c*((Hy-Hy[x-1]) - (Hx-Hx[y-1]))
 * on local planes [z_lo, z_hi) (1 <= z_lo, z_hi <= local_planes-1).  fdtdx_yee3d_step = H half step with ch, then
 * E half step with ce, on every interior plane (one rank). */
int fdtdx_yee3d_half(fdtdx_ctx *ctx, const char *const fields[6], int which, float coef, uint32_t order, uint64_t z_lo,
                     uint64_t z_hi, void *stream);
int fdtdx_yee3d_step(fdtdx_ctx *ctx, const char *const fields[6], float ce, float ch, uint32_t order, void *stream);
/* The same time step in ONE launch, out of place: reads the six cubes `before` (Ex, Ey, Ez, Hx, Hy, Hz), writes the
 * six cubes `after`; every field crosses HBM once in each direction (48 instead of 72 bytes per cell).  The step never
 * changes the shell, so the `after` cubes must already hold it (e.g. start them as copies of `before`); a time loop
 * ping-pongs between the two sets.  Results equal fdtdx_yee3d_step bit for bit.  Extension like the half steps. */
int fdtdx_yee3d_step_fused(fdtdx_ctx *ctx, const char *const before[6], const char *const after[6], float ce, float ch,
                           uint32_t order, void *stream);
/* One half step on a slab with neighbours (contexts created with M = 1 and a communicator): the one owned plane a
 * neighbour needs is updated first (which = 0: Hx, Hy of the last owned plane, for the slab above; which = 1: Ex, Ey of
 * the first owned plane, for the slab below), its two cubes' planes start towards that neighbour's ghost plane on the
 * exchange stream, the remaining planes are updated beside that.  fdtdx_step_end makes `stream` wait for the planes this
 * slab receives.  Slabs of one process: begin on every slab, then end on every slab.  Ez / Hz ghost planes are not kept
 * current (nothing differences them along z).  No counterpart in the reference. */
int fdtdx_yee3d_slab_half_begin(fdtdx_ctx *ctx, const char *const fields[6], int which, float coef, uint32_t order,
                                void *stream);
/* 2-D ex/ey/hz on dense [nx][ny] device arrays (ny fastest), `steps` steps numbered t0, t0+1, ...:
 *   ey[0][j] = t;  ey[i][j] -= ce*(hz[i][j]-hz[i-1][j]) (i>=1);  ex[i][j] -= ce*(hz[i][j]-hz[i][j-1]) (j>=1);
 *   hz[i][j] -= ch*(((ex[i][j+1]-ex[i][j]) + ey[i+1][j]) - ey[i][j])  (i<nx-1, j<ny-1). */
int fdtdx_yee2d_steps(int device, float *ex, float *ey, float *hz, uint64_t nx, uint64_t ny, int t0, int steps, float ce,
                      float ch, uint32_t order, void *stream);

/* Kernel-selection knobs, for tuning and tests: "zchunks" (z-chunks per sweep launch, 0 = auto),
 * "variant" (tile-shape index, -1 = auto), "temporal" (time steps fdtdx_sweep fuses per launch where
 * a fused kernel exists: 1 = as many as there is a kernel for, 2, 3, or 0 = one launch per step),
 * "exchange_priority" (1: the exchange stream of a communicator set up afterwards is created at the device's highest stream
 * priority; 0, default: an ordinary stream), "tb_variant" / "tb_zchunks" (shape and z-chunks of the fused kernel), "fuse_wide" (1: time loops and
 * fdtdx_propagate_fused_* may use the two-step kernels of star radius 3 and 4, which exist for measurement and run slower than
 * single steps; 0, default: those radii run one step per launch), "no_strip" (1: the few quads rows wider than one x tile have left
 * after their last full tile go to the half-width tile instead of the strip shape; for measurement), "reserve_cus" (CUs a sweep launch leaves
 * without a workgroup when it sizes its grid: room for the exchange's kernels beside an interior sweep),
 * "boundary_stream" (0, default: fdtdx_step_begin / fdtdx_hop_begin queue the boundary-plane sweep in the caller's
 * stream, ahead of the interior sweep; 1: on a stream of the library's own, beside it -- measured slower where the
 * interior sweep fills every CU, because the boundary sweep and the exchange behind it then wait for CUs). */
int fdtdx_set_option(fdtdx_ctx *ctx, const char *key, int64_t value);
/* The deepest fused launch the library has for `prog` on this context's grid: 3, 2, or 0 (none; e.g. radius >= 3, or the
 * 13-point program on rows of exactly 65 quads).  named_shells != 0: only kernels that take both shell sources by name
 * (fdtdx_propagate_fused_shells with shell_even != in).  A query, never an error for a valid program; negative = error
 * code negated.  Time loops written outside the library (the reference's drivers loop step by step, fdtd1.cu:68-80) ask
 * this before they plan fused hops, so that they fall back to single steps exactly where fdtdx_sweep / fdtdx_run do. */
int fdtdx_fused_depth(const fdtdx_ctx *ctx, const fdtdx_program *prog, int named_shells);
/* Describes the kernels the last sweeps launched (tile shape, grid, LDS; the last fused launch if
 * any; launches issued so far), for logs and bench. */
int fdtdx_last_launch_info(const fdtdx_ctx *ctx, char *buf, size_t len);

#ifdef __cplusplus
}
#endif
#endif /* FDTDX_H */
