// fdtdx_abi.hip -- host side of the C ABI declared in include/fdtdx.h.
//
// Mirrors the host runtime of the reference's Stencil<> class (stencil.h:357-842): named device
// buffer pool, z-slab decomposition, padded layout, padded transfers, sweep dispatch.  Process
// to process communication is NOT in here (the reference calls MPI at stencil.h:639-656): this
// library exposes the contiguous ghost/boundary plane regions (fdtdx_halo_region) and the
// caller moves them with RCCL send/recv (fdtd_amd/slab.py) while the interior sweep runs.
#include "fdtdx_ctx.h"
#include "sweep_common.h"
#include "tb_kernel.h"
#include "yee_kernel.h"

#include <algorithm>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <map>
#include <mutex>
#include <random>
#include <string>
#include <thread>
#include <vector>

namespace fdtdx {

// ---- variant registries ----------------------------------------------------------------------
static std::vector<VariantInfo> &vreg()
{
    static std::vector<VariantInfo> v;
    return v;
}
static std::vector<SpcVariantInfo> &sreg()
{
    static std::vector<SpcVariantInfo> v;
    return v;
}
int num_variants() { return (int)vreg().size(); }
const VariantInfo *variant(int i) { return &vreg()[i]; }
void register_variant(const VariantInfo &v) { vreg().push_back(v); }
static std::vector<TbVariantInfo> &treg()
{
    static std::vector<TbVariantInfo> v;
    return v;
}
int num_tb_variants() { return (int)treg().size(); }
const TbVariantInfo *tb_variant(int i) { return &treg()[i]; }
void register_tb_variant(const TbVariantInfo &v) { treg().push_back(v); }
int num_spc_variants() { return (int)sreg().size(); }
const SpcVariantInfo *spc_variant(int i) { return &sreg()[i]; }
void register_spc_variant(const SpcVariantInfo &v) { sreg().push_back(v); }

}  // namespace fdtdx

using namespace fdtdx;

// ---- error reporting -----------------------------------------------------------------------
static thread_local char g_err[512] = "";

int fdtdx_fail(int code, const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

extern "C" const char *fdtdx_last_error(void) { return g_err; }

// ---- decomposition (stencil.h:398-412) ---------------------------------------------------------
extern "C" int fdtdx_decompose(const fdtdx_config *cfg, fdtdx_geometry *out)
{
    if (!cfg || !out) return fail(FDTDX_ERR_INVALID, "null argument");
    const uint64_t M = cfg->M, np = cfg->nprocs, r = cfg->rank;
    if (M < 1 || M > 8) return fail(FDTDX_ERR_INVALID, "M=%u outside 1..8 (stencil.h:387)", cfg->M);
    if (np < 1 || r >= np) return fail(FDTDX_ERR_INVALID, "rank %u / nprocs %u", cfg->rank, cfg->nprocs);
    if (cfg->Z < 2 * M + 1 || cfg->Y < 2 * M + 1 || cfg->X < 2 * M + 1)
        return fail(FDTDX_ERR_INVALID, "grid %llux%llux%llu smaller than 2M+1 (stencil.h:387)",
                    (unsigned long long)cfg->Z, (unsigned long long)cfg->Y, (unsigned long long)cfg->X);
    uint64_t px = cfg->pitch_x ? cfg->pitch_x : (cfg->X / 32 + 1) * 32;
    if (px % 32 != 0 || px < cfg->X) return fail(FDTDX_ERR_INVALID, "pitch_x %llu must be a multiple of 32 >= X", (unsigned long long)px);
    if (cfg->Y * px * 4 > 0x7fffffffull) return fail(FDTDX_ERR_UNSUPPORTED, "plane of %llu floats exceeds 2^31 bytes", (unsigned long long)(cfg->Y * px));
    uint64_t off = 0, size = 0;
    for (uint64_t i = 0; i < np; i++) {
        uint64_t s = cfg->Z / np + ((cfg->Z % np) > i ? 1 : 0);  // mpiSize[i]
        if (s <= 2 * M) return fail(FDTDX_ERR_INVALID, "slab %llu has %llu planes, needs > 2M (stencil.h:403)", (unsigned long long)i, (unsigned long long)s);
        if (i == r) {
            size = s;
            break;
        }
        off += s;
    }
    fdtdx_geometry g;
    g.pitch_x = px;
    g.plane = cfg->Y * px;
    g.owned_begin = off;
    g.owned_planes = size;
    g.ghost_lo = (r == 0) ? 0 : M;
    g.ghost_hi = (r == np - 1) ? 0 : M;
    g.local_planes = g.ghost_lo + size + g.ghost_hi;
    g.first_global = off - g.ghost_lo;
    g.update_lo = M;
    g.update_hi = g.local_planes - M;
    g.cube_elems = g.local_planes * g.plane;
    *out = g;
    return FDTDX_OK;
}

// ---- scratch of the entry point that allocates nothing (fdtdx_composite_step_static) ------------------
// One float per listed injection address.  The array is part of the code object, so every GPU the library
// runs on holds its own copy, at its own address, until the process ends.  In host code the array's NAME is
// only a handle for hipGetSymbolAddress: the device address is asked for once per device (below, from
// fdtdx_create with that device current) and only that address is ever handed to kernels.
// (No `used` attribute on it: that makes the symbol local to the code object, and the runtime's lookup
// by name then aborts the process instead of returning an error.)
static __device__ __align__(256) float scratch[FDTDX_STATIC_ADDR_MAX];
constexpr int kMaxDevices = 64;
static float *g_scratch_addr[kMaxDevices];  // per device: where `scratch` lives there (nullptr: not asked yet)
static std::mutex g_scratch_mutex;

// the calling thread's current device must be `dev`
static int scratch_address(int dev, float **out)
{
    if (dev < 0 || dev >= kMaxDevices) return fail(FDTDX_ERR_UNSUPPORTED, "device ordinal %d outside 0..%d", dev, kMaxDevices - 1);
    std::lock_guard<std::mutex> lock(g_scratch_mutex);
    if (!g_scratch_addr[dev]) {
        float *ptr = nullptr;
        const hipError_t e = hipGetSymbolAddress((void **)&ptr, HIP_SYMBOL(scratch));
        if (e != hipSuccess || !ptr) return fail(FDTDX_ERR_HIP, "hipGetSymbolAddress(scratch) on device %d: %s", dev, hipGetErrorString(e));
        g_scratch_addr[dev] = ptr;
    }
    *out = g_scratch_addr[dev];
    return FDTDX_OK;
}

extern "C" int fdtdx_create(const fdtdx_config *cfg, fdtdx_ctx **out)
{
    if (!cfg || !out) return fail(FDTDX_ERR_INVALID, "null argument");
    fdtdx_geometry g;
    int rc = fdtdx_decompose(cfg, &g);
    if (rc) return rc;
    if (g.local_planes > 0x7fffffffull) return fail(FDTDX_ERR_UNSUPPORTED, "too many planes");
    int dev = cfg->device;
    if (dev >= 0)
        HIP_TRY(hipSetDevice(dev));
    else
        HIP_TRY(hipGetDevice(&dev));
    fdtdx_ctx *c = new (std::nothrow) fdtdx_ctx;
    if (!c) return fail(FDTDX_ERR_NOMEM, "out of host memory");
    c->cfg = *cfg;
    c->g = g;
    c->device = dev;
    {
        int cus = 0;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && cus > 0) c->num_cus = cus;
        else (void)hipGetLastError();
    }
    // set-up of the static scratch for this device (device `dev` is current here); a failure is kept for
    // fdtdx_composite_step_static to report, every other entry point works without the array
    if (scratch_address(dev, &c->static_scratch) != FDTDX_OK) c->static_scratch = nullptr;
    // two strips of two quads per row for the time loops (0.4 % of a cube each; the loops do without them if there is no room)
    for (int i = 0; i < 2; i++)
        if (hipMalloc((void **)&c->shell_strip[i], c->g.local_planes * c->cfg.Y * 8 * sizeof(float)) != hipSuccess) {
            (void)hipGetLastError();
            c->shell_strip[i] = nullptr;
        }
    *out = c;
    return FDTDX_OK;
}

extern "C" int fdtdx_destroy(fdtdx_ctx *ctx)
{
    if (!ctx) return FDTDX_OK;
    int rc = FDTDX_OK;
    if (hipSetDevice(ctx->device) != hipSuccess) rc = fail(FDTDX_ERR_HIP, "hipSetDevice(%d) failed", ctx->device);
    for (auto &kv : ctx->pool)
        if (kv.second.owned && kv.second.ptr && hipFree(kv.second.ptr) != hipSuccess) rc = fail(FDTDX_ERR_HIP, "hipFree(%s) failed", kv.first.c_str());
    if (ctx->tb_scratch && hipFree(ctx->tb_scratch) != hipSuccess) rc = fail(FDTDX_ERR_HIP, "hipFree(scratch) failed");
    if (ctx->stage && hipFree(ctx->stage) != hipSuccess) rc = fail(FDTDX_ERR_HIP, "hipFree(staging) failed");
    if (ctx->saved_sum && hipFree(ctx->saved_sum) != hipSuccess) rc = fail(FDTDX_ERR_HIP, "hipFree(saved) failed");
    for (int i = 0; i < 2; i++)
        if (ctx->shell_strip[i] && hipFree(ctx->shell_strip[i]) != hipSuccess) rc = fail(FDTDX_ERR_HIP, "hipFree(shell strip) failed");
    fdtdx_comm_release(ctx);
    for (auto &kv : ctx->addr)
        if (kv.second.ptr && hipFree(kv.second.ptr) != hipSuccess) rc = fail(FDTDX_ERR_HIP, "hipFree(%s) failed", kv.first.c_str());
    delete ctx;
    return rc;
}

extern "C" int fdtdx_get_geometry(const fdtdx_ctx *ctx, fdtdx_geometry *out)
{
    if (!ctx || !out) return fail(FDTDX_ERR_INVALID, "null argument");
    *out = ctx->g;
    return FDTDX_OK;
}

extern "C" int fdtdx_rank(const fdtdx_ctx *ctx) { return ctx ? (int)ctx->cfg.rank : -1; }
extern "C" int fdtdx_nprocs(const fdtdx_ctx *ctx) { return ctx ? (int)ctx->cfg.nprocs : -1; }

extern "C" int fdtdx_sync(fdtdx_ctx *ctx)
{
    USE_DEVICE(ctx);
    HIP_TRY(hipDeviceSynchronize());
    return FDTDX_OK;
}

// ---- pool ------------------------------------------------------------------------------------
extern "C" int fdtdx_malloc_cube(fdtdx_ctx *ctx, const char *name)
{
    if (!ctx || !name) return fail(FDTDX_ERR_INVALID, "null argument");
    if (ctx->pool.count(name)) return fail(FDTDX_ERR_EXISTS, "cube '%s' already exists (stencil.h:555)", name);
    USE_DEVICE(ctx);
    float *p = nullptr;
    size_t bytes = ctx->g.cube_elems * sizeof(float);
    HIP_TRY(hipMalloc((void **)&p, bytes));
    hipError_t e = hipMemset(p, 0, bytes);
    if (e != hipSuccess) {
        (void)hipFree(p);
        return fail(FDTDX_ERR_HIP, "hipMemset: %s", hipGetErrorString(e));
    }
    Cube c;
    c.ptr = p;
    c.owned = true;
    ctx->pool[name] = c;
    return FDTDX_OK;
}

extern "C" int fdtdx_attach_cube(fdtdx_ctx *ctx, const char *name, void *device_ptr, size_t bytes)
{
    if (!ctx || !name || !device_ptr) return fail(FDTDX_ERR_INVALID, "null argument");
    if (ctx->pool.count(name)) return fail(FDTDX_ERR_EXISTS, "cube '%s' already exists", name);
    if (bytes < ctx->g.cube_elems * sizeof(float))
        return fail(FDTDX_ERR_INVALID, "attach '%s': %zu bytes < cube size %zu", name, bytes, (size_t)(ctx->g.cube_elems * sizeof(float)));
    if (((uintptr_t)device_ptr & 15u) != 0) return fail(FDTDX_ERR_INVALID, "attach '%s': pointer not 16-byte aligned", name);
    Cube c;
    c.ptr = (float *)device_ptr;
    c.owned = false;
    ctx->pool[name] = c;
    return FDTDX_OK;
}

extern "C" int fdtdx_free(fdtdx_ctx *ctx, const char *name)
{
    if (!ctx || !name) return fail(FDTDX_ERR_INVALID, "null argument");
    auto it = ctx->pool.find(name);
    if (it == ctx->pool.end()) return fail(FDTDX_ERR_NOTFOUND, "no cube '%s' (stencil.h:470)", name);
    USE_DEVICE(ctx);
    if (it->second.owned) HIP_TRY(hipFree(it->second.ptr));
    ctx->pool.erase(it);
    return FDTDX_OK;
}

int fdtdx_find_cube(fdtdx_ctx *ctx, const char *name, float **p)
{
    if (!ctx || !name) return fail(FDTDX_ERR_INVALID, "null argument");
    auto it = ctx->pool.find(name);
    if (it == ctx->pool.end()) return fail(FDTDX_ERR_NOTFOUND, "no cube '%s' (stencil.h:580)", name);
    *p = it->second.ptr;
    return FDTDX_OK;
}

extern "C" int fdtdx_get_cube(fdtdx_ctx *ctx, const char *name, float **device_ptr)
{
    if (!device_ptr) return fail(FDTDX_ERR_INVALID, "null argument");
    return fdtdx_find_cube(ctx, name, device_ptr);
}

// ---- transfers (stencil.h:596-637): pitch conversion on the device ----------------------------------
// The unpadded planes cross PCIe as flat copies into (out of) a staging buffer on the device, and a device-to-device 2-D copy
// pads (unpads) them there.  (A 2-D copy straight between pageable host memory and the padded cube moves row by row: 0.2-0.3
// GB/s measured for the 2 KB rows of a 512^3 cube, 2.4 s per cube.)  The staging buffer belongs to the context (at most 256 MB,
// allocated on first use); without it the direct 2-D copy is used.  All copies go through the null stream, in order.
static uint64_t stage_planes(fdtdx_ctx *ctx, uint64_t nplanes)
{
    const uint64_t per_plane = ctx->cfg.X * ctx->cfg.Y;
    const uint64_t want = std::max<uint64_t>(1, std::min<uint64_t>(nplanes, (256ull << 20) / (per_plane * sizeof(float))));
    if (ctx->stage_floats < want * per_plane) {
        if (ctx->stage) (void)hipFree(ctx->stage);
        ctx->stage = nullptr, ctx->stage_floats = 0;
        if (hipMalloc((void **)&ctx->stage, want * per_plane * sizeof(float)) != hipSuccess) {
            (void)hipGetLastError();
            ctx->stage = nullptr;
            return 0;
        }
        ctx->stage_floats = want * per_plane;
    }
    return ctx->stage_floats / per_plane;
}

static int copy_planes_h2d(fdtdx_ctx *ctx, float *dev, const float *host, uint64_t nplanes)
{
    USE_DEVICE(ctx);
    const fdtdx_geometry &g = ctx->g;
    const uint64_t X = ctx->cfg.X, Y = ctx->cfg.Y;
    const uint64_t batch = stage_planes(ctx, nplanes);
    if (!batch) {
        HIP_TRY(hipMemcpy2D(dev, g.pitch_x * sizeof(float), host, X * sizeof(float), X * sizeof(float), nplanes * Y, hipMemcpyHostToDevice));
        return FDTDX_OK;
    }
    for (uint64_t z = 0; z < nplanes; z += batch) {
        const uint64_t n = std::min(batch, nplanes - z);
        HIP_TRY(hipMemcpy(ctx->stage, host + z * Y * X, n * Y * X * sizeof(float), hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy2D(dev + z * g.plane, g.pitch_x * sizeof(float), ctx->stage, X * sizeof(float), X * sizeof(float), n * Y, hipMemcpyDeviceToDevice));
    }
    return FDTDX_OK;
}
int fdtdx_copy_planes_d2h(fdtdx_ctx *ctx, float *host, const float *dev, uint64_t nplanes);
static int copy_planes_d2h(fdtdx_ctx *ctx, float *host, const float *dev, uint64_t nplanes) { return fdtdx_copy_planes_d2h(ctx, host, dev, nplanes); }
int fdtdx_copy_planes_d2h(fdtdx_ctx *ctx, float *host, const float *dev, uint64_t nplanes)
{
    USE_DEVICE(ctx);
    // stencil.h:615: the reference synchronises the device before it copies a cube back; sweeps queued on
    // any stream (non-blocking ones included) have finished when the copy reads the cube
    HIP_TRY(hipDeviceSynchronize());
    const fdtdx_geometry &g = ctx->g;
    const uint64_t X = ctx->cfg.X, Y = ctx->cfg.Y;
    const uint64_t batch = stage_planes(ctx, nplanes);
    if (!batch) {
        HIP_TRY(hipMemcpy2D(host, X * sizeof(float), dev, g.pitch_x * sizeof(float), X * sizeof(float), nplanes * Y, hipMemcpyDeviceToHost));
        return FDTDX_OK;
    }
    for (uint64_t z = 0; z < nplanes; z += batch) {
        const uint64_t n = std::min(batch, nplanes - z);
        HIP_TRY(hipMemcpy2D(ctx->stage, X * sizeof(float), dev + z * g.plane, g.pitch_x * sizeof(float), X * sizeof(float), n * Y, hipMemcpyDeviceToDevice));
        HIP_TRY(hipMemcpy(host + z * Y * X, ctx->stage, n * Y * X * sizeof(float), hipMemcpyDeviceToHost));
    }
    return FDTDX_OK;
}

extern "C" int fdtdx_transfer_cube_to_gpu(fdtdx_ctx *ctx, const char *name, const float *host)
{
    float *d;
    int rc = fdtdx_find_cube(ctx, name, &d);
    if (rc) return rc;
    if (!host) return fail(FDTDX_ERR_INVALID, "null host pointer");
    const fdtdx_geometry &g = ctx->g;
    return copy_planes_h2d(ctx, d, host + g.first_global * ctx->cfg.Y * ctx->cfg.X, g.local_planes);
}

extern "C" int fdtdx_upload_local(fdtdx_ctx *ctx, const char *name, const float *host_local)
{
    float *d;
    int rc = fdtdx_find_cube(ctx, name, &d);
    if (rc) return rc;
    if (!host_local) return fail(FDTDX_ERR_INVALID, "null host pointer");
    return copy_planes_h2d(ctx, d, host_local, ctx->g.local_planes);
}

extern "C" int fdtdx_transfer_cube_to_cpu(fdtdx_ctx *ctx, float *host, const char *name)
{
    float *d;
    int rc = fdtdx_find_cube(ctx, name, &d);
    if (rc) return rc;
    if (!host) return fail(FDTDX_ERR_INVALID, "null host pointer");
    const fdtdx_geometry &g = ctx->g;
    return copy_planes_d2h(ctx, host + g.owned_begin * ctx->cfg.Y * ctx->cfg.X, d + g.ghost_lo * g.plane, g.owned_planes);
}

extern "C" int fdtdx_download_owned(fdtdx_ctx *ctx, float *host_owned, const char *name)
{
    float *d;
    int rc = fdtdx_find_cube(ctx, name, &d);
    if (rc) return rc;
    if (!host_owned) return fail(FDTDX_ERR_INVALID, "null host pointer");
    const fdtdx_geometry &g = ctx->g;
    return copy_planes_d2h(ctx, host_owned, d + g.ghost_lo * g.plane, g.owned_planes);
}

// ---- driver initial conditions -------------------------------------------------------------------
extern "C" int fdtdx_init_mt(fdtdx_ctx *ctx, const char *name_p0, const char *name_p1)
{
    float *d0, *d1;
    int rc = fdtdx_find_cube(ctx, name_p0, &d0);
    if (rc) return rc;
    rc = fdtdx_find_cube(ctx, name_p1, &d1);
    if (rc) return rc;
    USE_DEVICE(ctx);
    const fdtdx_geometry &g = ctx->g;
    const uint64_t X = ctx->cfg.X, Y = ctx->cfg.Y;
    // stage a bounded number of planes at a time so host memory stays small for big slabs
    const uint64_t batch = std::max<uint64_t>(1, std::min<uint64_t>(g.local_planes, (256ull << 20) / (Y * X * sizeof(float))));
    std::vector<float> h0(batch * Y * X), h1(batch * Y * X);
    for (uint64_t zb = 0; zb < g.local_planes; zb += batch) {
        const uint64_t n = std::min(batch, g.local_planes - zb);
        // fdtd1.cu:21: the reference parallelises over planes with OpenMP; plain threads here
        auto fill = [&](uint64_t zi) {
            // fdtd1.cu:24-30: one generator per global plane, p0 then p1 per cell
            std::mt19937_64 gen(19950918ull + g.first_global + zb + zi);
            float *a = h0.data() + zi * Y * X, *b = h1.data() + zi * Y * X;
            for (uint64_t i = 0; i < Y * X; i++) {
                a[i] = float(gen()) / float(gen.max());
                b[i] = float(gen()) / float(gen.max());
            }
        };
        const unsigned nthr = std::max(1u, std::min<unsigned>(16u, std::min<unsigned>((unsigned)n, std::thread::hardware_concurrency())));
        std::vector<std::thread> pool;
        for (unsigned w = 0; w < nthr; w++)
            pool.emplace_back([&, w]() {
                for (uint64_t zi = w; zi < n; zi += nthr) fill(zi);
            });
        for (auto &th : pool) th.join();
        if ((rc = copy_planes_h2d(ctx, d0 + zb * g.plane, h0.data(), n))) return rc;
        if ((rc = copy_planes_h2d(ctx, d1 + zb * g.plane, h1.data(), n))) return rc;
    }
    return FDTDX_OK;
}

extern "C" int fdtdx_init_jacobi(fdtdx_ctx *ctx, const char *name)
{
    float *d;
    int rc = fdtdx_find_cube(ctx, name, &d);
    if (rc) return rc;
    USE_DEVICE(ctx);
    const fdtdx_geometry &g = ctx->g;
    const uint64_t X = ctx->cfg.X, Y = ctx->cfg.Y;
    const uint64_t batch = std::max<uint64_t>(1, std::min<uint64_t>(g.local_planes, (64ull << 20) / (Y * X * sizeof(float))));
    std::vector<float> h(batch * Y * X);
    for (uint64_t zb = 0; zb < g.local_planes; zb += batch) {
        const uint64_t n = std::min(batch, g.local_planes - zb);
        for (uint64_t z = 0; z < n; z++) {
            const uint64_t k = g.first_global + zb + z;
            for (uint64_t j = 0; j < Y; j++)
                for (uint64_t i = 0; i < X; i++)  // jacobi.cu:415: i - j + 1.0/(k+1) in double
                    h[(z * Y + j) * X + i] = (float)((double)((long long)i - (long long)j) + 1.0 / (double)(k + 1));
        }
        if ((rc = copy_planes_h2d(ctx, d + zb * g.plane, h.data(), n))) return rc;
    }
    return FDTDX_OK;
}

// ---- sweep dispatch (stencil.h:684-768) ------------------------------------------------------------
static int check_program(const fdtdx_ctx *ctx, const fdtdx_program *prog, int *radius)
{
    if (!prog) return fail(FDTDX_ERR_INVALID, "null program");
    int r = 1;
    if (prog->kind == FDTDX_STAR) {
        r = (int)prog->radius;
        if (r < 1 || r > 8) return fail(FDTDX_ERR_UNSUPPORTED, "star radius %d outside 1..8", r);
    } else if (prog->kind != FDTDX_SPC13 && prog->kind != FDTDX_JACOBI7) {
        return fail(FDTDX_ERR_UNSUPPORTED, "unknown program kind %u", prog->kind);
    }
    if (prog->order != FDTDX_ORDER_CHAIN && prog->order != FDTDX_ORDER_GOLDEN) return fail(FDTDX_ERR_INVALID, "unknown order %u", prog->order);
    if ((uint32_t)r > ctx->cfg.M) return fail(FDTDX_ERR_INVALID, "program radius %d exceeds the context's ghost width M=%u", r, ctx->cfg.M);
    *radius = r;
    return FDTDX_OK;
}

// Default tile shape per radius (quads x thread-rows x rows-per-thread).  Tile shapes measured fastest at 512^3 (tools/tune.py).  Radius <= 4 runs full-row tiles (no x
// halo at X <= 512) when the grid is wide enough to fill them; larger radii need the registers
// for the z queue and use 32-quad tiles.
static void default_shape(int kind, int r, int nq, int *tq, int *ly, int *ry, unsigned order = FDTDX_ORDER_CHAIN)
{
    if (kind == FDTDX_STAR && order == FDTDX_ORDER_GOLDEN && r >= 5) {
        // separate multiplies and adds: twice the arithmetic, and the 1024-lane shapes spill; 32-quad tiles on 512 lanes, at radius 5
        // with two adjacent rows per lane
        *tq = 32, *ly = 16, *ry = r == 5 ? 2 : 1;
        return;
    }
    if (kind == FDTDX_SPC13) {
        *tq = 64, *ly = 8, *ry = 2;
        return;
    }
    if (r == 3 && nq >= 128 && (nq % 64 == 0 || nq >= 512)) {
        // 64 x 16 cells-quads on 512 lanes (two rows per lane, two workgroups per CU): 1.4 % ahead of full-row tiles at 512 cells per
        // row, 3.5 % at 1024; rows that are not whole 64-quad tiles lose more to the ragged last tile (96 quads: 4 % behind)
        *tq = 64, *ly = 8, *ry = 2;
        return;
    }
    if (r <= 4) {
        if (nq > 64)
            *tq = 128, *ly = 8, *ry = 1;
        else if (nq > 32)
            *tq = 64, *ly = 16, *ry = 1;
        else
            *tq = 32, *ly = 32, *ry = 1;
        return;
    }
    switch (r) {
    case 5: *tq = 64, *ly = 16, *ry = 1; break;  // (launch_sweep prefers a 768-lane shape where its tiles fill the CUs)
    case 6: *tq = 32, *ly = 16, *ry = 2; break;  // 32 x 32 cells on 512 lanes: 2.5 % ahead of 64 x 16 in a time loop
    // radius 7 and 8: 32 x 32 tiles on 512 lanes, every lane two adjacent rows, late publish (star_pairs_late in star_kernel.h).  Against one
    // row per lane on 1024 lanes, on the drivers' fields: radius 8 +1.3 % over 100 steps; radius 7 +0.5 to +1.7 % over the first 20 steps on
    // three boxes, -1.5 to +0.6 % over 100
    case 7: *tq = 32, *ly = 16, *ry = 2; break;
    default: *tq = 32, *ly = 16, *ry = 2; break;
    }
}

// [z2_lo, z2_hi): a second plane range swept by the same launch (empty: none)
static int launch_sweep(fdtdx_ctx *ctx, float *out, const float *in, const fdtdx_program *prog, uint64_t z_lo, uint64_t z_hi, void *stream,
                        const fdtdx_composite *comp = nullptr, const float *vel = nullptr, float *sum = nullptr, const int64_t *sel = nullptr,
                        uint64_t z2_lo = 0, uint64_t z2_hi = 0, const float *shell = nullptr)
{
    int r;
    int rc;
    fdtdx_program cprog;
    if (comp) {  // composite update: the star machinery with another per-cell formula
        if (comp->radius < 1 || comp->radius > 8) return fail(FDTDX_ERR_UNSUPPORTED, "composite radius %u outside 1..8", comp->radius);
        memset(&cprog, 0, sizeof(cprog));
        cprog.kind = FDTDX_STAR, cprog.radius = comp->radius, cprog.order = comp->order;
        prog = &cprog;
    }
    rc = check_program(ctx, prog, &r);
    if (rc) return rc;
    USE_DEVICE(ctx);
    const fdtdx_geometry &g = ctx->g;
    if (z_lo < (uint64_t)r || z_hi > g.local_planes - r || z_lo > z_hi)
        return fail(FDTDX_ERR_INVALID, "plane range [%llu,%llu) outside [%d,%llu)", (unsigned long long)z_lo, (unsigned long long)z_hi, r, (unsigned long long)(g.local_planes - r));
    if (z2_lo < z2_hi && (z2_lo < z_hi || z2_hi > g.local_planes - r))
        return fail(FDTDX_ERR_INVALID, "second plane range [%llu,%llu) must lie behind the first and inside the cube", (unsigned long long)z2_lo, (unsigned long long)z2_hi);
    if (z_lo == z_hi) {  // only the second range has planes
        if (z2_lo >= z2_hi) return FDTDX_OK;
        z_lo = z2_lo, z_hi = z2_hi, z2_lo = z2_hi = 0;
    }
    if (out == in) return fail(FDTDX_ERR_INVALID, "sweep cannot run in place");

    SweepParams p;
    p.in = in;
    p.out = out;
    p.plane = (long long)g.plane;
    p.px = (int)g.pitch_x;
    p.pxq = (int)(g.pitch_x / 4);
    p.nx = (int)ctx->cfg.X;
    p.ny = (int)ctx->cfg.Y;
    p.h = r;
    p.z_lo = (int)z_lo;
    p.z_hi = (int)z_hi;
    memset(p.c, 0, sizeof(p.c));
    memcpy(p.c, prog->coef, 13 * sizeof(float));
    p.vel = vel;
    if (!shell && !comp && ctx->strip_radius == r)
        for (int i = 0; i < 2; i++)
            if (ctx->strip_cube[i] == out) shell = ctx->shell_strip[i];
    p.shell = shell, p.shell_plane = (long long)ctx->cfg.Y * 8;
    p.zg0 = (int)g.first_global;
    memset(p.kd, 0, sizeof(p.kd));
    p.sum = sum;
    for (int i = 0; i < 3; i++) p.sel[i] = sel ? (int)std::max<int64_t>(-1, std::min<int64_t>(sel[i], 0x7ffffffe)) : -1;
    if (comp) {
        const double kd[8] = {comp->cz, comp->cy, comp->cx, comp->scal_num, comp->scal_den, comp->wx, comp->wy, comp->wz};
        memcpy(p.kd, kd, sizeof(kd));
    }

    int tq, ly, ry, threads, lds, pf = 1;
    void (*launch)(const SweepParams &, dim3, hipStream_t) = nullptr;
    int dtq, dly, dry;
    default_shape(prog->kind, r, (int)((ctx->cfg.X + 3) / 4), &dtq, &dly, &dry, comp ? FDTDX_ORDER_CHAIN : prog->order);
    if (comp && ctx->cfg.X >= 128) dtq = 32, dly = 16, dry = 2;  // three or four input streams: measured 5-8 % ahead of full-row tiles
    if (!comp && prog->kind == FDTDX_STAR && (prog->order == FDTDX_ORDER_CHAIN ? (r == 4 || r == 5) : (r == 6 || r == 7))) {
        // 768-lane shapes (12 waves, 84 tiles x 3 z chunks on planes of 512 x 512 cells: a z chunk re-reads 2r planes, and three chunks
        // re-read fewer than four): radius 5 on 32 x 24 tiles 596 -> 622 Gcell/s, radius 4 on 64 x 12 tiles 637 -> 643 (radius 6: level;
        // 3, 7, 8: behind).  Only where the tiles fill the CUs: 344 tiles on planes of 1024 x 1024 need a second, mostly empty round
        // (radius 5: 461 against 620 Gcell/s), there the 1024-lane shape stays.  The host order takes 32 x 24 tiles at radius 6 and 7 (twelve
        // waves per CU instead of the eight of its 512-lane shape: 475 -> 505 and 427 -> 445 Gcell/s; at radius 8 they spill).
        const int ptq = r == 4 ? 64 : 32, ply = r == 4 ? 12 : 24;
        const long long cus = std::max<long long>(1, ctx->num_cus - ctx->opt_reserve_cus);
        const long long nz0 = std::max<long long>((long long)(z_hi - z_lo), z2_lo < z2_hi ? (long long)(z2_hi - z2_lo) : 0);
        const long long tiles = (long long)(((ctx->cfg.X + 3) / 4 + ptq - 1) / ptq) * (((long long)ctx->cfg.Y - 2 * r + ply - 1) / ply) * (z2_lo < z2_hi ? 2 : 1);
        double fill;
        if (tiles >= cus)
            fill = (double)tiles / (double)(((tiles + cus - 1) / cus) * cus);
        else
            fill = (double)(tiles * std::max<long long>(1, std::min(cus / tiles, std::max<long long>(1, nz0 / (12LL * r))))) / (double)cus;
        if (fill >= 0.95) dtq = ptq, dly = ply, dry = 1;
    }
    if (prog->kind == FDTDX_SPC13) {
        const SpcVariantInfo *best = nullptr;
        int idx = 0;
        for (int i = 0; i < num_spc_variants(); i++) {
            const SpcVariantInfo *v = spc_variant(i);
            if (v->order != (int)prog->order) continue;
            if (ctx->opt_variant >= 0 ? (idx == ctx->opt_variant) : (!best || (v->tq == dtq && v->ly == dly && v->ry == dry))) best = v;
            idx++;
        }
        if (!best) return fail(FDTDX_ERR_UNSUPPORTED, "no spc13 kernel for order %u", prog->order);
        tq = best->tq, ly = best->ly, ry = best->ry, threads = best->threads, lds = best->lds_bytes, launch = best->launch;
    } else {
        int ord = comp ? (prog->order == FDTDX_ORDER_CHAIN ? ORD_COMP_CHAIN : ORD_COMP_GOLDEN)
                  : (prog->kind == FDTDX_STAR) ? (prog->order == FDTDX_ORDER_CHAIN ? ORD_STAR_CHAIN : ORD_STAR_GOLDEN)
                                               : (prog->order == FDTDX_ORDER_CHAIN ? ORD_JACOBI_CHAIN : ORD_JACOBI_GOLDEN);
        const VariantInfo *best = nullptr;
        int idx = 0;
        for (int i = 0; i < num_variants(); i++) {
            const VariantInfo *v = variant(i);
            if (v->M != r || v->order != ord) continue;
            if (ctx->opt_variant >= 0 ? (idx == ctx->opt_variant) : (!best || (v->tq == dtq && v->ly == dly && v->ry == dry && v->pf == 1))) best = v;
            idx++;
        }
        if (!best) return fail(FDTDX_ERR_UNSUPPORTED, "no kernel for radius %d order %u variant %lld", r, prog->order, (long long)ctx->opt_variant);
        tq = best->tq, ly = best->ly, ry = best->ry, threads = best->threads, lds = best->lds_bytes, launch = best->launch;
        pf = best->pf;
    }

    const int nq = (p.nx + 3) / 4;                      // quads that hold logical cells
    const int rows = p.ny - 2 * r;                      // computed rows
    const int gx = (nq + tq - 1) / tq, gy = (rows + ly * ry - 1) / (ly * ry);
    const long long n2 = z2_lo < z2_hi ? (long long)(z2_hi - z2_lo) : 0;
    const long long nz = std::max<long long>((long long)(z_hi - z_lo), n2);  // chunks are sized for the longer range
    long long chunks = ctx->opt_zchunks;
    if (chunks <= 0) {
        // one workgroup per CU (256 CUs on MI355X) measured fastest: every extra z chunk re-reads 2r planes
        // to fill its queue, so split z only as far as needed to cover the chip, and keep the
        // fill below ~1/6 of a chunk
        const long long tiles = (long long)gx * gy * (n2 ? 2 : 1);
        const long long cus = std::max<long long>(1, ctx->num_cus - ctx->opt_reserve_cus);
        chunks = std::max<long long>(1, cus / tiles);  // never more workgroups than CUs: a second round would run nearly empty
        if (threads <= 256) chunks *= 2;  // small workgroups: two per CU
        const long long min_len = std::max<long long>(12LL * r, 16);
        // (a handful of planes -- the boundary planes of a slab -- are the exception: such a launch is all start-up latency,
        // and two chunks halve the planes a workgroup marches: 15.5 -> 11.3 us for 8 planes at radius 8)
        chunks = std::min(chunks, nz < min_len && nz >= 2 ? 2LL : std::max<long long>(1, nz / min_len));
        // Grids whose tiles fill the CUs badly in one round (176 tiles of a 700 x 800 plane on 256 CUs) do better with several rounds of
        // shorter chunks, which the CUs pick up as they come free: 600 x 700 x 800 at radius 5 463 -> 530 Gcell/s with 4 chunks, at
        // radius 8 402 -> 528 with 3.  Model: share of the CU slots of all rounds that hold a workgroup x share of a chunk's march that
        // is not queue fill (2r planes re-read + start-up).  It flatters many rounds (700 x 900 x 1000 at radius 3: +4 % predicted, 6 %
        // slower measured), so it only overrides the one-round rule where it promises 12 % or more; the tuned cases keep their chunks.
        const long long cap = cus * (threads <= 256 ? 2 : 1);
        auto eff = [&](long long cnt) {
            const long long len = (nz + cnt - 1) / cnt, n = tiles * ((nz + len - 1) / len);
            return (double)n / (double)((n + cap - 1) / cap * cap) * (double)len / (double)(len + 2 * r + 4);
        };
        const double e0 = eff(std::max<long long>(1, chunks));
        double ebest = e0;
        long long cbest = chunks;
        for (long long cnt = 1; cnt <= nz / min_len && cnt * tiles <= 8192; cnt++)
            if (eff(cnt) > ebest + 1e-9) ebest = eff(cnt), cbest = cnt;
        if (ebest > 1.12 * e0) chunks = cbest;
    }
    chunks = std::max<long long>(1, std::min(chunks, nz));
    p.zchunk = (int)((nz + chunks - 1) / chunks);
    p.gz1 = (int)(((long long)(z_hi - z_lo) + p.zchunk - 1) / p.zchunk);
    p.z_lo2 = (int)z2_lo, p.z_hi2 = (int)z2_hi;
    const int gz = p.gz1 + (int)((n2 + p.zchunk - 1) / p.zchunk);
    if ((long long)gx * gy * gz > 0x7fffffffLL) return fail(FDTDX_ERR_UNSUPPORTED, "grid too large (%d,%d,%d)", gx, gy, gz);

    p.gx = gx, p.gy = gy, p.gz = gz;
    launch(p, dim3((unsigned)gx * gy * gz), (hipStream_t)stream);
    HIP_TRY(hipGetLastError());
    ctx->launches++;
    snprintf(ctx->last_launch, sizeof(ctx->last_launch), "kind=%u r=%d order=%u tile=%dx%d(quads x rows) threads=%d ry=%d pf=%d lds=%d grid=%dx%dx%d zchunk=%d",
             prog->kind, r, prog->order, tq, ly * ry, threads, ry, pf, lds, gx, gy, gz, p.zchunk);
    return FDTDX_OK;
}

extern "C" int fdtdx_propagate_ptr(fdtdx_ctx *ctx, float *out, const float *in, const fdtdx_program *prog, uint64_t z_lo, uint64_t z_hi, void *stream)
{
    if (!ctx || !out || !in) return fail(FDTDX_ERR_INVALID, "null argument");
    return launch_sweep(ctx, out, in, prog, z_lo, z_hi, stream);
}

extern "C" int fdtdx_propagate_planes(fdtdx_ctx *ctx, const char *out, const char *in, const fdtdx_program *prog, uint64_t z_lo, uint64_t z_hi, void *stream)
{
    float *po, *pi;
    int rc = fdtdx_find_cube(ctx, out, &po);
    if (rc) return rc;
    rc = fdtdx_find_cube(ctx, in, &pi);
    if (rc) return rc;
    return launch_sweep(ctx, po, pi, prog, z_lo, z_hi, stream);
}

extern "C" int fdtdx_propagate(fdtdx_ctx *ctx, const char *out, const char *in, const fdtdx_program *prog, void *stream)
{
    if (!ctx) return fail(FDTDX_ERR_INVALID, "null context");
    int r;
    int rc = check_program(ctx, prog, &r);
    if (rc) return rc;
    // Planes this rank updates: stencil.h:690 (nz) -- every owned plane except the global shell.
    // Ghost planes are M deep, the program may be shallower (r <= M): the update range is
    // clipped to what the ghosts support.
    const fdtdx_geometry &g = ctx->g;
    uint64_t lo = g.ghost_lo ? g.ghost_lo : (uint64_t)r;
    uint64_t hi = g.ghost_hi ? g.local_planes - g.ghost_hi : g.local_planes - r;
    return fdtdx_propagate_planes(ctx, out, in, prog, lo, hi, stream);
}

// Fused steps (tb_kernel.h, tbm_kernel.h, tbs_kernel.h): out = f^kt(in), kt = 2 or 3, shell cells of the odd levels from `mid`.
static const TbVariantInfo *pick_tb(const fdtdx_ctx *ctx, const fdtdx_program *prog, int kt)
{
    const bool golden = prog->order == FDTDX_ORDER_GOLDEN;
    int ord;
    const int radius = prog->kind == FDTDX_STAR ? (int)prog->radius : 1;
    if (prog->kind == FDTDX_STAR)
        ord = golden ? ORD_STAR_GOLDEN : ORD_STAR_CHAIN;
    else if (prog->kind == FDTDX_JACOBI7)
        ord = golden ? ORD_JACOBI_GOLDEN : ORD_JACOBI_CHAIN;
    else if (prog->kind == FDTDX_SPC13) {
        ord = golden ? ORD_SPC_GOLDEN : ORD_SPC_CHAIN;
    } else
        return nullptr;
    // The fused 13-point kernel keeps ONE quad with x shell cells per lane and row.  A lane holds the quads 64 apart, so a
    // row whose first and last quad (both with a shell cell) are 64 apart (rows of exactly 65 quads) does not fit; such
    // grids run single steps.
    const bool shells_fit = prog->kind != FDTDX_SPC13 || (ctx->cfg.X + 3) / 4 != 65;
    // The two-step kernels of radius 3 and 4 exist for measurement: a tile of 24 rows x 64 quads stores 12 (radius 3) or 8
    // (radius 4) rows and 60 quads, so they compute 2-3 times what they store and run slower than single steps.  They are
    // only picked when the "fuse_wide" option asks for them.
    if (radius >= 3 && !ctx->opt_fuse_wide) return nullptr;
    const TbVariantInfo *best = nullptr;
    int idx = 0;
    for (int i = 0; i < num_tb_variants(); i++) {
        const TbVariantInfo *v = tb_variant(i);
        if (v->order != ord || v->kt != kt || v->radius != radius || v->seg != 1) continue;  // (strip shapes only ever run a row's tail)
        if (v->shells && !shells_fit) continue;
        if (ctx->opt_tb_variant >= 0 ? idx == ctx->opt_tb_variant : !best) best = v;
        idx++;
    }
    return best;
}

// for the multi-slab loop: is there a fused kernel of kt steps for the program on this grid (one that takes both shell sources by name)?
bool fdtdx_has_fused(const fdtdx_ctx *ctx, const fdtdx_program *prog, int kt, bool shells)
{
    const TbVariantInfo *v = pick_tb(ctx, prog, kt);
    return v && (!shells || v->shells);
}

extern "C" int fdtdx_fused_depth(const fdtdx_ctx *ctx, const fdtdx_program *prog, int named_shells)
{
    if (!ctx) return -fail(FDTDX_ERR_INVALID, "null context");
    int r;
    const int rc = check_program(ctx, prog, &r);
    if (rc) return -rc;
    for (int kt = 3; kt >= 2; kt--)
        if (fdtdx_has_fused(ctx, prog, kt, named_shells != 0)) return kt;
    return 0;
}

static int launch_tb(fdtdx_ctx *ctx, const TbVariantInfo *v, float *out, const float *in, const float *mid, const fdtdx_program *prog,
                     uint64_t z_lo, uint64_t z_hi, void *stream, const float *even = nullptr, uint64_t z2_lo = 0, uint64_t z2_hi = 0)
{
    const fdtdx_geometry &g = ctx->g;
    if (z_lo >= z_hi) {  // only the second range (if any) has planes
        if (z2_lo >= z2_hi) return FDTDX_OK;
        z_lo = z2_lo, z_hi = z2_hi, z2_lo = z2_hi = 0;
    }
    USE_DEVICE(ctx);
    TbParams p;
    p.in = in, p.mid = mid, p.out = out;
    p.even = even ? even : in;
    if (p.even != in && !v->shells) return fail(FDTDX_ERR_UNSUPPORTED, "this fused kernel takes its level-0 shell from `in` only");
    p.plane = (long long)g.plane;
    p.px = (int)g.pitch_x, p.pxq = (int)(g.pitch_x / 4);
    p.nx = (int)ctx->cfg.X, p.ny = (int)ctx->cfg.Y, p.nz = (int)g.local_planes;
    p.zg0 = (int)g.first_global, p.nzg = (int)ctx->cfg.Z;
    p.z_lo = (int)z_lo, p.z_hi = (int)z_hi;
    memcpy(p.c, prog->coef, 13 * sizeof(float));
    const int reach = v->kt * v->radius;  // cells a fused launch reaches on each side
    const int owned = v->nw * v->ry - 2 * reach;
    p.gy = (p.ny - 2 * v->radius + owned - 1) / owned;
    const int nq = (p.nx + 3) / 4, tq = 64 * v->rx;
    // Rows wider than one tile are cut into x tiles that overlap by `ovq` quads on each side (a tile stores
    // tq - 2 ovq quads; ovq = 1 for every fused kernel of radius 1 and 2).  Where the last tile has at most 62 quads left to store and the shape exists at half
    // the width, that tile is launched separately with the half-width kernel: rows of 1024 cells then
    // cost 2.5 tile widths instead of 3.
    const TbVariantInfo *narrow = nullptr;
    int gx_full = 1;
    const int ovq = (reach + 3) / 4;  // quads an x tile shares with its neighbour on each side: what kt steps reach into
    if (nq <= tq) {
        p.gx = 1, p.qstep = tq, p.qbase = 0, p.xt = 0;
    } else {
        p.qstep = tq - 2 * ovq, p.qbase = -ovq, p.xt = 1;
        gx_full = (nq + p.qstep - 1) / p.qstep;
        p.gx = gx_full;
        const int rem = nq - (gx_full - 1) * p.qstep;
        if (rem <= 64 - 2 * ovq && v->rx == 2 && ctx->opt_tb_variant < 0)
            for (int i = 0; i < num_tb_variants(); i++) {
                const TbVariantInfo *c = tb_variant(i);
                if (c->order != v->order || c->kt != v->kt || c->radius != v->radius || c->nw != v->nw || c->ry != v->ry || c->rx != 1) continue;
                // the narrowest shape the tail fits: a strip (segments of 64/seg quads, each in another y tile) before the half-width tile
                if (c->seg == 1 ? !narrow : (rem <= 64 / c->seg - 2 * ovq && !ctx->opt_no_strip && (!narrow || c->seg > narrow->seg))) narrow = c;
            }
        if (narrow) p.gx = gx_full - 1;
    }
    const long long n1 = p.z_hi - p.z_lo, n2 = z2_lo < z2_hi ? (long long)(z2_hi - z2_lo) : 0;
    const long long nz = std::max(n1, n2);  // chunks are sized for the longer range
    p.z_lo2 = (int)z2_lo, p.z_hi2 = (int)z2_hi;
    auto launch_part = [&](const TbVariantInfo *k, TbParams q) -> int {
        long long chunks = ctx->opt_tb_zchunks;
        if (chunks <= 0) {
            // The fused kernel runs one workgroup per CU (registers, LDS), so N workgroups take
            // ceil(N/CUs) rounds of the device's CUs, and every chunk pays for filling its queues (the planes a
            // launch reaches beyond its range are re-read, plus start-up: about 6 planes' worth in all, from
            // timings at 512^3 and 1024^3): take the chunk count with the best product of the two efficiencies.
            const long long tiles = (long long)q.gx * q.gy * (n2 ? 2 : 1);
            const long long cus = std::max<long long>(1, ctx->num_cus - ctx->opt_reserve_cus);
            double best = -1.0;
            chunks = 1;
            for (long long cnt = 1; cnt <= std::max<long long>(1, nz / 16) && cnt * tiles <= 4096; cnt++) {
                const long long len = (nz + cnt - 1) / cnt, n = tiles * ((nz + len - 1) / len);
                const double eff = (double)n / (double)((n + cus - 1) / cus * cus) * (double)len / (double)(len + 4 + reach);
                if (eff > best + 1e-9) best = eff, chunks = cnt;
            }
        }
        chunks = std::max<long long>(1, std::min(chunks, nz));
        q.zchunk = (int)((nz + chunks - 1) / chunks);
        q.gz1 = (int)((n1 + q.zchunk - 1) / q.zchunk);
        q.gz = q.gz1 + (int)((n2 + q.zchunk - 1) / q.zchunk);
        if ((long long)q.gx * q.gy * q.gz > 0x7fffffffLL) return fail(FDTDX_ERR_UNSUPPORTED, "grid too large");
        k->launch(q, dim3((unsigned)(q.gx * q.gy * q.gz)), (hipStream_t)stream);
        HIP_TRY(hipGetLastError());
        p.gz = q.gz, p.zchunk = q.zchunk;
        return FDTDX_OK;
    };
    int rc;
    if (narrow) {  // first the half-width tail (its grid is reported last otherwise), then the full-width tiles
        TbParams t = p;
        t.gx = 1, t.qbase = -ovq + (gx_full - 1) * p.qstep;
        t.gy = (p.gy + narrow->seg - 1) / narrow->seg;
        if ((rc = launch_part(narrow, t))) return rc;
    }
    if ((rc = launch_part(v, p))) return rc;
    ctx->launches++;
    snprintf(ctx->last_fused, sizeof(ctx->last_fused), "fused%d kind=%u r=%d waves=%d rows/wave=%d quads/lane/row=%d tile_rows=%d(stored %d) threads=%d lds=%d grid=%dx%dx%d%s zchunk=%d",
             v->kt, prog->kind, v->radius, v->nw, v->ry, v->rx, v->nw * v->ry, owned, v->threads, v->lds_bytes, p.gx, p.gy, p.gz,
             !narrow ? "" : narrow->seg > 1 ? "+strip tail" : "+half-width tail", p.zchunk);
    return FDTDX_OK;
}

// out = f^nsteps(in*), in* = in with the shell of `even`; the odd intermediate levels take their shell from `odd`,
// the even ones from `even`; the shell of `out` is not written
static int fused_launch_checked(fdtdx_ctx *ctx, const char *out, const char *in, const char *even, const char *odd, const fdtdx_program *prog,
                                uint32_t nsteps, uint64_t z_lo, uint64_t z_hi, void *stream, uint64_t z2_lo = 0, uint64_t z2_hi = 0)
{
    if (!ctx) return fail(FDTDX_ERR_INVALID, "null context");
    int r;
    int rc = check_program(ctx, prog, &r);
    if (rc) return rc;
    if (nsteps != 2 && nsteps != 3) return fail(FDTDX_ERR_UNSUPPORTED, "fused sweeps advance 2 or 3 steps, not %u", nsteps);
    float *po, *pi, *pe, *pm;
    if ((rc = fdtdx_find_cube(ctx, out, &po)) || (rc = fdtdx_find_cube(ctx, in, &pi)) || (rc = fdtdx_find_cube(ctx, even, &pe)) || (rc = fdtdx_find_cube(ctx, odd, &pm)))
        return rc;
    const TbVariantInfo *tb = pick_tb(ctx, prog, (int)nsteps);
    if (!tb)
        return fail(FDTDX_ERR_UNSUPPORTED, "no fused kernel for this program (radius 1 star/jacobi: 2 or 3 steps; radius 2 star, 13-point: 2 steps)");
    // a launch reads interior cells of `in` and shell cells of `even`/`odd`, and writes interior cells of `out`: `out` may be one of the
    // shell sources, never `in`.  Kernels that rewrite the shell cells of partly stored quads need `out` to carry the shell of its level.
    if (po == pi) return fail(FDTDX_ERR_INVALID, "fused sweep cannot write the cube it reads");
    if (!tb->shells && (pe != pi || pm == pi || (po == pm && nsteps == 2)))
        return fail(FDTDX_ERR_UNSUPPORTED, "this fused kernel takes the even levels' shell from `in` and needs `out` to hold the shell of its level");
    const fdtdx_geometry &g = ctx->g;
    // k steps of radius r need k*r planes of support on each side: ghost planes, or the global shell (r planes)
    const uint64_t reach = (uint64_t)nsteps * r;
    const uint64_t need_lo = g.ghost_lo ? reach : r, need_hi = g.ghost_hi ? reach : r;
    if ((g.ghost_lo && g.ghost_lo < reach) || (g.ghost_hi && g.ghost_hi < reach) || z_lo < need_lo || z_hi > g.local_planes - need_hi || z_lo > z_hi)
        return fail(FDTDX_ERR_INVALID, "fused plane range [%llu,%llu) needs %llu ghost planes towards a neighbour", (unsigned long long)z_lo, (unsigned long long)z_hi, (unsigned long long)reach);
    if (z2_lo < z2_hi && (z2_lo < z_hi || z2_hi > g.local_planes - need_hi))
        return fail(FDTDX_ERR_INVALID, "second fused plane range [%llu,%llu) must lie behind the first and inside the cube", (unsigned long long)z2_lo, (unsigned long long)z2_hi);
    return launch_tb(ctx, tb, po, pi, pm, prog, z_lo, z_hi, stream, pe, z2_lo, z2_hi);
}

// One launch over two plane ranges (the planes next to either neighbour of a slab): for the multi-slab step.
extern "C" int fdtdx_propagate_two_ranges(fdtdx_ctx *ctx, const char *out, const char *in, const fdtdx_program *prog, uint64_t z_lo, uint64_t z_hi,
                               uint64_t z2_lo, uint64_t z2_hi, void *stream)
{
    float *po, *pi;
    int rc;
    if ((rc = fdtdx_find_cube(ctx, out, &po)) || (rc = fdtdx_find_cube(ctx, in, &pi))) return rc;
    return launch_sweep(ctx, po, pi, prog, z_lo, z_hi, stream, nullptr, nullptr, nullptr, nullptr, z2_lo, z2_hi);
}
extern "C" int fdtdx_propagate_fused_two_ranges(fdtdx_ctx *ctx, const char *out, const char *in, const char *even, const char *odd, const fdtdx_program *prog,
                                     uint32_t nsteps, uint64_t z_lo, uint64_t z_hi, uint64_t z2_lo, uint64_t z2_hi, void *stream)
{
    return fused_launch_checked(ctx, out, in, even, odd, prog, nsteps, z_lo, z_hi, stream, z2_lo, z2_hi);
}

extern "C" int fdtdx_propagate_fused_planes(fdtdx_ctx *ctx, const char *out, const char *in, const char *mid, const fdtdx_program *prog,
                                            uint32_t nsteps, uint64_t z_lo, uint64_t z_hi, void *stream)
{
    return fused_launch_checked(ctx, out, in, in, mid, prog, nsteps, z_lo, z_hi, stream);
}

extern "C" int fdtdx_propagate_fused_shells(fdtdx_ctx *ctx, const char *out, const char *in, const char *shell_even, const char *shell_odd,
                                            const fdtdx_program *prog, uint32_t nsteps, uint64_t z_lo, uint64_t z_hi, void *stream)
{
    return fused_launch_checked(ctx, out, in, shell_even, shell_odd, prog, nsteps, z_lo, z_hi, stream);
}

extern "C" int fdtdx_propagate2_planes(fdtdx_ctx *ctx, const char *out, const char *in, const char *mid, const fdtdx_program *prog,
                                       uint64_t z_lo, uint64_t z_hi, void *stream)
{
    return fdtdx_propagate_fused_planes(ctx, out, in, mid, prog, 2, z_lo, z_hi, stream);
}

// The quads of a row that straddle the x shell (at most one on each side: quads ql and qr, -1: none), copied into a strip of
// two quads per row: what a time loop's single-step sweeps merge into those quads instead of re-reading them from the
// output cube every step (64-byte sectors for 16 bytes: 65 MB per 512^3 sweep, 6 % of its time at radius 7).
__global__ void gather_shell_kernel(const float *cube, float *strip, long long plane, int px, int ny, long long rows, int ql, int qr)
{
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= 2 * rows) return;
    const long long row = i >> 1;
    const int q = (i & 1) ? qr : ql;
    const long long z = row / ny, y = row % ny;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (q >= 0) v = *reinterpret_cast<const float4 *>(cube + z * plane + y * px + 4 * q);
    reinterpret_cast<float4 *>(strip)[i] = v;
}

// fills ctx->shell_strip[i] from cube[i] for a program of radius r; false: no strips (not allocated, or no such quads)
bool fdtdx_strips_bind(fdtdx_ctx *ctx, const float *a, const float *b, int r, void *stream)
{
    const float *cube[2] = {a, b};
    ctx->strip_radius = 0;
    if (hipSetDevice(ctx->device) != hipSuccess) return false;
    const int nx = (int)ctx->cfg.X;
    const int ql = (r % 4 != 0) ? r / 4 : -1, qr = ((nx - r) % 4 != 0) ? (nx - r) / 4 : -1;
    if (!ctx->shell_strip[0] || !ctx->shell_strip[1] || (ql < 0 && qr < 0) || nx < 2 * r + 1) return false;
    const long long rows = (long long)ctx->g.local_planes * (long long)ctx->cfg.Y;
    if (2 * rows > 0x7fffffffLL * 256) return false;
    for (int i = 0; i < 2; i++)
        hipLaunchKernelGGL(gather_shell_kernel, dim3((unsigned)((2 * rows + 255) / 256)), dim3(256), 0, (hipStream_t)stream, cube[i], ctx->shell_strip[i],
                           (long long)ctx->g.plane, (int)ctx->g.pitch_x, (int)ctx->cfg.Y, rows, ql, qr);
    if (hipGetLastError() != hipSuccess) return false;
    ctx->strip_cube[0] = a, ctx->strip_cube[1] = b, ctx->strip_radius = r;
    return true;
}
void fdtdx_strips_release(fdtdx_ctx *ctx) { ctx->strip_radius = 0; }

// `work`: a third cube for the two-step loops that need one (nullptr: allocated on first use and kept by the context)
static int sweep_loop(fdtdx_ctx *ctx, const char *a, const char *b, const char *work, bool may_allocate, const fdtdx_program *prog,
                      uint32_t steps, void *stream)
{
    if (!ctx) return fail(FDTDX_ERR_INVALID, "null context");
    if (ctx->cfg.nprocs != 1) return fail(FDTDX_ERR_UNSUPPORTED, "fdtdx_sweep runs whole time loops on one rank only; use propagate + halo exchange per step");
    int r;
    int rc = check_program(ctx, prog, &r);
    if (rc) return rc;
    float *pa, *pb;
    rc = fdtdx_find_cube(ctx, a, &pa);
    if (rc) return rc;
    rc = fdtdx_find_cube(ctx, b, &pb);
    if (rc) return rc;
    if (pa == pb) return fail(FDTDX_ERR_INVALID, "sweep needs two different cubes");
    float *pw = nullptr;
    if (work) {
        rc = fdtdx_find_cube(ctx, work, &pw);
        if (rc) return rc;
        if (pw == pa || pw == pb) return fail(FDTDX_ERR_INVALID, "the work cube must be a third cube");
    }
    const fdtdx_geometry &g = ctx->g;
    const uint64_t lo = r, hi = g.local_planes - r;
    // Level t lives in b for even t and in a for odd t (step 0 writes a = f(b)), and so does its shell: even levels see b's
    // shell, odd levels a's.  The last one to four steps are always single steps so that a and b end holding exactly the two
    // newest levels, as after the reference's loop.
    const int want = (int)ctx->opt_temporal;  // 0 off, 1 auto (deepest), 2 / 3 steps per launch
    const TbVariantInfo *tb = nullptr;
    if (steps >= 4 && want != 0) {
        if (want != 2) tb = pick_tb(ctx, prog, 3);
        if (!tb && want != 3) tb = pick_tb(ctx, prog, 2);
    }
    uint32_t done = 0;
    if (tb && tb->kt == 3) {
        // Three steps per launch hop straight between the two cubes: level t+3 belongs in the
        // other cube, whose shell is also the one levels t+1 and t+3 see.
        const uint32_t hops = (steps - 1) / 3;
        for (uint32_t k = 0; k < hops; k++) {
            float *out = (done % 2 == 0) ? pa : pb;
            const float *in = (done % 2 == 0) ? pb : pa;
            rc = launch_tb(ctx, tb, out, in, out, prog, lo, hi, stream);
            if (rc) return rc;
            done += 3;
        }
    } else if (tb && tb->shells) {
        // Two steps per launch, in place between the two cubes: b -> a -> b ...  Every launch takes the even levels' shell
        // from b and the odd levels' from a and writes interior cells only, so after the first hop a holds level 2 inside
        // its own (odd) shell; the next hop reads that interior with b's shell in its place.  Hops come in pairs, so that
        // the newest even level ends up in b, inside the shell that belongs to it.
        const uint32_t hops = 2 * ((steps - 1) / 4);
        for (uint32_t k = 0; k < hops; k++) {
            float *out = (k % 2 == 0) ? pa : pb;
            const float *in = (k % 2 == 0) ? pb : pa;
            rc = launch_tb(ctx, tb, out, in, pa, prog, lo, hi, stream, pb);
            if (rc) return rc;
            done += 2;
        }
    } else if (tb) {
        // Two steps per launch hop between b and a third cube c that carries b's shell; a only
        // lends its shell to the skipped odd levels.
        uint32_t pairs = (steps % 2 == 0) ? steps / 2 - 1 : ((steps - 1) / 2) & ~1u;
        if (pairs > 0 && !pw) {
            const size_t bytes = g.cube_elems * sizeof(float);
            USE_DEVICE(ctx);
            if (!may_allocate) {
                pairs = 0;  // no third cube named: single steps (same results)
            } else if (!ctx->tb_scratch && hipMalloc((void **)&ctx->tb_scratch, bytes) != hipSuccess) {
                // no room for the third cube: run the loop as single steps (same results)
                (void)hipGetLastError();
                ctx->tb_scratch = nullptr;
                pairs = 0;
            }
            pw = ctx->tb_scratch;
        }
        if (pairs > 0) {
            const size_t bytes = g.cube_elems * sizeof(float);
            USE_DEVICE(ctx);
            HIP_TRY(hipMemcpyAsync(pw, pb, bytes, hipMemcpyDeviceToDevice, (hipStream_t)stream));
            float *cur = pb, *oth = pw;
            for (uint32_t k = 0; k < pairs; k++) {
                rc = launch_tb(ctx, tb, oth, cur, pa, prog, lo, hi, stream);
                if (rc) return rc;
                std::swap(cur, oth);
            }
            done = 2 * pairs;
            if (cur != pb) {
                // even step count, odd number of pairs: the newest even level sits in the scratch
                // cube; the next single step reads it from there (b is rewritten afterwards)
                rc = launch_sweep(ctx, pa, cur, prog, lo, hi, stream);
                if (rc) return rc;
                done++;
            }
        }
    }
    // remaining single steps: level `done` is in b if done is even, in a if odd.  Where there are more than a few, the
    // star sweeps take the shell components of partly stored quads from strips gathered once, here.
    if (prog->kind != FDTDX_SPC13 && steps - done >= 8) (void)fdtdx_strips_bind(ctx, pa, pb, r, stream);
    for (uint32_t t = done; t < steps && !rc; t++) {
        float *out = (t % 2 == 0) ? pa : pb;
        const float *in = (t % 2 == 0) ? pb : pa;
        rc = launch_sweep(ctx, out, in, prog, lo, hi, stream);
    }
    fdtdx_strips_release(ctx);
    return rc;
}

extern "C" int fdtdx_sweep(fdtdx_ctx *ctx, const char *a, const char *b, const fdtdx_program *prog, uint32_t steps, void *stream)
{
    return sweep_loop(ctx, a, b, nullptr, true, prog, steps, stream);
}

extern "C" int fdtdx_sweep_work(fdtdx_ctx *ctx, const char *a, const char *b, const char *work, const fdtdx_program *prog, uint32_t steps,
                                void *stream)
{
    return sweep_loop(ctx, a, b, work, false, prog, steps, stream);
}

// ---- the composite pipeline of main.cu: sweep, inject, filt, mul ---------------------------------------
static int owned_range(const fdtdx_ctx *ctx, uint64_t z_lo, uint64_t z_hi, const char *what)
{
    const fdtdx_geometry &g = ctx->g;
    if (z_lo > z_hi || z_hi > g.local_planes) return fail(FDTDX_ERR_INVALID, "%s: plane range [%llu,%llu) outside the local cube", what, (unsigned long long)z_lo, (unsigned long long)z_hi);
    return FDTDX_OK;
}

extern "C" int fdtdx_propagate_composite_planes(fdtdx_ctx *ctx, const char *out, const char *in, const char *vel, const fdtdx_composite *comp,
                                                uint64_t z_lo, uint64_t z_hi, void *stream)
{
    if (!ctx || !comp) return fail(FDTDX_ERR_INVALID, "null argument");
    float *po, *pi, *pv;
    int rc;
    if ((rc = fdtdx_find_cube(ctx, out, &po)) || (rc = fdtdx_find_cube(ctx, in, &pi)) || (rc = fdtdx_find_cube(ctx, vel, &pv))) return rc;
    if (po == pv) return fail(FDTDX_ERR_INVALID, "the composite update cannot write its velocity cube");
    if (comp->scal_den == 0.0) return fail(FDTDX_ERR_INVALID, "scal_den is zero");
    return launch_sweep(ctx, po, pi, nullptr, z_lo, z_hi, stream, comp, pv);
}

extern "C" int fdtdx_propagate_composite(fdtdx_ctx *ctx, const char *out, const char *in, const char *vel, const fdtdx_composite *comp, void *stream)
{
    if (!ctx || !comp) return fail(FDTDX_ERR_INVALID, "null argument");
    const fdtdx_geometry &g = ctx->g;
    const uint64_t r = comp->radius;
    const uint64_t lo = g.ghost_lo ? g.ghost_lo : r;
    const uint64_t hi = g.ghost_hi ? g.local_planes - g.ghost_hi : g.local_planes - r;
    return fdtdx_propagate_composite_planes(ctx, out, in, vel, comp, lo, hi, stream);
}

extern "C" int fdtdx_malloc_addr(fdtdx_ctx *ctx, const char *name, uint64_t capacity)
{
    if (!ctx || !name) return fail(FDTDX_ERR_INVALID, "null argument");
    if (ctx->addr.count(name)) return fail(FDTDX_ERR_EXISTS, "address list '%s' exists (stencil.h:443)", name);
    fdtdx_ctx::AddrList a;
    a.capacity = capacity;
    HIP_TRY(hipSetDevice(ctx->device));
    if (capacity) HIP_TRY(hipMalloc((void **)&a.ptr, capacity * sizeof(uint64_t)));
    ctx->addr[name] = a;
    return FDTDX_OK;
}

// stencil.h:518-544: keep the addresses whose plane this rank owns and re-base them to the local padded cube
extern "C" int fdtdx_transfer_addr_to_gpu(fdtdx_ctx *ctx, const char *name, const uint64_t *global_addr, uint64_t n)
{
    if (!ctx || !name || (!global_addr && n)) return fail(FDTDX_ERR_INVALID, "null argument");
    auto it = ctx->addr.find(name);
    if (it == ctx->addr.end()) return fail(FDTDX_ERR_NOTFOUND, "no address list '%s'", name);
    const fdtdx_geometry &g = ctx->g;
    const uint64_t X = ctx->cfg.X, Y = ctx->cfg.Y, Z = ctx->cfg.Z;
    std::vector<uint64_t> local;
    for (uint64_t i = 0; i < n; i++) {
        const uint64_t a = global_addr[i];
        const uint64_t z = a / (Y * X), y = (a % (Y * X)) / X, x = a % X;
        if (z >= Z) return fail(FDTDX_ERR_INVALID, "address %llu outside the grid", (unsigned long long)a);
        if (z < g.owned_begin || z >= g.owned_begin + g.owned_planes) continue;
        local.push_back((z - g.first_global) * g.plane + y * g.pitch_x + x);
    }
    if (local.size() > it->second.capacity) return fail(FDTDX_ERR_INVALID, "%zu addresses exceed the capacity %zu of '%s'", local.size(), it->second.capacity, name);
    HIP_TRY(hipSetDevice(ctx->device));
    if (!local.empty()) HIP_TRY(hipMemcpy(it->second.ptr, local.data(), local.size() * sizeof(uint64_t), hipMemcpyHostToDevice));
    it->second.count = local.size();
    return FDTDX_OK;
}

extern "C" int fdtdx_addr_size(fdtdx_ctx *ctx, const char *name, uint64_t *count)
{
    if (!ctx || !name || !count) return fail(FDTDX_ERR_INVALID, "null argument");
    auto it = ctx->addr.find(name);
    *count = (it == ctx->addr.end()) ? 0 : it->second.count;  // getAddrSize (stencil.h:500-504): 0 for an unknown name
    return FDTDX_OK;
}

extern "C" int fdtdx_get_addr(fdtdx_ctx *ctx, const char *name, uint64_t **device_ptr)
{
    if (!ctx || !name || !device_ptr) return fail(FDTDX_ERR_INVALID, "null argument");
    auto it = ctx->addr.find(name);
    if (it == ctx->addr.end()) return fail(FDTDX_ERR_NOTFOUND, "no address list named '%s'", name);
    *device_ptr = it->second.ptr;  // get<size_t>(name) of stencil.h:479-484: borrowed
    return FDTDX_OK;
}

__global__ void inject_planes_kernel(float *cube, const uint64_t *addr, uint64_t n, float add, uint64_t plane, uint64_t z_lo, uint64_t z_hi)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint64_t a = addr[i], z = a / plane;
    if (z >= z_lo && z < z_hi) atomicAdd(cube + a, add);
}

extern "C" int fdtdx_inject(fdtdx_ctx *ctx, const char *cube, const char *addr_name, float add, uint64_t z_lo, uint64_t z_hi, void *stream)
{
    float *pc;
    int rc = fdtdx_find_cube(ctx, cube, &pc);
    if (rc) return rc;
    if (!addr_name) return fail(FDTDX_ERR_INVALID, "null argument");
    if ((rc = owned_range(ctx, z_lo, z_hi, "inject"))) return rc;
    auto it = ctx->addr.find(addr_name);
    if (it == ctx->addr.end()) return fail(FDTDX_ERR_NOTFOUND, "no address list '%s'", addr_name);
    const uint64_t n = it->second.count;
    if (n == 0 || z_lo == z_hi) return FDTDX_OK;
    USE_DEVICE(ctx);
    hipLaunchKernelGGL(inject_planes_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, pc, it->second.ptr, n, add,
                       ctx->g.plane, z_lo, z_hi);
    HIP_TRY(hipGetLastError());
    return FDTDX_OK;
}

// filt (stencil.h:817-823 with main.cu:129-133's selection) touches three thin sets of the planes [z_lo, z_lo+nz):
// every row of the selected z plane and the selected row of every other plane (one wave per row: blocks from
// `cell_blocks` on), and one cell of every remaining row (one LANE per row: blocks below `cell_blocks`).
// With `sum` set, the scaled cells that are interior cells also get their product added (sum += cube*other):
// the fused step leaves exactly these cells to this kernel.
__global__ void filt_planes_kernel(float *cube, long long plane, int px, int nx, int ny, long long zg0, long long z_lo, long long nz,
                                   long long zsel, long long ysel, long long xsel, double scale, int golden, float *sum, const float *other,
                                   int h, long long mz_lo, long long mz_hi, unsigned cell_blocks)
{
    const float sf = (float)scale;
    const long long zl = zsel - zg0;  // the selected plane in local numbering
    const bool zin = zl >= z_lo && zl < z_lo + nz, yin = ysel >= 0 && ysel < ny;
    auto one = [&](long long z, int y, int x) {
        const long long a = z * plane + (long long)y * px + x;
        const float v = golden ? (float)((double)cube[a] * scale) : cube[a] * sf;
        cube[a] = v;
        if (sum && z >= mz_lo && z < mz_hi && y >= h && y < ny - h && x >= h && x < nx - h)
            sum[a] = golden ? sum[a] + v * other[a] : __builtin_fmaf(v, other[a], sum[a]);
    };
    if (blockIdx.x < cell_blocks) {
        const long long row = (long long)blockIdx.x * blockDim.x + threadIdx.x;
        if (row >= nz * ny || xsel < 0 || xsel >= nx) return;
        const long long z = z_lo + row / ny;
        const int y = (int)(row % ny);
        if (z != zl && y != ysel) one(z, y, (int)xsel);
        return;
    }
    const long long nwz = zin ? ny : 0;
    const long long w = (long long)(blockIdx.x - cell_blocks) * (blockDim.x / 64) + (threadIdx.x >> 6);
    long long z;
    int y;
    if (w < nwz) {
        z = zl, y = (int)w;
    } else {
        const long long q = w - nwz;  // q-th plane of the range that is not the selected one
        if (!yin || q >= nz - (zin ? 1 : 0)) return;
        z = z_lo + q + ((zin && z_lo + q >= zl) ? 1 : 0), y = (int)ysel;
    }
    for (int x = threadIdx.x & 63; x < nx; x += 64) one(z, y, x);
}

static int launch_filt(fdtdx_ctx *ctx, float *pc, uint64_t z_lo, uint64_t z_hi, int64_t zsel, int64_t ysel, int64_t xsel, double scale, int golden,
                       float *sum, const float *other, int h, uint64_t mz_lo, uint64_t mz_hi, hipStream_t s)
{
    USE_DEVICE(ctx);
    const fdtdx_geometry &g = ctx->g;
    const long long nz = (long long)(z_hi - z_lo), ny = (long long)ctx->cfg.Y;
    const long long cell_blocks = (nz * ny + 255) / 256, row_blocks = (ny + nz + 3) / 4;
    if (cell_blocks + row_blocks > 0x7fffffffLL) return fail(FDTDX_ERR_UNSUPPORTED, "grid too large");
    hipLaunchKernelGGL(filt_planes_kernel, dim3((unsigned)(cell_blocks + row_blocks)), dim3(256), 0, s, pc, (long long)g.plane, (int)g.pitch_x,
                       (int)ctx->cfg.X, (int)ny, (long long)g.first_global, (long long)z_lo, nz, (long long)zsel, (long long)ysel, (long long)xsel,
                       scale, golden, sum, other, h, (long long)mz_lo, (long long)mz_hi, (unsigned)cell_blocks);
    HIP_TRY(hipGetLastError());
    return FDTDX_OK;
}

__global__ void gather_kernel(float *saved, const float *cube, const uint64_t *addr, uint64_t n)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) saved[i] = cube[addr[i]];
}

// the cells inject changed after the fused sweep had already added their product: redo it from the saved sum
__global__ void redo_product_kernel(float *sum, const float *saved, const float *cube, const float *other, const uint64_t *addr, uint64_t n,
                                    long long plane, int px, int nx, int ny, int h, long long mz_lo, long long mz_hi, long long zg0,
                                    long long zsel, long long ysel, long long xsel, int golden)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const long long a = (long long)addr[i];
    const long long z = a / plane, rem = a % plane;
    const int y = (int)(rem / px), x = (int)(rem % px);
    if (z < mz_lo || z >= mz_hi || y < h || y >= ny - h || x < h || x >= nx - h) return;  // not an interior cell: no product
    if (zg0 + z == zsel || y == ysel || x == xsel) return;                                // filt added it after scaling
    sum[a] = golden ? saved[i] + cube[a] * other[a] : __builtin_fmaf(cube[a], other[a], saved[i]);
}

extern "C" int fdtdx_filt_planes(fdtdx_ctx *ctx, const char *cube, int64_t zsel, int64_t ysel, int64_t xsel, double scale, uint32_t order,
                                 uint64_t z_lo, uint64_t z_hi, void *stream)
{
    float *pc;
    int rc = fdtdx_find_cube(ctx, cube, &pc);
    if (rc) return rc;
    if ((rc = owned_range(ctx, z_lo, z_hi, "filt"))) return rc;
    if (order != FDTDX_ORDER_CHAIN && order != FDTDX_ORDER_GOLDEN) return fail(FDTDX_ERR_INVALID, "unknown order %u", order);
    if (z_lo == z_hi) return FDTDX_OK;
    return launch_filt(ctx, pc, z_lo, z_hi, zsel, ysel, xsel, scale, order == FDTDX_ORDER_GOLDEN ? 1 : 0, nullptr, nullptr, 0, 0, 0, (hipStream_t)stream);
}

// One whole step of main.cu:160-163 on the planes this rank owns, with the product-accumulate folded
// into the update kernel (24 instead of 32 bytes per cell); results are those of the four separate calls.
// `saved` receives one float per listed address (the sum values before the step); the two entry points
// below differ only in where that buffer lives.
struct CompositeStepArgs {
    float *po, *pi, *pv, *ps;
    const uint64_t *addr;
    uint64_t n;
};

static int composite_step_check(fdtdx_ctx *ctx, const char *out, const char *in, const char *vel, const char *sum, const char *addr_name,
                                const fdtdx_composite *comp, CompositeStepArgs *a)
{
    if (!ctx || !comp || !addr_name) return fail(FDTDX_ERR_INVALID, "null argument");
    int rc;
    if ((rc = fdtdx_find_cube(ctx, out, &a->po)) || (rc = fdtdx_find_cube(ctx, in, &a->pi)) || (rc = fdtdx_find_cube(ctx, vel, &a->pv)) ||
        (rc = fdtdx_find_cube(ctx, sum, &a->ps)))
        return rc;
    if (a->po == a->pv || a->ps == a->po || a->ps == a->pi || a->ps == a->pv || a->po == a->pi)
        return fail(FDTDX_ERR_INVALID, "composite step needs four different cubes");
    if (comp->scal_den == 0.0) return fail(FDTDX_ERR_INVALID, "scal_den is zero");
    auto it = ctx->addr.find(addr_name);
    if (it == ctx->addr.end()) return fail(FDTDX_ERR_NOTFOUND, "no address list '%s'", addr_name);
    a->addr = it->second.ptr;
    a->n = it->second.count;
    return FDTDX_OK;
}

static int composite_step_run(fdtdx_ctx *ctx, const CompositeStepArgs &a, const char *out, const char *addr_name, const fdtdx_composite *comp,
                              float add, int64_t zsel, int64_t ysel, int64_t xsel, double scale, float *saved, void *stream)
{
    const fdtdx_geometry &g = ctx->g;
    const uint64_t r = comp->radius, n = a.n;
    const uint64_t lo = g.ghost_lo ? g.ghost_lo : r, hi = g.ghost_hi ? g.local_planes - g.ghost_hi : g.local_planes - r;
    const uint64_t olo = g.ghost_lo, ohi = g.ghost_lo + g.owned_planes;
    const int golden = comp->order == FDTDX_ORDER_GOLDEN ? 1 : 0;
    hipStream_t s = (hipStream_t)stream;
    int rc;
    if (n) {  // every word of `saved` the step reads is written here first
        hipLaunchKernelGGL(gather_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, saved, (const float *)a.ps, a.addr, n);
        HIP_TRY(hipGetLastError());
    }
    const int64_t sel[3] = {zsel, ysel, xsel};
    if ((rc = launch_sweep(ctx, a.po, a.pi, nullptr, lo, hi, stream, comp, a.pv, a.ps, sel))) return rc;
    if ((rc = fdtdx_inject(ctx, out, addr_name, add, olo, ohi, stream))) return rc;
    if (ohi > olo && (rc = launch_filt(ctx, a.po, olo, ohi, zsel, ysel, xsel, scale, golden, a.ps, a.pi, (int)r, lo, hi, s))) return rc;
    if (n) {
        hipLaunchKernelGGL(redo_product_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, a.ps, (const float *)saved,
                           (const float *)a.po, (const float *)a.pi, a.addr, n, (long long)g.plane, (int)g.pitch_x, (int)ctx->cfg.X,
                           (int)ctx->cfg.Y, (int)r, (long long)lo, (long long)hi, (long long)g.first_global, (long long)zsel, (long long)ysel,
                           (long long)xsel, golden);
        HIP_TRY(hipGetLastError());
    }
    return FDTDX_OK;
}

extern "C" int fdtdx_composite_step(fdtdx_ctx *ctx, const char *out, const char *in, const char *vel, const char *sum, const char *addr_name,
                                    const fdtdx_composite *comp, float add, int64_t zsel, int64_t ysel, int64_t xsel, double scale,
                                    void *stream)
{
    CompositeStepArgs a;
    int rc = composite_step_check(ctx, out, in, vel, sum, addr_name, comp, &a);
    if (rc) return rc;
    USE_DEVICE(ctx);
    if (a.n > ctx->saved_cap) {
        if (ctx->saved_sum) HIP_TRY(hipFree(ctx->saved_sum));
        ctx->saved_sum = nullptr, ctx->saved_cap = 0;
        HIP_TRY(hipMalloc((void **)&ctx->saved_sum, a.n * sizeof(float)));
        ctx->saved_cap = a.n;
    }
    return composite_step_run(ctx, a, out, addr_name, comp, add, zsel, ysel, xsel, scale, ctx->saved_sum, stream);
}

// The same step for callers that cannot allocate (HIP graph capture, serving loops, harnesses that forbid hipMalloc):
// the per-address buffer is the file-scope `scratch` array instead of context memory.  No hipMalloc, hipFree or other
// allocating routine is called on this path.  All calls on one GPU share that GPU's array, so calls that overlap
// on one GPU (two streams, two contexts) race; every word a call reads from the array it has written itself
// earlier in the same call (the gather), so nothing has to be cleared between calls.
extern "C" int fdtdx_composite_step_static(fdtdx_ctx *ctx, const char *out, const char *in, const char *vel, const char *sum,
                                           const char *addr_name, const fdtdx_composite *comp, float add, int64_t zsel, int64_t ysel,
                                           int64_t xsel, double scale, void *stream)
{
    CompositeStepArgs a;
    int rc = composite_step_check(ctx, out, in, vel, sum, addr_name, comp, &a);
    if (rc) return rc;
    if (a.n > FDTDX_STATIC_ADDR_MAX)
        return fail(FDTDX_ERR_UNSUPPORTED, "address list '%s' holds %llu addresses on this rank, the static scratch holds %llu: use fdtdx_composite_step",
                    addr_name, (unsigned long long)a.n, (unsigned long long)FDTDX_STATIC_ADDR_MAX);
    USE_DEVICE(ctx);
    if (!ctx->static_scratch && (rc = scratch_address(ctx->device, &ctx->static_scratch))) return rc;
    return composite_step_run(ctx, a, out, addr_name, comp, add, zsel, ysel, xsel, scale, ctx->static_scratch, stream);
}

// sum += a*b on the interior: one quad per lane; grid = (quad blocks, interior rows, planes)
__global__ void mul_acc_kernel(float *sum, const float *a, const float *b, long long plane, int px, int nx, int h, long long z_lo, int golden)
{
    const long long z = z_lo + blockIdx.z;
    const int y = h + (int)blockIdx.y;
    const int q = blockIdx.x * blockDim.x + threadIdx.x;
    const int x0 = 4 * q;
    if (x0 >= nx - h || x0 + 3 < h) return;
    const long long off = z * plane + (long long)y * px + x0;
    const float4 s = *reinterpret_cast<const float4 *>(sum + off);
    const float4 va = *reinterpret_cast<const float4 *>(a + off), vb = *reinterpret_cast<const float4 *>(b + off);
    float sv[4] = {s.x, s.y, s.z, s.w};
    const float av[4] = {va.x, va.y, va.z, va.w}, bv[4] = {vb.x, vb.y, vb.z, vb.w};
#pragma unroll
    for (int i = 0; i < 4; i++)
        if (x0 + i >= h && x0 + i < nx - h) sv[i] = golden ? sv[i] + av[i] * bv[i] : __builtin_fmaf(av[i], bv[i], sv[i]);
    *reinterpret_cast<float4 *>(sum + off) = make_float4(sv[0], sv[1], sv[2], sv[3]);
}

extern "C" int fdtdx_mul_acc(fdtdx_ctx *ctx, const char *sum, const char *a, const char *b, uint32_t radius, uint32_t order, uint64_t z_lo,
                             uint64_t z_hi, void *stream)
{
    float *ps, *pa, *pb;
    int rc;
    if ((rc = fdtdx_find_cube(ctx, sum, &ps)) || (rc = fdtdx_find_cube(ctx, a, &pa)) || (rc = fdtdx_find_cube(ctx, b, &pb))) return rc;
    if ((rc = owned_range(ctx, z_lo, z_hi, "mul_acc"))) return rc;
    if (order != FDTDX_ORDER_CHAIN && order != FDTDX_ORDER_GOLDEN) return fail(FDTDX_ERR_INVALID, "unknown order %u", order);
    const uint64_t X = ctx->cfg.X, Y = ctx->cfg.Y;
    if (radius < 1 || 2 * (uint64_t)radius >= Y || 2 * (uint64_t)radius >= X) return fail(FDTDX_ERR_INVALID, "radius %u leaves no interior", radius);
    if (Y - 2 * radius > 65535) return fail(FDTDX_ERR_UNSUPPORTED, "more than 65535 interior rows");
    USE_DEVICE(ctx);
    const fdtdx_geometry &g = ctx->g;
    const int nq = (int)((X + 3) / 4);
    for (uint64_t z = z_lo; z < z_hi; z += 65535) {  // planes go to the grid's z dimension
        const uint64_t n = std::min<uint64_t>(65535, z_hi - z);
        hipLaunchKernelGGL(mul_acc_kernel, dim3((unsigned)((nq + 127) / 128), (unsigned)(Y - 2 * radius), (unsigned)n), dim3(128), 0,
                           (hipStream_t)stream, ps, pa, pb, (long long)g.plane, (int)g.pitch_x, (int)X, (int)radius, (long long)z,
                           order == FDTDX_ORDER_GOLDEN ? 1 : 0);
        HIP_TRY(hipGetLastError());
    }
    return FDTDX_OK;
}

// ---- extension: staggered-grid E/H updates (no counterpart in the reference; yee_kernel.h) ------------
extern "C" int fdtdx_yee3d_half(fdtdx_ctx *ctx, const char *const fields[6], int which, float coef, uint32_t order, uint64_t z_lo,
                                uint64_t z_hi, void *stream)
{
    if (!ctx || !fields) return fail(FDTDX_ERR_INVALID, "null argument");
    if (which != 0 && which != 1) return fail(FDTDX_ERR_INVALID, "which must be 0 (H half step) or 1 (E half step)");
    if (order != FDTDX_ORDER_CHAIN && order != FDTDX_ORDER_GOLDEN) return fail(FDTDX_ERR_INVALID, "unknown order %u", order);
    const fdtdx_geometry &g = ctx->g;
    if (z_lo < 1 || z_hi > g.local_planes - 1 || z_lo > z_hi) return fail(FDTDX_ERR_INVALID, "plane range [%llu,%llu) needs one plane of support on each side", (unsigned long long)z_lo, (unsigned long long)z_hi);
    YeeParams p;
    int rc;
    for (int i = 0; i < 6; i++) {
        if ((rc = fdtdx_find_cube(ctx, fields[i], &p.f[i]))) return rc;
        for (int k = 0; k < i; k++)
            if (p.f[k] == p.f[i]) return fail(FDTDX_ERR_INVALID, "the six fields must be six different cubes");
    }
    if (z_lo == z_hi) return FDTDX_OK;
    USE_DEVICE(ctx);
    p.plane = (long long)g.plane, p.px = (int)g.pitch_x, p.nx = (int)ctx->cfg.X, p.ny = (int)ctx->cfg.Y;
    p.z_lo = (int)z_lo, p.z_hi = (int)z_hi, p.coef = coef;
    const unsigned gx = (unsigned)(((ctx->cfg.X + 3) / 4 + 63) / 64), gy = (unsigned)((ctx->cfg.Y - 2 + 3) / 4);
    const long long nz = (long long)(z_hi - z_lo);
    // a few thousand workgroups of 4 waves keep the chip busy; z chunks of at least 16 planes keep the register reuse
    long long chunks = std::max<long long>(1, std::min<long long>(nz / 16, 4096 / std::max<long long>(1, (long long)gx * gy)));
    p.zchunk = (int)((nz + chunks - 1) / chunks);
    const unsigned gz = (unsigned)((nz + p.zchunk - 1) / p.zchunk);
    if (gy > 65535 || gz > 65535) return fail(FDTDX_ERR_UNSUPPORTED, "grid too large");
    const dim3 grid(gx, gy, gz), block(256);
    hipStream_t s = (hipStream_t)stream;
    const bool gold = order == FDTDX_ORDER_GOLDEN;
    if (which == 0 && !gold) hipLaunchKernelGGL((yee3d_half_kernel<0, 0>), grid, block, 0, s, p);
    if (which == 0 && gold) hipLaunchKernelGGL((yee3d_half_kernel<0, 1>), grid, block, 0, s, p);
    if (which == 1 && !gold) hipLaunchKernelGGL((yee3d_half_kernel<1, 0>), grid, block, 0, s, p);
    if (which == 1 && gold) hipLaunchKernelGGL((yee3d_half_kernel<1, 1>), grid, block, 0, s, p);
    HIP_TRY(hipGetLastError());
    return FDTDX_OK;
}

extern "C" int fdtdx_yee3d_step(fdtdx_ctx *ctx, const char *const fields[6], float ce, float ch, uint32_t order, void *stream)
{
    if (!ctx) return fail(FDTDX_ERR_INVALID, "null context");
    if (ctx->cfg.nprocs != 1) return fail(FDTDX_ERR_UNSUPPORTED, "fdtdx_yee3d_step runs on one rank; use fdtdx_yee3d_half with an exchange of H after the first half");
    const uint64_t hi = ctx->g.local_planes - 1;
    int rc = fdtdx_yee3d_half(ctx, fields, 0, ch, order, 1, hi, stream);
    if (rc) return rc;
    return fdtdx_yee3d_half(ctx, fields, 1, ce, order, 1, hi, stream);
}

extern "C" int fdtdx_yee3d_step_fused(fdtdx_ctx *ctx, const char *const before[6], const char *const after[6], float ce, float ch,
                                      uint32_t order, void *stream)
{
    if (!ctx || !before || !after) return fail(FDTDX_ERR_INVALID, "null argument");
    if (ctx->cfg.nprocs != 1) return fail(FDTDX_ERR_UNSUPPORTED, "fdtdx_yee3d_step_fused runs on one rank");
    if (order != FDTDX_ORDER_CHAIN && order != FDTDX_ORDER_GOLDEN) return fail(FDTDX_ERR_INVALID, "unknown order %u", order);
    const fdtdx_geometry &g = ctx->g;
    YeeFusedParams p;
    float *f[12];
    int rc;
    for (int i = 0; i < 12; i++) {
        if ((rc = fdtdx_find_cube(ctx, i < 6 ? before[i] : after[i - 6], &f[i]))) return rc;
        for (int k = 0; k < i; k++)
            if (f[k] == f[i]) return fail(FDTDX_ERR_INVALID, "the fused step needs twelve different cubes (six before, six after)");
    }
    USE_DEVICE(ctx);
    for (int i = 0; i < 3; i++) p.eo[i] = f[i], p.ho[i] = f[3 + i], p.en[i] = f[6 + i], p.hn[i] = f[9 + i];
    p.plane = (long long)g.plane, p.px = (int)g.pitch_x, p.nx = (int)ctx->cfg.X, p.ny = (int)ctx->cfg.Y, p.nz = (int)g.local_planes;
    p.z_lo = 1, p.z_hi = (int)g.local_planes - 1, p.ce = ce, p.ch = ch;
#ifndef YEE_LY
#define YEE_LY 7
#endif
    constexpr int LY = YEE_LY;
    const unsigned gx = (unsigned)(((ctx->cfg.X + 3) / 4 + 63) / 64), gy = (unsigned)((ctx->cfg.Y - 2 + LY - 1) / LY);
    const long long nz = p.z_hi - p.z_lo;
    // four workgroups of 8 waves fit a CU (64 VGPRs, 32 KB LDS): z chunks that make the launch one full round of the
    // 1024 slots ran fastest (7 chunks at 512^3: 92 Gcell/s against 84-90 for other counts); every chunk recomputes
    // the H plane below it, so chunks stay at least 8 planes long
    long long chunks = std::max<long long>(1, std::min<long long>(nz / 8, 1024 / std::max<long long>(1, (long long)gx * gy)));
    if (ctx->opt_zchunks > 0) chunks = std::min<long long>(ctx->opt_zchunks, nz);
    p.zchunk = (int)((nz + chunks - 1) / chunks);
    const unsigned gz = (unsigned)((nz + p.zchunk - 1) / p.zchunk);
    if ((long long)gx * gy * gz > 0x7fffffffLL) return fail(FDTDX_ERR_UNSUPPORTED, "grid too large");
    p.gx = (int)gx, p.gy = (int)gy, p.gz = (int)gz;
    const dim3 grid(gx * gy * gz), block(64 * (LY + 1));
    if (order == FDTDX_ORDER_GOLDEN)
        hipLaunchKernelGGL((yee3d_fused_kernel<LY, 1>), grid, block, 0, (hipStream_t)stream, p);
    else
        hipLaunchKernelGGL((yee3d_fused_kernel<LY, 0>), grid, block, 0, (hipStream_t)stream, p);
    HIP_TRY(hipGetLastError());
    return FDTDX_OK;
}

extern "C" int fdtdx_yee2d_steps(int device, float *ex, float *ey, float *hz, uint64_t nx, uint64_t ny, int t0, int steps, float ce, float ch,
                                 uint32_t order, void *stream)
{
    if (!ex || !ey || !hz) return fail(FDTDX_ERR_INVALID, "null argument");
    if (nx < 2 || ny < 2 || nx > 65535 || ny > 0x7fffffffu) return fail(FDTDX_ERR_INVALID, "2-D grid %llux%llu out of range", (unsigned long long)nx, (unsigned long long)ny);
    if (order != FDTDX_ORDER_CHAIN && order != FDTDX_ORDER_GOLDEN) return fail(FDTDX_ERR_INVALID, "unknown order %u", order);
    if (device >= 0) HIP_TRY(hipSetDevice(device));
    const dim3 grid((unsigned)((ny + 255) / 256), (unsigned)nx), block(256);
    hipStream_t s = (hipStream_t)stream;
    for (int t = t0; t < t0 + steps; t++) {
        if (order == FDTDX_ORDER_GOLDEN) {
            hipLaunchKernelGGL((yee2d_e_kernel<1>), grid, block, 0, s, ex, ey, (const float *)hz, (int)nx, (int)ny, (float)t, ce);
            hipLaunchKernelGGL((yee2d_h_kernel<1>), grid, block, 0, s, (const float *)ex, (const float *)ey, hz, (int)nx, (int)ny, ch);
        } else {
            hipLaunchKernelGGL((yee2d_e_kernel<0>), grid, block, 0, s, ex, ey, (const float *)hz, (int)nx, (int)ny, (float)t, ce);
            hipLaunchKernelGGL((yee2d_h_kernel<0>), grid, block, 0, s, (const float *)ex, (const float *)ey, hz, (int)nx, (int)ny, ch);
        }
    }
    HIP_TRY(hipGetLastError());
    return FDTDX_OK;
}

// ---- halo regions (stencil.h:639-647) ---------------------------------------------------------------
extern "C" int fdtdx_halo_region(fdtdx_ctx *ctx, const char *name, int side, int what, float **ptr, size_t *count)
{
    float *d;
    int rc = fdtdx_find_cube(ctx, name, &d);
    if (rc) return rc;
    if (!ptr || !count || (side != 0 && side != 1) || (what != 0 && what != 1)) return fail(FDTDX_ERR_INVALID, "bad halo_region argument");
    const fdtdx_geometry &g = ctx->g;
    const uint64_t M = ctx->cfg.M;
    const bool has = side == 0 ? g.ghost_lo > 0 : g.ghost_hi > 0;
    if (!has) {
        *ptr = nullptr;
        *count = 0;
        return FDTDX_OK;
    }
    uint64_t first;
    if (side == 0)
        first = (what == 0) ? M : 0;  // send the first M owned planes, receive into the M ghost planes below
    else
        first = (what == 0) ? g.local_planes - 2 * M : g.local_planes - M;
    *ptr = d + first * g.plane;
    *count = (size_t)(M * g.plane);
    return FDTDX_OK;
}

extern "C" int fdtdx_set_option(fdtdx_ctx *ctx, const char *key, int64_t value)
{
    if (!ctx || !key) return fail(FDTDX_ERR_INVALID, "null argument");
    if (!strcmp(key, "zchunks"))
        ctx->opt_zchunks = value;
    else if (!strcmp(key, "variant"))
        ctx->opt_variant = value;
    else if (!strcmp(key, "temporal"))
        ctx->opt_temporal = value;
    else if (!strcmp(key, "tb_variant"))
        ctx->opt_tb_variant = value;
    else if (!strcmp(key, "tb_zchunks"))
        ctx->opt_tb_zchunks = value;
    else if (!strcmp(key, "boundary_stream"))
        ctx->opt_boundary_stream = value != 0;
    else if (!strcmp(key, "exchange_priority"))
        ctx->opt_exchange_priority = value != 0;
    else if (!strcmp(key, "fuse_wide"))
        ctx->opt_fuse_wide = value != 0;
    else if (!strcmp(key, "no_strip"))
        ctx->opt_no_strip = value != 0;
    else if (!strcmp(key, "reserve_cus"))
        ctx->opt_reserve_cus = std::max<int64_t>(0, std::min<int64_t>(value, 255));
    else
        return fail(FDTDX_ERR_INVALID, "unknown option '%s'", key);
    return FDTDX_OK;
}

extern "C" int fdtdx_last_launch_info(const fdtdx_ctx *ctx, char *buf, size_t len)
{
    if (!ctx || !buf || !len) return fail(FDTDX_ERR_INVALID, "null argument");
    snprintf(buf, len, "%s%s%s launches=%llu", ctx->last_launch, ctx->last_fused[0] ? " | " : "", ctx->last_fused, (unsigned long long)ctx->launches);
    return FDTDX_OK;
}
