// fdtdx_ctx.h -- what the translation units behind the C ABI (fdtdx_abi.hip, fdtdx_comm.hip) share: the context
// object, error reporting, device selection.  Private to the library.
#pragma once
#include <hip/hip_runtime.h>

#include <map>
#include <string>

#include "../../include/fdtdx.h"

// ---- error reporting: every entry point returns a code and leaves a message for fdtdx_last_error() ----
int fdtdx_fail(int code, const char *fmt, ...) __attribute__((format(printf, 2, 3)));
#define fail fdtdx_fail

#define HIP_TRY(expr)                                                                         \
    do {                                                                                      \
        hipError_t e_ = (expr);                                                               \
        if (e_ != hipSuccess) return fail(FDTDX_ERR_HIP, "%s: %s", #expr, hipGetErrorString(e_)); \
    } while (0)

struct Cube {
    float *ptr = nullptr;
    bool owned = false;
};

struct fdtdx_comm;  // the exchange state of a context (fdtdx_comm.hip)

struct fdtdx_ctx {
    fdtdx_config cfg;
    fdtdx_geometry g;
    int device = 0;
    int num_cus = 256;                 // compute units of the device (hipDeviceAttributeMultiprocessorCount), asked in fdtdx_create
    std::map<std::string, Cube> pool;  // memoryPool of stencil.h:361
    struct AddrList {
        uint64_t *ptr = nullptr;
        size_t capacity = 0, count = 0;
    };
    std::map<std::string, AddrList> addr;  // address lists (stencil.h:363: addrLen)
    int64_t opt_zchunks = 0;           // 0 = auto
    int64_t opt_variant = -1;          // -1 = auto
    int64_t opt_temporal = 1;          // fuse two steps per launch where a fused kernel exists
    int64_t opt_tb_variant = -1;       // shape of the fused kernel, -1 = auto
    int64_t opt_tb_zchunks = 0;
    int64_t opt_boundary_stream = 0;   // 1: a multi-rank step's boundary-plane sweep runs on a stream of its own, beside the interior sweep
    int64_t opt_exchange_priority = 0; // 1: the exchange stream is created at the device's highest stream priority (read when the communicator is set up)
    int64_t opt_fuse_wide = 0;         // 1: time loops may use the two-step kernels of radius 3 and 4 (slower than single steps: kept for measurement)
    int64_t opt_reserve_cus = 0;       // CUs a sweep launch leaves without a workgroup (room for the exchange's kernels)
    float *tb_scratch = nullptr;       // third cube of the fused time loop (owned)
    int64_t opt_no_strip = 0;          // "no_strip": the tail of wide rows always goes to the half-width tile (measurement)
    float *stage = nullptr;            // staging buffer of the padded transfers (owned, allocated on first use)
    size_t stage_floats = 0;
    float *shell_strip[2] = {nullptr, nullptr};  // time loops: the shell components of the quads that straddle the x shell, per cube of the loop (owned)
    const float *strip_cube[2] = {nullptr, nullptr};  // the cubes the strips mirror while a time loop runs
    int strip_radius = 0;                             // radius they were gathered for (0: the strips are not valid)
    float *saved_sum = nullptr;        // fdtdx_composite_step: sum values at the injection addresses
    size_t saved_cap = 0;
    float *static_scratch = nullptr;   // device address of the file-scope `scratch` array on ctx->device
    uint64_t launches = 0;             // sweep launches issued so far (bench bookkeeping)
    char last_launch[256] = "";
    char last_fused[256] = "";         // last fused (two-step) launch, "" if none since creation
    fdtdx_comm *comm = nullptr;        // set by fdtdx_comm_init / fdtdx_comm_init_local
};

// Every entry point that touches the GPU first makes the context's device current, so that several
// contexts in one process (one per GPU, or a thread whose current device is another one) reach the right
// card.  (The reference never selects a device: one rank sees one GPU, run.sh:20.)
#define USE_DEVICE(ctx)                                              \
    do {                                                             \
        if (!(ctx)) return fail(FDTDX_ERR_INVALID, "null context");  \
        HIP_TRY(hipSetDevice((ctx)->device));                        \
    } while (0)

int fdtdx_find_cube(fdtdx_ctx *ctx, const char *name, float **p);
// is there a fused kernel of kt steps per launch for the program on this context's grid (shells: one that takes both shell sources by name)?
// Time loops: gather the strips for cubes a and b (a program of radius r) on `stream` / forget them.  While bound, single-step
// star sweeps into a or b take the shell components of partly stored quads from the strips.
bool fdtdx_strips_bind(fdtdx_ctx *ctx, const float *a, const float *b, int r, void *stream);
void fdtdx_strips_release(fdtdx_ctx *ctx);
bool fdtdx_has_fused(const fdtdx_ctx *ctx, const fdtdx_program *prog, int kt, bool shells);
void fdtdx_comm_release(fdtdx_ctx *ctx);  // called by fdtdx_destroy
// `nplanes` padded planes at device address `dev` into unpadded [nplanes][Y][X] host memory (the device is drained first)
int fdtdx_copy_planes_d2h(fdtdx_ctx *ctx, float *host, const float *dev, uint64_t nplanes);
