// spc_kernel.h -- 13-point "special" stencil sweep (fdtdspc.cu:64-73) for gfx950.
//
// Replaces prop_center_bf_kernel (stencil.h:75-93), which issues 13 uncached-pattern global
// loads per cell.  Here a workgroup keeps planes z-1, z, z+1 of its tile (plus a one-cell halo
// INCLUDING corners, because the stencil has in-plane and cross-plane diagonals) in a ring of
// four LDS plane images; each plane is read from HBM once per tile with coalesced 16-byte loads
// and every neighbour quad, diagonal or not, is an LDS read (the single cells left/right of a quad
// come from the neighbouring lanes' registers by DPP).  Four ring slots let the loads of
// plane z+2 land while plane z is being computed with a single barrier per plane.  Global traffic
// uses raw buffer loads/stores with out-of-range offsets as lane masks (no divergent branches in
// the z loop, counted waits), exactly as in star_kernel.h.
#pragma once
#include "sweep_common.h"

namespace fdtdx {

__device__ __forceinline__ float lane_from_prev(float v)  // lane l receives lane l-1 (lane 0: lane 63)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x13C, 0xF, 0xF, true));
}
__device__ __forceinline__ float lane_from_next(float v)  // lane l receives lane l+1 (lane 63: lane 0)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x134, 0xF, 0xF, true));
}

template <int TQ, int LY, int RY, int GOLDEN>
__global__ __launch_bounds__(TQ *LY) void spc13_sweep_kernel(const SweepParams p)
{
    constexpr int NT = TQ * LY;
    constexpr int CY = LY * RY;
    constexpr int LROWS = CY + 2;
    constexpr int LPQ = TQ + 2;
    constexpr int NHY = 2 * LPQ;             // full halo rows (corners included)
    constexpr int NHQ = NHY + 2 * CY;        // + left/right halo quads
    constexpr int NHK = (NHQ + NT - 1) / NT;
    constexpr int SLOT = LROWS * LPQ;

    __shared__ float4 ring[4 * SLOT];

    const int t = threadIdx.x;
    const int tx = t % TQ, ty = t / TQ;
    // lanes whose left / right neighbour quad is not held by the previous / next lane of the wave
    const bool first_in_run = (tx == 0) || ((t & 63) == 0);
    const bool last_in_run = (tx == TQ - 1) || ((t & 63) == 63);
    int bx, by, bz;
    logical_tile(p, bx, by, bz);
    const int q0 = bx * TQ;
    const int y0 = 1 + by * CY;
    int zs, ze, zrange_end;
    chunk_planes(p, bz, zs, ze, zrange_end);
    (void)zrange_end;
    if (zs >= ze) return;

    const int qx = q0 + tx;
    const bool qok = qx < p.pxq;
    const int x0 = 4 * qx;
    const bool m0 = (x0 + 0 >= 1) && (x0 + 0 < p.nx - 1);
    const bool m1 = (x0 + 1 >= 1) && (x0 + 1 < p.nx - 1);
    const bool m2 = (x0 + 2 >= 1) && (x0 + 2 < p.nx - 1);
    const bool m3 = (x0 + 3 >= 1) && (x0 + 3 < p.nx - 1);
    const bool mall = m0 && m1 && m2 && m3;
    const bool many = m0 || m1 || m2 || m3;
    // per-lane byte offsets inside a plane; OOB masks the lane (load returns 0, store is dropped)
    unsigned goff[RY], soff[RY], eoff[RY];
    int ownl[RY];
#pragma unroll
    for (int j = 0; j < RY; j++) {
        const int y = y0 + ty + j * LY;
        const unsigned off = (unsigned)(y * p.px + 4 * qx) * 4u;
        const bool rok = qok && (y < p.ny);
        const bool cok = qok && (y < p.ny - 1) && many;
        goff[j] = rok ? off : OOB;
        soff[j] = cok ? off : OOB;
        eoff[j] = (cok && !mall) ? off : OOB;
        ownl[j] = (1 + ty + j * LY) * LPQ + 1 + tx;
    }
    unsigned hgoff[NHK];
    int hloff[NHK];
#pragma unroll
    for (int k = 0; k < NHK; k++) {
        const int hidx = t + k * NT;
        int lrow, lcol;
        if (hidx < NHY) {
            const int rr = hidx / LPQ;
            lcol = hidx % LPQ;
            lrow = (rr == 0) ? 0 : CY + 1;
        } else {
            const int h2 = hidx - NHY;
            lrow = 1 + (h2 >> 1);
            lcol = (h2 & 1) ? TQ + 1 : 0;
        }
        const int gy = y0 - 1 + lrow;
        const int gq = q0 - 1 + lcol;
        const bool hin = (k < NHK - 1) || (hidx < NHQ);
        const bool hok = hin && (gy < p.ny) && (gq >= 0) && (gq < p.pxq);
        hgoff[k] = hok ? (unsigned)(gy * p.px + 4 * gq) * 4u : OOB;
        hloff[k] = hin ? lrow * LPQ + lcol : -1;
    }

    const unsigned plane_bytes = (unsigned)(p.plane * 4);
    // prologue: planes zs-1 and zs straight into their ring slots, plane zs+1 into registers
    const float *pl = p.in + (long long)(zs - 1) * p.plane;
#pragma unroll
    for (int s = 0; s < 2; s++) {
        float4 *slot = ring + ((zs - 1 + s) & 3) * SLOT;
        const __amdgpu_buffer_rsrc_t r = make_rsrc(pl, plane_bytes);
#pragma unroll
        for (int j = 0; j < RY; j++) slot[ownl[j]] = buf_ld4(r, goff[j]);
#pragma unroll
        for (int k = 0; k < NHK; k++)
            if (hloff[k] >= 0) slot[hloff[k]] = buf_ld4(r, hgoff[k]);
        pl += p.plane;
    }
    float4 nxt[RY], hv[NHK], eo[RY];
    float *po = p.out + (long long)zs * p.plane;
    {
        const __amdgpu_buffer_rsrc_t r = make_rsrc(pl, plane_bytes);
#pragma unroll
        for (int j = 0; j < RY; j++) nxt[j] = buf_ld4(r, goff[j]);
#pragma unroll
        for (int k = 0; k < NHK; k++) hv[k] = buf_ld4(r, hgoff[k]);
        const __amdgpu_buffer_rsrc_t ro = make_rsrc(po, plane_bytes);
#pragma unroll
        for (int j = 0; j < RY; j++) eo[j] = buf_ld4(ro, eoff[j]);
    }

    float c[13];
#pragma unroll
    for (int i = 0; i < 13; i++) c[i] = p.c[i];

    for (int z = zs; z < ze; z++) {
        float4 *sp = ring + ((z + 1) & 3) * SLOT;
#pragma unroll
        for (int j = 0; j < RY; j++) sp[ownl[j]] = nxt[j];
#pragma unroll
        for (int k = 0; k < NHK; k++)
            if (hloff[k] >= 0) sp[hloff[k]] = hv[k];
        // prefetch plane z+2 and the output's shell components of plane z+1; empty descriptors
        // (all lanes out of range) when the chunk ends
        const unsigned nb = (z + 1 < ze) ? plane_bytes : 0u;
        pl += p.plane;
        const __amdgpu_buffer_rsrc_t rn = make_rsrc(pl, nb);
        const __amdgpu_buffer_rsrc_t ron = make_rsrc(po + p.plane, nb);
        const __amdgpu_buffer_rsrc_t ro = make_rsrc(po, plane_bytes);
        float4 eo_cur[RY];
#pragma unroll
        for (int j = 0; j < RY; j++) {
            eo_cur[j] = eo[j];
            nxt[j] = buf_ld4(rn, goff[j]);
        }
#pragma unroll
        for (int k = 0; k < NHK; k++) hv[k] = buf_ld4(rn, hgoff[k]);
#pragma unroll
        for (int j = 0; j < RY; j++) eo[j] = buf_ld4(ron, eoff[j]);
        __syncthreads();

        const float4 *s0 = ring + (z & 3) * SLOT;
        const float4 *sm = ring + ((z - 1) & 3) * SLOT;
#pragma unroll
        for (int j = 0; j < RY; j++) {
            const int o = ownl[j];
            // whole-quad reads, kept whole: split into a 12-byte and a 4-byte read (which the compiler
            // does to feed the lane rotates early) they would be bank conflicts at the 16-byte lane stride
            auto quad = [](const float4 *q) {
                f32x4 v = *reinterpret_cast<const f32x4 *>(q);
                asm("" : "+v"(v));
                return make_float4(v[0], v[1], v[2], v[3]);
            };
            const float4 C = quad(s0 + o);
            const float4 U = quad(s0 + o + LPQ), D = quad(s0 + o - LPQ);
            const float4 ZP = quad(sp + o), ZM = quad(sm + o);
            const float4 ZPU = quad(sp + o + LPQ), ZMD = quad(sm + o - LPQ);
            // The six single cells left/right of those quads belong to the neighbouring lanes'
            // quads: take them from their registers with a DPP wave rotate.  (Read from LDS, 4-byte
            // reads at a 16-byte lane stride are 4-way bank conflicts.)  Only the lanes at the end of
            // a wave or of a tile row have no such neighbour and read LDS, a few lanes at a time.
            float Lw = lane_from_prev(C.w), DLw = lane_from_prev(D.w), ZMDLw = lane_from_prev(ZMD.w);
            float Rx = lane_from_next(C.x), URx = lane_from_next(U.x), ZPURx = lane_from_next(ZPU.x);
            if (first_in_run) {
                Lw = reinterpret_cast<const float *>(s0 + o - 1)[3];
                DLw = reinterpret_cast<const float *>(s0 + o - LPQ - 1)[3];
                ZMDLw = reinterpret_cast<const float *>(sm + o - LPQ - 1)[3];
            }
            if (last_in_run) {
                Rx = reinterpret_cast<const float *>(s0 + o + 1)[0];
                URx = reinterpret_cast<const float *>(s0 + o + LPQ + 1)[0];
                ZPURx = reinterpret_cast<const float *>(sp + o + LPQ + 1)[0];
            }

            const float cv[4] = {C.x, C.y, C.z, C.w};
            const float xp[4] = {C.y, C.z, C.w, Rx}, xm[4] = {Lw, C.x, C.y, C.z};
            const float yp[4] = {U.x, U.y, U.z, U.w}, ym[4] = {D.x, D.y, D.z, D.w};
            const float zp[4] = {ZP.x, ZP.y, ZP.z, ZP.w}, zm[4] = {ZM.x, ZM.y, ZM.z, ZM.w};
            const float ypxp[4] = {U.y, U.z, U.w, URx}, ymxm[4] = {DLw, D.x, D.y, D.z};
            const float zpyp[4] = {ZPU.x, ZPU.y, ZPU.z, ZPU.w}, zmym[4] = {ZMD.x, ZMD.y, ZMD.z, ZMD.w};
            const float zpypxp[4] = {ZPU.y, ZPU.z, ZPU.w, ZPURx}, zmymxm[4] = {ZMDLw, ZMD.x, ZMD.y, ZMD.z};
            float acc[4];
#pragma unroll
            for (int i = 0; i < 4; i++) {
                float a = c[0] * cv[i];
                if constexpr (GOLDEN) {
                    a = a + (c[1] * xp[i] + c[2] * xm[i]);
                    a = a + (c[3] * yp[i] + c[4] * ym[i]);
                    a = a + (c[5] * zp[i] + c[6] * zm[i]);
                    a = a + (c[7] * ypxp[i] + c[8] * ymxm[i]);
                    a = a + (c[9] * zpyp[i] + c[10] * zmym[i]);
                    a = a + (c[11] * zpypxp[i] + c[12] * zmymxm[i]);
                } else {
                    a = __builtin_fmaf(c[1], xp[i], a);
                    a = __builtin_fmaf(c[2], xm[i], a);
                    a = __builtin_fmaf(c[3], yp[i], a);
                    a = __builtin_fmaf(c[4], ym[i], a);
                    a = __builtin_fmaf(c[5], zp[i], a);
                    a = __builtin_fmaf(c[6], zm[i], a);
                    a = __builtin_fmaf(c[7], ypxp[i], a);
                    a = __builtin_fmaf(c[8], ymxm[i], a);
                    a = __builtin_fmaf(c[9], zpyp[i], a);
                    a = __builtin_fmaf(c[10], zmym[i], a);
                    a = __builtin_fmaf(c[11], zpypxp[i], a);
                    a = __builtin_fmaf(c[12], zmymxm[i], a);
                }
                acc[i] = a;
            }
            buf_st4(ro, soff[j], m0 ? acc[0] : eo_cur[j].x, m1 ? acc[1] : eo_cur[j].y, m2 ? acc[2] : eo_cur[j].z,
                    m3 ? acc[3] : eo_cur[j].w);
        }
        po += p.plane;
    }
}

template <int TQ, int LY, int RY, int GOLDEN>
void spc13_launch(const SweepParams &p, dim3 grid, hipStream_t s)
{
    hipLaunchKernelGGL((spc13_sweep_kernel<TQ, LY, RY, GOLDEN>), grid, dim3(TQ * LY), 0, s, p);
}

template <int TQ, int LY, int RY, int GOLDEN>
struct SpcRegistrar {
    SpcRegistrar()
    {
        SpcVariantInfo v;
        v.order = GOLDEN;
        v.tq = TQ;
        v.ly = LY;
        v.ry = RY;
        v.threads = TQ * LY;
        v.lds_bytes = 4 * (LY * RY + 2) * (TQ + 2) * 16;
        v.launch = &spc13_launch<TQ, LY, RY, GOLDEN>;
        register_spc_variant(v);
    }
};

#define FDTDX_SPC_VARIANT(TQ, LY, RY, GOLDEN) \
    static ::fdtdx::SpcRegistrar<TQ, LY, RY, GOLDEN> spcreg_##TQ##_##LY##_##RY##_##GOLDEN;

}  // namespace fdtdx
