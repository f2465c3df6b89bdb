// star_kernel.h -- radius-M star sweep for gfx950 (wave64), one time step per launch.
//
// Replaces prop_inner_kernel / prop_center_kernel / prop_halo_kernel / prop_small_kernel of the
// reference (stencil.h:127-335) and the device lambdas of fdtd1..8.cu (fdtd1.cu:61-64 ...
// fdtd8.cu:82-93) and jacobi.cu:31-38.
//
// Shape of the computation
//   * A workgroup owns a tile of TQ float4 "quads" (4*TQ cells, x unit-stride) by CY = LY*RY rows
//     and marches a chunk of z planes.  Thread (tx, ty) owns quad tx of rows ty, ty+LY, ...
//   * z neighbours live in a per-thread register queue of 2M+1 float4 (planes z-M..z+M); each
//     plane is read from HBM exactly once per tile as one coalesced 16-byte-per-lane load.
//   * x and y neighbours come from the centre plane staged in LDS: every thread publishes its
//     centre quad, and the tile's halo (M rows above/below, ceil(M/4) quads left/right, no
//     corners -- a star never reads them) is fetched by extra predicated loads spread over the
//     workgroup.  The x neighbours of a quad are the aligned quads left/right of it, so shifted
//     operands are picked by register naming, not by unaligned LDS reads or lane shuffles.
//   * LDS is double buffered: one barrier per plane.
//   * The loads of plane z+M+1 (and of the halo of plane z+1) are issued before the arithmetic of
//     plane z, so they fly across the barrier and the FMA chain.
#pragma once
#include "sweep_common.h"

namespace fdtdx {

template <int ORDER>
__device__ __forceinline__ float pair_or_chain(float acc, float cp, float vp, float cm, float vm)
{
    if constexpr (ORDER == ORD_STAR_GOLDEN) {
        return acc + (cp * vp + cm * vm);
    } else {
        acc = __builtin_fmaf(cp, vp, acc);
        return __builtin_fmaf(cm, vm, acc);
    }
}

template <int M, int TQ, int LY, int RY, int ORDER>
__global__ __launch_bounds__(TQ *LY) void star_sweep_kernel(const SweepParams p)
{
    constexpr int HQ = (M + 3) / 4;                // halo quads on each side in x
    constexpr int NT = TQ * LY;                    // threads
    constexpr int CY = LY * RY;                    // computed rows per tile
    constexpr int LROWS = CY + 2 * M;              // LDS rows (tile + y halo)
    constexpr int LPQ = TQ + 2 * HQ;               // LDS row pitch in quads (tile + x halo)
    constexpr int NHY = 2 * M * TQ;                // halo quads above/below
    constexpr int NHQ = NHY + CY * 2 * HQ;         // + halo quads left/right
    constexpr int NHK = (NHQ + NT - 1) / NT;       // halo loads per thread
    constexpr int QN = 2 * M + 1;

    __shared__ float4 lds[2][LROWS * LPQ];

    const int t = threadIdx.x;
    const int tx = t % TQ, ty = t / TQ;
    const int q0 = blockIdx.x * TQ;                // first quad of the tile
    const int y0 = p.h + blockIdx.y * CY;          // first computed row of the tile
    const int zs = p.z_lo + blockIdx.z * p.zchunk;
    const int ze = min(zs + p.zchunk, p.z_hi);
    if (zs >= ze) return;                          // whole workgroup leaves together

    // ---- own quads -------------------------------------------------------------------------
    const int qx = q0 + tx;
    const bool qok = qx < p.pxq;
    int goff[RY];       // in-plane offset of the own quad of row j
    bool rok[RY];       // row exists (may be loaded)
    bool cok[RY];       // row is computed
#pragma unroll
    for (int j = 0; j < RY; j++) {
        const int y = y0 + ty + j * LY;
        rok[j] = qok && (y < p.ny);
        cok[j] = qok && (y < p.ny - p.h);
        goff[j] = y * p.px + 4 * qx;
    }
    // per-cell store mask of the own quad: h <= x < nx-h
    const int x0 = 4 * qx;
    const bool m0 = (x0 + 0 >= p.h) && (x0 + 0 < p.nx - p.h);
    const bool m1 = (x0 + 1 >= p.h) && (x0 + 1 < p.nx - p.h);
    const bool m2 = (x0 + 2 >= p.h) && (x0 + 2 < p.nx - p.h);
    const bool m3 = (x0 + 3 >= p.h) && (x0 + 3 < p.nx - p.h);
    const bool mall = m0 && m1 && m2 && m3;
    const bool many = m0 || m1 || m2 || m3;

    // ---- halo quads assigned to this thread --------------------------------------------------
    int hgoff[NHK], hloff[NHK];
    bool hok[NHK], hin[NHK];
#pragma unroll
    for (int k = 0; k < NHK; k++) {
        const int hidx = t + k * NT;
        int lrow, lcol;
        if (hidx < NHY) {
            const int rr = hidx / TQ, cc = hidx % TQ;
            lrow = (rr < M) ? rr : CY + rr;
            lcol = HQ + cc;
        } else {
            const int h2 = hidx - NHY;
            const int rr = h2 / (2 * HQ), cc = h2 % (2 * HQ);
            lrow = M + rr;
            lcol = (cc < HQ) ? cc : TQ + cc;
        }
        const int gy = y0 - M + lrow;
        const int gq = q0 - HQ + lcol;
        hin[k] = (k < NHK - 1) || (hidx < NHQ);
        hok[k] = hin[k] && (gy < p.ny) && (gq >= 0) && (gq < p.pxq);
        hgoff[k] = gy * p.px + 4 * gq;
        hloff[k] = lrow * LPQ + lcol;
    }

    // ---- prologue: fill the z queue for centre plane zs ---------------------------------------
    const float *pin = p.in + (long long)(zs - M) * p.plane;  // plane zs-M
    float4 q[RY][QN];
    float4 nxt[RY];
    float4 hv[NHK];
    const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int j = 0; j < RY; j++) {
#pragma unroll
        for (int k = 1; k < QN; k++) q[j][k] = rok[j] ? ld4(pin + (long long)(k - 1) * p.plane + goff[j]) : zero4;
        q[j][0] = zero4;
        nxt[j] = rok[j] ? ld4(pin + (long long)(2 * M) * p.plane + goff[j]) : zero4;
    }
    const float *pc = p.in + (long long)zs * p.plane;  // centre plane
#pragma unroll
    for (int k = 0; k < NHK; k++) hv[k] = hok[k] ? ld4(pc + hgoff[k]) : zero4;
    float *po = p.out + (long long)zs * p.plane;
    const float *pn = pc + (long long)(M + 1) * p.plane;  // plane z+M+1 for the next iteration

    const float c0 = p.c[0], cxp = p.c[1], cxm = p.c[2], cyp = p.c[3], cym = p.c[4], czp = p.c[5], czm = p.c[6];

    for (int z = zs; z < ze; z++) {
        float4 *buf = lds[(z - zs) & 1];
        // shift the queue, take in plane z+M, publish the centre plane and its halo
#pragma unroll
        for (int j = 0; j < RY; j++) {
#pragma unroll
            for (int k = 0; k < QN - 1; k++) q[j][k] = q[j][k + 1];
            q[j][QN - 1] = nxt[j];
            buf[(M + ty + j * LY) * LPQ + HQ + tx] = q[j][M];
        }
#pragma unroll
        for (int k = 0; k < NHK; k++)
            if (hin[k]) buf[hloff[k]] = hv[k];
        // prefetch for the next centre plane
        if (z + 1 < ze) {
            pc += p.plane;
#pragma unroll
            for (int j = 0; j < RY; j++) nxt[j] = rok[j] ? ld4(pn + goff[j]) : zero4;
#pragma unroll
            for (int k = 0; k < NHK; k++) hv[k] = hok[k] ? ld4(pc + hgoff[k]) : zero4;
            pn += p.plane;
        }
        __syncthreads();

#pragma unroll
        for (int j = 0; j < RY; j++) {
            if (!(cok[j] && many)) continue;
            const float4 *row = buf + (M + ty + j * LY) * LPQ + HQ + tx;
            // x window: quads [-HQ .. +HQ] around the own quad, as a flat register array
            float xs[4 * (2 * HQ + 1)];
#pragma unroll
            for (int s = -HQ; s <= HQ; s++) {
                const float4 v = (s == 0) ? q[j][M] : row[s];
                xs[4 * (s + HQ) + 0] = v.x;
                xs[4 * (s + HQ) + 1] = v.y;
                xs[4 * (s + HQ) + 2] = v.z;
                xs[4 * (s + HQ) + 3] = v.w;
            }
            constexpr int XB = 4 * HQ;  // index of the own quad's first cell in xs
            float acc[4];
            if constexpr (ORDER == ORD_STAR_CHAIN || ORDER == ORD_STAR_GOLDEN) {
#pragma unroll
                for (int i = 0; i < 4; i++) acc[i] = c0 * xs[XB + i];
#pragma unroll
                for (int r = 1; r <= M; r++) {
                    const float4 yp = row[r * LPQ], ym = row[-r * LPQ];
                    const float4 zp = q[j][M + r], zm = q[j][M - r];
                    const float ypv[4] = {yp.x, yp.y, yp.z, yp.w}, ymv[4] = {ym.x, ym.y, ym.z, ym.w};
                    const float zpv[4] = {zp.x, zp.y, zp.z, zp.w}, zmv[4] = {zm.x, zm.y, zm.z, zm.w};
#pragma unroll
                    for (int i = 0; i < 4; i++) {
                        acc[i] = pair_or_chain<ORDER>(acc[i], cxp, xs[XB + i + r], cxm, xs[XB + i - r]);
                        acc[i] = pair_or_chain<ORDER>(acc[i], cyp, ypv[i], cym, ymv[i]);
                        acc[i] = pair_or_chain<ORDER>(acc[i], czp, zpv[i], czm, zmv[i]);
                    }
                }
            } else {
                // jacobi.cu:31-38 order: ce*E + cw*W + cs*S + cn*N + ct*T + cb*B + cc*C
                // coefficient slots: c0=cc, cxp=ce, cxm=cw, cyp=cs, cym=cn, czp=ct, czm=cb
                static_assert(ORDER == ORD_STAR_CHAIN || ORDER == ORD_STAR_GOLDEN || M == 1, "jacobi is radius 1");
                const float4 yp = row[LPQ], ym = row[-LPQ];
                const float4 zp = q[j][M + 1], zm = q[j][M - 1];
                const float ypv[4] = {yp.x, yp.y, yp.z, yp.w}, ymv[4] = {ym.x, ym.y, ym.z, ym.w};
                const float zpv[4] = {zp.x, zp.y, zp.z, zp.w}, zmv[4] = {zm.x, zm.y, zm.z, zm.w};
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    if constexpr (ORDER == ORD_JACOBI_GOLDEN) {
                        acc[i] = cxp * xs[XB + i + 1] + cxm * xs[XB + i - 1] + cyp * ypv[i] + cym * ymv[i] +
                                 czp * zpv[i] + czm * zmv[i] + c0 * xs[XB + i];
                    } else {
                        float a = cxp * xs[XB + i + 1];
                        a = __builtin_fmaf(cxm, xs[XB + i - 1], a);
                        a = __builtin_fmaf(cyp, ypv[i], a);
                        a = __builtin_fmaf(cym, ymv[i], a);
                        a = __builtin_fmaf(czp, zpv[i], a);
                        a = __builtin_fmaf(czm, zmv[i], a);
                        acc[i] = __builtin_fmaf(c0, xs[XB + i], a);
                    }
                }
            }
            float *o = po + goff[j];
            if (mall) {
                *reinterpret_cast<float4 *>(o) = make_float4(acc[0], acc[1], acc[2], acc[3]);
            } else {
                if (m0) o[0] = acc[0];
                if (m1) o[1] = acc[1];
                if (m2) o[2] = acc[2];
                if (m3) o[3] = acc[3];
            }
        }
        po += p.plane;
    }
}

template <int M, int TQ, int LY, int RY, int ORDER>
void star_launch(const SweepParams &p, dim3 grid, hipStream_t s)
{
    hipLaunchKernelGGL((star_sweep_kernel<M, TQ, LY, RY, ORDER>), grid, dim3(TQ * LY), 0, s, p);
}

template <int M, int TQ, int LY, int RY, int ORDER>
struct StarRegistrar {
    StarRegistrar()
    {
        constexpr int HQ = (M + 3) / 4;
        VariantInfo v;
        v.M = M;
        v.order = ORDER;
        v.tq = TQ;
        v.ly = LY;
        v.ry = RY;
        v.threads = TQ * LY;
        v.lds_bytes = 2 * (LY * RY + 2 * M) * (TQ + 2 * HQ) * 16;
        v.launch = &star_launch<M, TQ, LY, RY, ORDER>;
        register_variant(v);
    }
};

#define FDTDX_STAR_VARIANT(M, TQ, LY, RY, ORDER) \
    static ::fdtdx::StarRegistrar<M, TQ, LY, RY, ORDER> reg_##M##_##TQ##_##LY##_##RY##_##ORDER;

}  // namespace fdtdx
