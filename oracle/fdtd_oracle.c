/*
 * fdtd_oracle.c -- CPU restatement of the reference's field-update sweeps.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the product path (fdtd_amd/, include/)
 * may call, link or load this file; it is the checker for the HIP kernels and
 * the "port" CPU baseline of bench.py.
 *
 * Parity status: PINNED.  Every *_golden function below is checked bit-for-bit
 * (tests/test_oracle_cpu.py) against the reference's own CPU golden loops
 * compiled from /root/reference by oracle/build_ref.sh with strict IEEE flags
 * (-O2 -ffp-contract=off), and against the committed fixtures in tests/golden/
 * that were produced by those compiled loops.
 *
 * Two summation orders exist for every stencil because the reference itself has
 * two implementations of each update:
 *   ORDER_GOLDEN  the host loop: centre + parenthesised (+r,-r) pairs, plain
 *                 IEEE mul/add          (fdtd1.cu:46-49 ... fdtd8.cu:46-70,
 *                                        fdtdspc.cu:46-52, jacobi.cu:478-485)
 *   ORDER_CHAIN   the device lambda: one flat left-to-right sum, every "+ c*v"
 *                 contracted to an FMA as nvcc's default -fmad=true does
 *                                       (fdtd1.cu:61-64 ... fdtd8.cu:82-93,
 *                                        fdtdspc.cu:64-73, jacobi.cu:31-38)
 *
 * Build: see oracle/Makefile (gcc -O2 -ffp-contract=off -fopenmp -shared).
 */
#include <math.h>
#include <stddef.h>
#include <stdint.h>
#include <string.h>

#define ORDER_GOLDEN 0
#define ORDER_CHAIN 1

#define IDX(z, y, x, ny, nx) (((size_t)(z) * (ny) + (y)) * (nx) + (x))

/* ------------------------------------------------------------------------- */
/* std::mt19937_64 (ISO C++ [rand.predef]): the generator the reference seeds  */
/* per z-plane with 19950918+z (fdtd1.cu:24).                                  */
/* ------------------------------------------------------------------------- */
typedef struct {
    uint64_t s[312];
    int i;
} mt64_t;

static void mt64_seed(mt64_t *g, uint64_t seed)
{
    g->s[0] = seed;
    for (int k = 1; k < 312; k++)
        g->s[k] = 6364136223846793005ULL * (g->s[k - 1] ^ (g->s[k - 1] >> 62)) + (uint64_t)k;
    g->i = 312;
}

static uint64_t mt64_next(mt64_t *g)
{
    if (g->i >= 312) {
        const uint64_t UM = 0xFFFFFFFF80000000ULL, LM = 0x7FFFFFFFULL, A = 0xB5026F5AA96619E9ULL;
        for (int k = 0; k < 312; k++) {
            uint64_t x = (g->s[k] & UM) | (g->s[(k + 1) % 312] & LM);
            g->s[k] = g->s[(k + 156) % 312] ^ (x >> 1) ^ ((x & 1) ? A : 0);
        }
        g->i = 0;
    }
    uint64_t x = g->s[g->i++];
    x ^= (x >> 29) & 0x5555555555555555ULL;
    x ^= (x << 17) & 0x71D67FFFEDA60000ULL;
    x ^= (x << 37) & 0xFFF7EEE000000000ULL;
    x ^= x >> 43;
    return x;
}

/* float(rand())/rand.max(): both operands become float, max() == 2^64-1 rounds to 2^64. */
static inline float mt64_unit(mt64_t *g)
{
    return (float)mt64_next(g) / (float)UINT64_MAX;
}

/*
 * Initial fields of the fdtdN / fdtdspc drivers (fdtd1.cu:21-31): per global plane z a
 * generator seeded 19950918+z draws p0 then p1 for every cell in (y,x) order.
 * Fills planes [z_begin, z_begin+nz_local) of the global grid into local arrays.
 * nfields = 2 (fdtdN) or 3 (main.cu:27-35 draws vel as a third value); f2 may be NULL.
 */
void oracle_init_mt(float *p0, float *p1, float *f2, int nfields, size_t z_begin, size_t nz_local,
                    size_t ny, size_t nx)
{
#pragma omp parallel for
    for (size_t z = 0; z < nz_local; z++) {
        mt64_t g;
        mt64_seed(&g, 19950918ULL + z_begin + z);
        for (size_t y = 0; y < ny; y++)
            for (size_t x = 0; x < nx; x++) {
                size_t a = IDX(z, y, x, ny, nx);
                p0[a] = mt64_unit(&g);
                p1[a] = mt64_unit(&g);
                if (nfields == 3) {
                    float v = mt64_unit(&g);
                    if (f2) f2[a] = v;
                }
            }
    }
}

/* Initial field of jacobi.cu:412-417: i - j + 1.0/(k+1) evaluated in double, stored as float. */
void oracle_init_jacobi(float *a, size_t nz, size_t ny, size_t nx)
{
#pragma omp parallel for
    for (size_t k = 0; k < nz; k++)
        for (size_t j = 0; j < ny; j++)
            for (size_t i = 0; i < nx; i++)
                a[IDX(k, j, i, ny, nx)] = (float)((double)((int64_t)i - (int64_t)j) + 1.0 / (double)(k + 1));
}

/* ------------------------------------------------------------------------- */
/* star-M: coef = {centre, x+, x-, y+, y-, z+, z-}, the same six for every r.  */
/* Interior [M, n-M) per axis; the shell of `out` is not touched.              */
/* The sweep covers planes [z_lo, z_hi) of a local array whose planes are      */
/* addressed 0..nzl-1 (callers pass z_lo>=M, z_hi<=nzl-M).                      */
/* ------------------------------------------------------------------------- */
void oracle_star_step(float *out, const float *in, size_t nzl, size_t ny, size_t nx, int M,
                      const float *coef, int order, size_t z_lo, size_t z_hi)
{
    (void)nzl;
    const float c0 = coef[0], cxp = coef[1], cxm = coef[2], cyp = coef[3], cym = coef[4],
                czp = coef[5], czm = coef[6];
    const size_t sy = nx, sz = ny * nx;
#pragma omp parallel for
    for (size_t z = z_lo; z < z_hi; z++)
        for (size_t y = (size_t)M; y < ny - (size_t)M; y++)
            for (size_t x = (size_t)M; x < nx - (size_t)M; x++) {
                const float *p = in + IDX(z, y, x, ny, nx);
                float acc = c0 * p[0];
                if (order == ORDER_GOLDEN) {
                    for (int r = 1; r <= M; r++) {
                        acc = acc + (cxp * p[r] + cxm * p[-r]);
                        acc = acc + (cyp * p[(size_t)r * sy] + cym * p[-(ptrdiff_t)((size_t)r * sy)]);
                        acc = acc + (czp * p[(size_t)r * sz] + czm * p[-(ptrdiff_t)((size_t)r * sz)]);
                    }
                } else {
                    for (int r = 1; r <= M; r++) {
                        acc = fmaf(cxp, p[r], acc);
                        acc = fmaf(cxm, p[-r], acc);
                        acc = fmaf(cyp, p[(size_t)r * sy], acc);
                        acc = fmaf(cym, p[-(ptrdiff_t)((size_t)r * sy)], acc);
                        acc = fmaf(czp, p[(size_t)r * sz], acc);
                        acc = fmaf(czm, p[-(ptrdiff_t)((size_t)r * sz)], acc);
                    }
                }
                out[IDX(z, y, x, ny, nx)] = acc;
            }
}

/* ------------------------------------------------------------------------- */
/* spc-13 (fdtdspc.cu:46-52 host, :66-72 device).  coef[13] in lambda order:   */
/*  0 C | 1 x+1 | 2 x-1 | 3 y+1 | 4 y-1 | 5 z+1 | 6 z-1 | 7 (y+1,x+1) |         */
/*  8 (y-1,x-1) | 9 (z+1,y+1) | 10 (z-1,y-1) | 11 (z+1,y+1,x+1) | 12 (z-1,y-1,x-1) */
/* ------------------------------------------------------------------------- */
void oracle_spc13_step(float *out, const float *in, size_t nzl, size_t ny, size_t nx,
                       const float *c, int order, size_t z_lo, size_t z_hi)
{
    (void)nzl;
    const ptrdiff_t sy = (ptrdiff_t)nx, sz = (ptrdiff_t)(ny * nx);
#pragma omp parallel for
    for (size_t z = z_lo; z < z_hi; z++)
        for (size_t y = 1; y < ny - 1; y++)
            for (size_t x = 1; x < nx - 1; x++) {
                const float *p = in + IDX(z, y, x, ny, nx);
                float acc;
                if (order == ORDER_GOLDEN) {
                    acc = c[0] * p[0];
                    acc = acc + (c[1] * p[1] + c[2] * p[-1]);
                    acc = acc + (c[3] * p[sy] + c[4] * p[-sy]);
                    acc = acc + (c[5] * p[sz] + c[6] * p[-sz]);
                    acc = acc + (c[7] * p[sy + 1] + c[8] * p[-sy - 1]);
                    acc = acc + (c[9] * p[sz + sy] + c[10] * p[-sz - sy]);
                    acc = acc + (c[11] * p[sz + sy + 1] + c[12] * p[-sz - sy - 1]);
                } else {
                    acc = c[0] * p[0];
                    acc = fmaf(c[1], p[1], acc);
                    acc = fmaf(c[2], p[-1], acc);
                    acc = fmaf(c[3], p[sy], acc);
                    acc = fmaf(c[4], p[-sy], acc);
                    acc = fmaf(c[5], p[sz], acc);
                    acc = fmaf(c[6], p[-sz], acc);
                    acc = fmaf(c[7], p[sy + 1], acc);
                    acc = fmaf(c[8], p[-sy - 1], acc);
                    acc = fmaf(c[9], p[sz + sy], acc);
                    acc = fmaf(c[10], p[-sz - sy], acc);
                    acc = fmaf(c[11], p[sz + sy + 1], acc);
                    acc = fmaf(c[12], p[-sz - sy - 1], acc);
                }
                out[IDX(z, y, x, ny, nx)] = acc;
            }
}

/* ------------------------------------------------------------------------- */
/* jacobi-7 (jacobi.cu:31-38 device, :478-485 host check loop).                */
/* coef = {cc, ce(x+1), cw(x-1), cs(y+1), cn(y-1), ct(z+1), cb(z-1)};          */
/* evaluation order ce*E + cw*W + cs*S + cn*N + ct*T + cb*B + cc*C.            */
/* Boundary cells are never written (both arrays start identical).            */
/* ------------------------------------------------------------------------- */
void oracle_jacobi7_step(float *out, const float *in, size_t nzl, size_t ny, size_t nx,
                         const float *coef, int order, size_t z_lo, size_t z_hi)
{
    (void)nzl;
    const float cc = coef[0], ce = coef[1], cw = coef[2], cs = coef[3], cn = coef[4], ct = coef[5],
                cb = coef[6];
    const ptrdiff_t sy = (ptrdiff_t)nx, sz = (ptrdiff_t)(ny * nx);
#pragma omp parallel for
    for (size_t z = z_lo; z < z_hi; z++)
        for (size_t y = 1; y < ny - 1; y++)
            for (size_t x = 1; x < nx - 1; x++) {
                const float *p = in + IDX(z, y, x, ny, nx);
                float acc;
                if (order == ORDER_GOLDEN) {
                    acc = ce * p[1] + cw * p[-1] + cs * p[sy] + cn * p[-sy] + ct * p[sz] + cb * p[-sz] +
                          cc * p[0];
                } else {
                    acc = ce * p[1];
                    acc = fmaf(cw, p[-1], acc);
                    acc = fmaf(cs, p[sy], acc);
                    acc = fmaf(cn, p[-sy], acc);
                    acc = fmaf(ct, p[sz], acc);
                    acc = fmaf(cb, p[-sz], acc);
                    acc = fmaf(cc, p[0], acc);
                }
                out[IDX(z, y, x, ny, nx)] = acc;
            }
}

/* kind: 0 star (M, coef[7]) | 1 spc13 (coef[13]) | 2 jacobi7 (coef[7]).       */
/* Runs `steps` ping-pong steps exactly as the drivers do (fdtd1.cu:40-52):    */
/* step t writes a = f(b) then swaps the roles.  On return the arrays hold     */
/* what the reference holds: after an odd count `a` is the newest field.       */
void oracle_run(int kind, float *a, float *b, size_t nz, size_t ny, size_t nx, int M, const float *coef,
                int order, int steps)
{
    float *out = a, *in = b;
    for (int t = 0; t < steps; t++) {
        if (kind == 0)
            oracle_star_step(out, in, nz, ny, nx, M, coef, order, (size_t)M, nz - (size_t)M);
        else if (kind == 1)
            oracle_spc13_step(out, in, nz, ny, nx, coef, order, 1, nz - 1);
        else
            oracle_jacobi7_step(out, in, nz, ny, nx, coef, order, 1, nz - 1);
        float *t2 = out;
        out = in;
        in = t2;
    }
}

/* ------------------------------------------------------------------------- */
/* main.cu composite step: weighted symmetric star with a coordinate term and  */
/* "- old output", then scatter-add, plane scaling and product-accumulate.      */
/* k = {cz, cy, cx, snum, sden, wx, wy, wz} as doubles:                          */
/*   out = cz*z + cy*y + cx*x + vel*snum/sden*(C + wx*SUM_r(X+r + X-r)           */
/*                                + wy*SUM_r(Y..) + wz*SUM_r(Z..)) - out         */
/*   ORDER_GOLDEN  the host loop (main.cu:55-72): the constants are double       */
/*                 literals, so the whole right-hand side is evaluated in double */
/*                 (pairs are added in float first) and rounded once             */
/*   ORDER_CHAIN   the device lambda (main.cu:104-118): float constants, float   */
/*                 arithmetic, every "+ c*v" an FMA; scal = (float)(snum/sden)   */
/*                 as the call site passes it (main.cu:160)                      */
/* z, y, x are GLOBAL coordinates: zg0 = global z of local plane 0.              */
/* ------------------------------------------------------------------------- */
void oracle_composite_sweep(float *out, const float *in, const float *vel, size_t nzl, size_t ny, size_t nx,
                            int M, const double *k, int order, size_t z_lo, size_t z_hi, size_t zg0)
{
    (void)nzl;
    const double cz = k[0], cy = k[1], cx = k[2], snum = k[3], sden = k[4], wx = k[5], wy = k[6], wz = k[7];
    const float czf = (float)cz, cyf = (float)cy, cxf = (float)cx, scal = (float)(snum / sden);
    const float wxf = (float)wx, wyf = (float)wy, wzf = (float)wz;
    const ptrdiff_t sy = (ptrdiff_t)nx, sz = (ptrdiff_t)(ny * nx);
#pragma omp parallel for
    for (size_t z = z_lo; z < z_hi; z++)
        for (size_t y = (size_t)M; y < ny - (size_t)M; y++)
            for (size_t x = (size_t)M; x < nx - (size_t)M; x++) {
                const size_t a = IDX(z, y, x, ny, nx);
                const float *p = in + a;
                const size_t gz = zg0 + z;
                if (order == ORDER_GOLDEN) {
                    double s = (double)p[0];
                    for (int r = 1; r <= M; r++) s = s + wx * (double)(p[r] + p[-r]);
                    for (int r = 1; r <= M; r++) s = s + wy * (double)(p[r * sy] + p[-r * sy]);
                    for (int r = 1; r <= M; r++) s = s + wz * (double)(p[r * sz] + p[-r * sz]);
                    double t = cz * (double)gz + cy * (double)y;
                    t = t + cx * (double)x;
                    t = t + (double)vel[a] * snum / sden * s;
                    out[a] = (float)(t - (double)out[a]);
                } else {
                    float s = p[0];
                    for (int r = 1; r <= M; r++) s = fmaf(wxf, p[r] + p[-r], s);
                    for (int r = 1; r <= M; r++) s = fmaf(wyf, p[r * sy] + p[-r * sy], s);
                    for (int r = 1; r <= M; r++) s = fmaf(wzf, p[r * sz] + p[-r * sz], s);
                    float t = czf * (float)(uint32_t)gz;
                    t = fmaf(cyf, (float)(uint32_t)y, t);
                    t = fmaf(cxf, (float)(uint32_t)x, t);
                    t = fmaf(vel[a] * scal, s, t);
                    out[a] = t - out[a];
                }
            }
}

/* inject (stencil.h:810-815, main.cu:73 / :120-123): cube[addr[i]] += add, addresses are linear */
/* indices into the local unpadded array; repeated addresses add repeatedly.                      */
void oracle_inject(float *cube, const uint64_t *addr, size_t na, float add)
{
    for (size_t i = 0; i < na; i++) cube[addr[i]] += add;
}

/* filt with the functor of main.cu:124-128 (host: :75-79): scale every cell, shell included,   */
/* whose global z == zsel or y == ysel or x == xsel.  Host order multiplies by the double        */
/* literal and rounds; device order multiplies by the float constant.                            */
void oracle_filt(float *cube, size_t nzl, size_t ny, size_t nx, size_t zg0, size_t zsel, size_t ysel, size_t xsel,
                 double scale, int order, size_t z_lo, size_t z_hi)
{
    (void)nzl;
    const float sf = (float)scale;
#pragma omp parallel for
    for (size_t z = z_lo; z < z_hi; z++)
        for (size_t y = 0; y < ny; y++)
            for (size_t x = 0; x < nx; x++)
                if (zg0 + z == zsel || y == ysel || x == xsel) {
                    float *c = cube + IDX(z, y, x, ny, nx);
                    *c = (order == ORDER_GOLDEN) ? (float)((double)*c * scale) : *c * sf;
                }
}

/* mul (main.cu:81-84 / :129-133): sum += a*b on the interior; the device contracts to an FMA. */
void oracle_mul_acc(float *sum, const float *a, const float *b, size_t nzl, size_t ny, size_t nx, int M, int order,
                    size_t z_lo, size_t z_hi)
{
    (void)nzl;
#pragma omp parallel for
    for (size_t z = z_lo; z < z_hi; z++)
        for (size_t y = (size_t)M; y < ny - (size_t)M; y++)
            for (size_t x = (size_t)M; x < nx - (size_t)M; x++) {
                const size_t i = IDX(z, y, x, ny, nx);
                sum[i] = (order == ORDER_GOLDEN) ? sum[i] + a[i] * b[i] : fmaf(a[i], b[i], sum[i]);
            }
}

/* One step of main.cu:54-84 on an undecomposed grid: p0 = composite(p1, vel, p0); inject +1;    */
/* scale the planes z = nz/2, y = ny/3, x = nx/4 by 0.9; sum += p0*p1.                            */
void oracle_main_step(float *p0, const float *p1, const float *vel, float *sum, const uint64_t *addr, size_t na,
                      size_t nz, size_t ny, size_t nx, int order)
{
    const double k[8] = {0.5, -0.2, -0.3, 1.0, 6 * 8 + 1, 1.0, 2.0, 3.0};
    oracle_composite_sweep(p0, p1, vel, nz, ny, nx, 4, k, order, 4, nz - 4, 0);
    oracle_inject(p0, addr, na, 1.0f);
    oracle_filt(p0, nz, ny, nx, 0, nz / 2, ny / 3, nx / 4, 0.9, order, 0, nz);
    oracle_mul_acc(sum, p0, p1, nz, ny, nx, 4, order, 4, nz - 4);
}

/* ------------------------------------------------------------------------- */
/* EXTENSION -- NO REFERENCE ORACLE.  The two kernels below are named by the   */
/* build target but do not exist in the reference; this file is their only     */
/* definition ("parity unpinned": nothing of the reference pins them).         */
/*                                                                             */
/* 2-D TE-mode update on [NX][NY] arrays (j fastest), the classic ex/ey/hz      */
/* scheme: per step t                                                           */
/*   ey[0][j]  = (float)t                                                       */
/*   ey[i][j] -= 0.5f*(hz[i][j] - hz[i-1][j])          i >= 1                   */
/*   ex[i][j] -= 0.5f*(hz[i][j] - hz[i][j-1])          j >= 1                   */
/*   hz[i][j] -= 0.7f*(((ex[i][j+1] - ex[i][j]) + ey[i+1][j]) - ey[i][j])       */
/*                                                     i < NX-1, j < NY-1       */
/* ORDER_GOLDEN: separate multiply and subtract; ORDER_CHAIN: a = fmaf(-c,d,a). */
/* ------------------------------------------------------------------------- */
void oracle_yee2d_init(float *ex, float *ey, float *hz, size_t nx, size_t ny)
{
    for (size_t i = 0; i < nx; i++)
        for (size_t j = 0; j < ny; j++) {
            ex[i * ny + j] = ((float)i * (float)(j + 1)) / (float)nx;
            ey[i * ny + j] = ((float)i * (float)(j + 2)) / (float)ny;
            hz[i * ny + j] = ((float)i * (float)(j + 3)) / (float)nx;
        }
}

void oracle_yee2d_step(float *ex, float *ey, float *hz, size_t nx, size_t ny, int t, float ce, float ch, int order)
{
    for (size_t j = 0; j < ny; j++) ey[j] = (float)t;
#pragma omp parallel for
    for (size_t i = 1; i < nx; i++)
        for (size_t j = 0; j < ny; j++) {
            const float d = hz[i * ny + j] - hz[(i - 1) * ny + j];
            ey[i * ny + j] = (order == ORDER_GOLDEN) ? ey[i * ny + j] - ce * d : fmaf(-ce, d, ey[i * ny + j]);
        }
#pragma omp parallel for
    for (size_t i = 0; i < nx; i++)
        for (size_t j = 1; j < ny; j++) {
            const float d = hz[i * ny + j] - hz[i * ny + j - 1];
            ex[i * ny + j] = (order == ORDER_GOLDEN) ? ex[i * ny + j] - ce * d : fmaf(-ce, d, ex[i * ny + j]);
        }
#pragma omp parallel for
    for (size_t i = 0; i < nx - 1; i++)
        for (size_t j = 0; j < ny - 1; j++) {
            const float d = ((ex[i * ny + j + 1] - ex[i * ny + j]) + ey[(i + 1) * ny + j]) - ey[i * ny + j];
            hz[i * ny + j] = (order == ORDER_GOLDEN) ? hz[i * ny + j] - ch * d : fmaf(-ch, d, hz[i * ny + j]);
        }
}

/* 3-D Yee update, six fields [z][y][x], scalar coefficients, outer shell of every field untouched   */
/* (perfect conductor box).  One step = H half step then E half step on the interior [1, n-1):        */
/*   Hx -= ch*((Ez[y+1] - Ez) - (Ey[z+1] - Ey))      Ex += ce*((Hz - Hz[y-1]) - (Hy - Hy[z-1]))       */
/*   Hy -= ch*((Ex[z+1] - Ex) - (Ez[x+1] - Ez))      Ey += ce*((Hx - Hx[z-1]) - (Hz - Hz[x-1]))       */
/*   Hz -= ch*((Ey[x+1] - Ey) - (Ex[y+1] - Ex))      Ez += ce*((Hy - Hy[x-1]) - (Hx - Hx[y-1]))       */
/* f[0..2] = Ex,Ey,Ez, f[3..5] = Hx,Hy,Hz.  which: 0 = H half step, 1 = E half step, on planes         */
/* [z_lo, z_hi).                                                                                      */
void oracle_yee3d_half(float *const *f, int which, size_t nz, size_t ny, size_t nx, float coef, int order, size_t z_lo,
                       size_t z_hi)
{
    (void)nz;
    const ptrdiff_t sx = 1, sy = (ptrdiff_t)nx, sz = (ptrdiff_t)(ny * nx);
    const float *Ex = f[0], *Ey = f[1], *Ez = f[2], *Hx = f[3], *Hy = f[4], *Hz = f[5];
#pragma omp parallel for
    for (size_t z = z_lo; z < z_hi; z++)
        for (size_t y = 1; y < ny - 1; y++)
            for (size_t x = 1; x < nx - 1; x++) {
                const size_t a = IDX(z, y, x, ny, nx);
                float d[3];
                float *dst[3];
                if (which == 0) {
                    d[0] = (Ez[a + sy] - Ez[a]) - (Ey[a + sz] - Ey[a]);
                    d[1] = (Ex[a + sz] - Ex[a]) - (Ez[a + sx] - Ez[a]);
                    d[2] = (Ey[a + sx] - Ey[a]) - (Ex[a + sy] - Ex[a]);
                    dst[0] = f[3] + a, dst[1] = f[4] + a, dst[2] = f[5] + a;
                } else {
                    d[0] = (Hz[a] - Hz[a - sy]) - (Hy[a] - Hy[a - sz]);
                    d[1] = (Hx[a] - Hx[a - sz]) - (Hz[a] - Hz[a - sx]);
                    d[2] = (Hy[a] - Hy[a - sx]) - (Hx[a] - Hx[a - sy]);
                    dst[0] = f[0] + a, dst[1] = f[1] + a, dst[2] = f[2] + a;
                }
                const float c = (which == 0) ? -coef : coef;
                for (int k = 0; k < 3; k++) *dst[k] = (order == ORDER_GOLDEN) ? *dst[k] + c * d[k] : fmaf(c, d[k], *dst[k]);
            }
}
