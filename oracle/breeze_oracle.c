/*
 * breeze_oracle.c — CPU restatement of Breeze.jl's anelastic hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under breeze.jl_amd/ (the product) may
 * include, link, import or call this file.  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg use it, as the checker.
 *
 * PARITY STATUS: "parity unpinned" for the WENO reconstruction arithmetic.
 * The reference (Julia) cannot run in the build container and the arithmetic of
 * WENO / operators / halo filling / the Fourier-tridiagonal solver lives in
 * Oceananigans.jl (compat 0.110.14, /root/reference/Project.toml:42), which is
 * not vendored.  Those parts restate the published Oceananigans algorithm
 * (WENO-Z, Jiang-Shu smoothness x3, buffer-scheme order reduction at Bounded
 * walls, Centered(order-1) advecting-flux interpolation, Thomas algorithm with
 * the |beta| > 10 eps guard, mean removal).  The Breeze-side arithmetic follows
 * the cited reference lines and is pinned by the reference's own known-answer
 * tests restated in tests/test_oracle_*.py (analytic Poisson, projection ->
 * divergence-free, momentum conservation, reference-column closed forms).
 *
 * Layout: every 3-D array is the *parent* (halo-inclusive) array of an
 * Oceananigans Field, column-major with i fastest:
 *     element (i,j,k) [0-based interior] at  p[(i+Hx) + Sx*((j+Hy) + Sy*(k+Hz))]
 * Sx = Nx+2Hx, Sy = Ny+2Hy; a z-face field has Nz+1 levels (k = 0..Nz).
 * Face index convention: face i is the low-side face of cell i
 * (Julia face index I = i+1;  cf. test/substepper_structural.jl:139-155).
 * Flat directions have N = 1, H = 0 and every operator in that direction
 * vanishes (differences) or is the identity (interpolations).
 *
 * All arithmetic is IEEE double; compile with -ffp-contract=off so the
 * operation order below is what is executed (Julia does not contract).
 */
#include <math.h>
#include <stddef.h>
#include <stdlib.h>
#include <string.h>

#define PERIODIC 0
#define BOUNDED 1
#define FLAT 2
#define SLAB 3   /* periodic stencils; halos are filled by the caller (y-slab decomposition tests) */

typedef struct {
    int Nx, Ny, Nz;
    int Hx, Hy, Hz;
    int tx, ty, tz;       /* topology per direction */
    double dx, dy;        /* uniform horizontal spacings (1 for Flat) */
    const double *dzc;    /* cell thickness,  length Nz+2Hz,   entry k+Hz  (k=-Hz..Nz+Hz-1) */
    const double *dzf;    /* centre spacing at face k, length Nz+1+2Hz, entry k+Hz */
    /* reference columns, length Nz+2Hz, entry k+Hz (halo-inclusive) */
    const double *rho_r;  /* reference density   (reference_states.jl:416-423) */
    const double *p_r;    /* reference pressure  (reference_states.jl:425-428) */
    const double *T_r;    /* reference temperature (:436-438) */
    /* thermodynamic constants (thermodynamics_constants.jl:182-212) */
    double g, Rd, Rv, cpd, cpv, p_st;
} og_grid;

static inline size_t SX(const og_grid *G) { return (size_t)(G->Nx + 2 * G->Hx); }
static inline size_t SY(const og_grid *G) { return (size_t)(G->Ny + 2 * G->Hy); }
#define IDX(G, i, j, k) \
    ((size_t)((i) + (G)->Hx) + SX(G) * ((size_t)((j) + (G)->Hy) + SY(G) * (size_t)((k) + (G)->Hz)))

/* ------------------------------------------------------------------------- */
/* WENO reconstruction (Oceananigans.Advection, recalled).      */
/* ------------------------------------------------------------------------- */

static const double WENO_EPS = 1e-8;

/* The open parity hypothesis (the device twin is BZ_WENO_FT2 of csrc/bz_weno.h): what recent Oceananigans
 * evaluates in its second float type FT2 (Float32 by default) cannot be read here.  0 (default): everything in Float64;
 * 1: the quotients tau / (beta_s + eps) through newton_div(Float32, a, b) — Float32 reciprocal + one Newton step in Float64;
 * 2: beta, tau, alpha and the normalised weights in Float32, candidate polynomials and their weighted sum in Float64.
 * Process-wide switch (the oracle is single-model test infrastructure). */
static int og_weno_ft2 = 0;
void og_set_weno_ft2(int level) { og_weno_ft2 = (level == 1 || level == 2) ? level : 0; }
int og_get_weno_ft2(void) { return og_weno_ft2; }
static inline double newton_div32(double a, double b)
{
    const double inv = (double)(1.0f / (float)b);
    double x = a * inv;
    x = x + (a - x * b) * inv;
    return x;
}
static inline double weno_quot(double tau, double bpe) { return og_weno_ft2 == 1 ? newton_div32(tau, bpe) : tau / bpe; }

/* 5th order: cells (a,b,c,d,e), upwind cell c, value at the face between c and d.
 * Stencil 0 = (c,d,e), 1 = (b,c,d), 2 = (a,b,c); optimal weights 3/10, 3/5, 1/10.
 * Smoothness indicators are 3x Jiang-Shu (Oceananigans coefficient tables). */
static inline double weno5(double a, double b, double c, double d, double e)
{
    double b0 = c * (10.0 * c - 31.0 * d + 11.0 * e) + d * (25.0 * d - 19.0 * e) + e * (4.0 * e);
    double b1 = b * (4.0 * b - 13.0 * c + 5.0 * d) + c * (13.0 * c - 13.0 * d) + d * (4.0 * d);
    double b2 = a * (4.0 * a - 19.0 * b + 11.0 * c) + b * (25.0 * b - 31.0 * c) + c * (10.0 * c);
    double tau = fabs(b0 - b2);
    double p0 = (1.0 / 3.0) * c + (5.0 / 6.0) * d - (1.0 / 6.0) * e;
    double p1 = -(1.0 / 6.0) * b + (5.0 / 6.0) * c + (1.0 / 3.0) * d;
    double p2 = (1.0 / 3.0) * a - (7.0 / 6.0) * b + (11.0 / 6.0) * c;
    if (og_weno_ft2 == 2) {
        const float af = (float)a, bf = (float)b, cf = (float)c, df = (float)d, ef = (float)e;
        const float f0 = cf * (10.0f * cf - 31.0f * df + 11.0f * ef) + df * (25.0f * df - 19.0f * ef) + ef * (4.0f * ef);
        const float f1 = bf * (4.0f * bf - 13.0f * cf + 5.0f * df) + cf * (13.0f * cf - 13.0f * df) + df * (4.0f * df);
        const float f2 = af * (4.0f * af - 19.0f * bf + 11.0f * cf) + bf * (25.0f * bf - 31.0f * cf) + cf * (10.0f * cf);
        const float tf = fabsf(f0 - f2);
        const float q0 = tf / (f0 + 1e-8f), q1 = tf / (f1 + 1e-8f), q2 = tf / (f2 + 1e-8f);
        const float w0 = (3.0f / 10.0f) * (1.0f + q0 * q0), w1 = (3.0f / 5.0f) * (1.0f + q1 * q1), w2 = (1.0f / 10.0f) * (1.0f + q2 * q2);
        const float sum = w0 + w1 + w2;
        return (double)(w0 / sum) * p0 + (double)(w1 / sum) * p1 + (double)(w2 / sum) * p2;
    }
    double r0 = weno_quot(tau, b0 + WENO_EPS);
    double r1 = weno_quot(tau, b1 + WENO_EPS);
    double r2 = weno_quot(tau, b2 + WENO_EPS);
    double a0 = (3.0 / 10.0) * (1.0 + r0 * r0);
    double a1 = (3.0 / 5.0) * (1.0 + r1 * r1);
    double a2 = (1.0 / 10.0) * (1.0 + r2 * r2);
    return (a0 * p0 + a1 * p1 + a2 * p2) / (a0 + a1 + a2);
}

/* 3rd order buffer scheme: cells (a,b,c), upwind cell b, face between b and c. */
static inline double weno3(double a, double b, double c)
{
    double b0 = (c - b) * (c - b);
    double b1 = (b - a) * (b - a);
    double tau = fabs(b0 - b1);
    double p0 = 0.5 * b + 0.5 * c;
    double p1 = -0.5 * a + 1.5 * b;
    if (og_weno_ft2 == 2) {
        const float af = (float)a, bf = (float)b, cf = (float)c;
        const float f0 = (cf - bf) * (cf - bf), f1 = (bf - af) * (bf - af);
        const float tf = fabsf(f0 - f1);
        const float q0 = tf / (f0 + 1e-8f), q1 = tf / (f1 + 1e-8f);
        const float w0 = (2.0f / 3.0f) * (1.0f + q0 * q0), w1 = (1.0f / 3.0f) * (1.0f + q1 * q1);
        const float sum = w0 + w1;
        return (double)(w0 / sum) * p0 + (double)(w1 / sum) * p1;
    }
    double r0 = weno_quot(tau, b0 + WENO_EPS);
    double r1 = weno_quot(tau, b1 + WENO_EPS);
    double a0 = (2.0 / 3.0) * (1.0 + r0 * r0);
    double a1 = (1.0 / 3.0) * (1.0 + r1 * r1);
    return (a0 * p0 + a1 * p1) / (a0 + a1);
}

/* ---- orders 7 and 9 (WENO(order = 9) and its buffer cascade 9 -> 7 -> 5 -> 3 -> 1): tables derived in exact arithmetic by
 * tools/gen_weno_tables.py (they reproduce the order-5 table above and the Balsara & Shu (2000) tables for r = 4, 5 that
 * Oceananigans tabulates; WENO-Z with tau = |b0 + 3 b1 - 3 b2 - b3| (r = 4), |b0 + 2 b1 - 6 b2 + 2 b3 + b4| (r = 5): recalled from
 * Oceananigans.Advection, PARITY UNPINNED).  v[0 .. 2r-2], upwind cell v[r-1], value at the face between v[r-1] and v[r].
 * Same order of operations as weno5 above: nested upper-triangular smoothness sums, three-quotient WENO-Z weights, normalisation. */
#include "weno_tables.h"
#define OG_WENO_GENERIC(R)                                                                                    \
    static inline double weno_r##R(const double *v)                                                          \
    {                                                                                                        \
        double beta[R], p[R], tau = 0.0, num = 0.0, den = 0.0;                                               \
        for (int s = 0; s < R; ++s) {                                                                        \
            const double *w = v + (R - 1 - s);                                                               \
            double b = 0.0, q = 0.0;                                                                         \
            for (int j = 0; j < R; ++j) {                                                                    \
                double in = OGW_B##R[s][j][j] * w[j];                                                        \
                for (int l = j + 1; l < R; ++l) in += OGW_B##R[s][j][l] * w[l];                              \
                b = (j == 0) ? w[j] * in : b + w[j] * in;                                                    \
                q = (j == 0) ? OGW_C##R[s][j] * w[j] : q + OGW_C##R[s][j] * w[j];                            \
            }                                                                                                \
            beta[s] = b; p[s] = q;                                                                           \
            tau = (s == 0) ? OGW_T##R[s] * b : tau + OGW_T##R[s] * b;                                        \
        }                                                                                                    \
        tau = fabs(tau);                                                                                     \
        if (og_weno_ft2 == 2) {                                                                              \
            float af[R], sum = 0.0f;                                                                         \
            double acc = 0.0;                                                                                \
            for (int s = 0; s < R; ++s) {                                                                    \
                const float rr = (float)tau / ((float)beta[s] + 1e-8f);                                      \
                af[s] = (float)OGW_D##R[s] * (1.0f + rr * rr);                                               \
                sum += af[s];                                                                                \
            }                                                                                                \
            for (int s = 0; s < R; ++s) acc += (double)(af[s] / sum) * p[s];                                 \
            return acc;                                                                                      \
        }                                                                                                    \
        for (int s = 0; s < R; ++s) {                                                                        \
            double rr = weno_quot(tau, beta[s] + WENO_EPS);                                                  \
            double a = OGW_D##R[s] * (1.0 + rr * rr);                                                        \
            num = (s == 0) ? a * p[s] : num + a * p[s];                                                      \
            den = (s == 0) ? a : den + a;                                                                    \
        }                                                                                                    \
        return num / den;                                                                                    \
    }
OG_WENO_GENERIC(4)
OG_WENO_GENERIC(5)
OG_WENO_GENERIC(3)      /* cross-check of the generic form against weno5 (tests) */

/* WENO(order = 2 R - 1) selected for the run: R = 3 (order 5, default), 4 or 5.  Process-wide: the oracle is single-model test infrastructure. */
static int og_weno_R = 3;
void og_set_weno_order(int order) { og_weno_R = (order == 9) ? 5 : (order == 7) ? 4 : 3; }
int og_get_weno_order(void) { return 2 * og_weno_R - 1; }

/* Largest buffer (3,2,1) usable at index idx of a Bounded direction with N cells.
 * at_face: interpolation target is face idx; else centre idx (from face data).
 * Oceananigans topologically_conditional_interpolation: buffer B is used when
 *   face:   B <= idx <= N-B      (Julia: B+1 <= I <= N+1-B)
 *   centre: B-1 <= idx <= N-B    (Julia: B   <= I <= N+1-B) */
static inline int buffer_at(int idx, int N, int bounded, int at_face)
{
    if (!bounded) return og_weno_R;
    for (int B = og_weno_R; B >= 2; --B) {
        int lo = at_face ? B : B - 1;
        if (idx >= lo && idx <= N - B) return B;
    }
    return 1;
}

/* Upwind-biased value at FACE idx of cell-centred data; p points at cell idx. */
static inline double biased_face(const double *p, ptrdiff_t s, int left, int idx, int N, int bounded)
{
#ifdef OG_CENTERED2   /* libbreeze_oracle_centered2.so: advection = Centered(order = 2): symmetric 2-point interpolation */
    (void)left; (void)idx; (void)N; (void)bounded;
    return 0.5 * (p[-s] + p[0]);
#endif
    int B = buffer_at(idx, N, bounded, 1);
    if (B >= 4) {
        double v[9];
        for (int j = 0; j < 2 * B - 1; ++j) v[j] = left ? p[(j - B) * s] : p[(B - 1 - j) * s];
        return B == 5 ? weno_r5(v) : weno_r4(v);
    }
    if (B == 3)
        return left ? weno5(p[-3 * s], p[-2 * s], p[-s], p[0], p[s])
                    : weno5(p[2 * s], p[s], p[0], p[-s], p[-2 * s]);
    if (B == 2)
        return left ? weno3(p[-2 * s], p[-s], p[0]) : weno3(p[s], p[0], p[-s]);
    return left ? p[-s] : p[0];
}

/* Upwind-biased value at CENTRE idx of face data; p points at face idx. */
static inline double biased_center(const double *p, ptrdiff_t s, int left, int idx, int N, int bounded)
{
#ifdef OG_CENTERED2
    (void)left; (void)idx; (void)N; (void)bounded;
    return 0.5 * (p[0] + p[s]);
#endif
    int B = buffer_at(idx, N, bounded, 0);
    if (B >= 4) {
        double v[9];
        for (int j = 0; j < 2 * B - 1; ++j) v[j] = left ? p[(j - (B - 1)) * s] : p[(B - j) * s];
        return B == 5 ? weno_r5(v) : weno_r4(v);
    }
    if (B == 3)
        return left ? weno5(p[-2 * s], p[-s], p[0], p[s], p[2 * s])
                    : weno5(p[3 * s], p[2 * s], p[s], p[0], p[-s]);
    if (B == 2)
        return left ? weno3(p[-s], p[0], p[s]) : weno3(p[2 * s], p[s], p[0]);
    return left ? p[0] : p[s];
}

/* Centered(order 4) / Centered(order 2) symmetric interpolation of the four
 * values (q[-2], q[-1], q[0], q[+1]) straddling the target (between q[-1], q[0]). */
static inline double symm4(double qm2, double qm1, double q0, double qp1)
{
#ifdef OG_CENTERED2
    (void)qm2; (void)qp1;
    return 0.5 * (qm1 + q0);
#else
    return (7.0 / 12.0) * (qm1 + q0) - (1.0 / 12.0) * (qm2 + qp1);
#endif
}
static inline double symm2(double qm1, double q0) { return 0.5 * (qm1 + q0); }
/* Centered(order 2 (B - 1)) of the 2 (B - 1) values q[0 .. 2B-3] straddling the target (between q[B-2] and q[B-1]); B >= 4 */
static inline double symm_wide(const double *q, int B)
{
    const double *c = (B == 5) ? OGW_S8 : OGW_S6;
    const int h = B - 1;
    double acc = c[0] * (q[h - 1] + q[h]);
    for (int d = 1; d < h; ++d) acc += c[d] * (q[h - 1 - d] + q[h + d]);
    return acc;
}

/* bias(u) = u > 0 ? LeftBias : RightBias */
static inline int left_bias(double u) { return u > 0.0; }

/* ------------------------------------------------------------------------- */
/* Halo filling (Oceananigans fill_halo_regions!, recalled).    */
/* ------------------------------------------------------------------------- */

/* Periodic wrap in x and y over all z levels of the parent array (nz_tot levels). */
void og_fill_halo_periodic_xy(const og_grid *G, double *f, int nz_tot)
{
    size_t sx = SX(G), sy = SY(G);
    int Nx = G->Nx, Ny = G->Ny, Hx = G->Hx, Hy = G->Hy;
#pragma omp parallel for schedule(static)
    for (int kk = 0; kk < nz_tot; ++kk) {
        double *pl = f + sx * sy * (size_t)kk;
        if (G->tx == PERIODIC)
            for (int jj = Hy; jj < Hy + Ny; ++jj) {
                double *row = pl + sx * (size_t)jj;
                for (int h = 0; h < Hx; ++h) {
                    row[h] = row[h + Nx];
                    row[Hx + Nx + h] = row[Hx + h];
                }
            }
        if (G->ty == PERIODIC)
            for (int h = 0; h < Hy; ++h) {
                memcpy(pl + sx * (size_t)h, pl + sx * (size_t)(h + Ny), sx * sizeof(double));
                memcpy(pl + sx * (size_t)(Hy + Ny + h), pl + sx * (size_t)(Hy + h), sx * sizeof(double));
            }
    }
}

/* Bounded y (round 3: (Periodic, Bounded, Bounded), the reference benchmark's PBB option, benchmarking/run_benchmarks.jl:130): the same
 * conventions as Bounded z below — a field that is a centre in y gets its first halo row from the adjacent interior row (no-flux), a
 * y-face field (rho v, v) carries zeros on its wall faces j = 0 and j = Ny (impenetrable walls; face Ny lives in the first upper halo row). */
/* Bounded x ((Bounded, Flat, Bounded): examples/cloudy_thermal_bubble.jl, tropical_cyclone_with_rainband.jl): the same conventions along
 * the row — first halo cell of a centre-in-x field from the adjacent interior cell, wall faces i = 0 and i = Nx of an x-face field zero. */
void og_fill_halo_x_noflux(const og_grid *G, double *f, int nz_tot)
{
    size_t sx = SX(G), sy = SY(G);
    if (G->tx != BOUNDED) return;
#pragma omp parallel for schedule(static)
    for (int kk = 0; kk < nz_tot; ++kk)
        for (size_t jj = 0; jj < sy; ++jj) {
            double *row = f + sx * sy * (size_t)kk + sx * jj;
            row[G->Hx - 1] = row[G->Hx];
            row[G->Hx + G->Nx] = row[G->Hx + G->Nx - 1];
        }
}
void og_fill_halo_x_wall(const og_grid *G, double *u, int nz_tot)
{
    size_t sx = SX(G), sy = SY(G);
    if (G->tx != BOUNDED) return;
#pragma omp parallel for schedule(static)
    for (int kk = 0; kk < nz_tot; ++kk)
        for (size_t jj = 0; jj < sy; ++jj) {
            double *row = u + sx * sy * (size_t)kk + sx * jj;
            row[G->Hx] = 0.0;
            row[G->Hx + G->Nx] = 0.0;
        }
}
void og_fill_halo_y_noflux(const og_grid *G, double *f, int nz_tot)
{
    size_t sx = SX(G), sy = SY(G);
    if (G->ty != BOUNDED) return;
#pragma omp parallel for schedule(static)
    for (int kk = 0; kk < nz_tot; ++kk) {
        double *pl = f + sx * sy * (size_t)kk;
        memcpy(pl + sx * (size_t)(G->Hy - 1), pl + sx * (size_t)G->Hy, sx * sizeof(double));
        memcpy(pl + sx * (size_t)(G->Hy + G->Ny), pl + sx * (size_t)(G->Hy + G->Ny - 1), sx * sizeof(double));
    }
}
void og_fill_halo_y_wall(const og_grid *G, double *v, int nz_tot)
{
    size_t sx = SX(G), sy = SY(G);
    if (G->ty != BOUNDED) return;
#pragma omp parallel for schedule(static)
    for (int kk = 0; kk < nz_tot; ++kk) {
        double *pl = v + sx * sy * (size_t)kk;
        memset(pl + sx * (size_t)G->Hy, 0, sx * sizeof(double));
        memset(pl + sx * (size_t)(G->Hy + G->Ny), 0, sx * sizeof(double));
    }
}

/* Centre field on Bounded z with the default no-flux BC: first halo cell only,
 * c[-1] = c[0], c[Nz] = c[Nz-1]. */
void og_fill_halo_z_noflux(const og_grid *G, double *f)
{
    size_t pl = SX(G) * SY(G);
    double *b0 = f + pl * (size_t)(G->Hz);
    double *t0 = f + pl * (size_t)(G->Hz + G->Nz - 1);
    memcpy(b0 - pl, b0, pl * sizeof(double));
    memcpy(t0 + pl, t0, pl * sizeof(double));
}

/* z-face field on Bounded z, impenetrable walls: w[0] = w[Nz] = 0
 * (anelastic_time_stepping.jl:29 resets wall faces after each RK update). */
void og_fill_halo_z_wall(const og_grid *G, double *w)
{
    size_t pl = SX(G) * SY(G);
    memset(w + pl * (size_t)(G->Hz), 0, pl * sizeof(double));
    memset(w + pl * (size_t)(G->Hz + G->Nz), 0, pl * sizeof(double));
}

/* ------------------------------------------------------------------------- */
/* a8: velocities from momentum (update_atmosphere_model_state.jl:248-254)    */
/* launch covers k = 0..Nz (Face length on Bounded, :138-145).               */
/* ------------------------------------------------------------------------- */
void og_compute_velocities(const og_grid *G, double *u, double *v, double *w,
                           const double *ru, const double *rv, const double *rw)
{
    const double *rho = G->rho_r + G->Hz;
#pragma omp parallel for collapse(2) schedule(static)
    for (int k = 0; k <= G->Nz; ++k)
        for (int j = 0; j < G->Ny; ++j)
            for (int i = 0; i < G->Nx; ++i) {
                size_t n = IDX(G, i, j, k);
                double rc = rho[k];                       /* Ix, Iy of a column field */
                double rf = 0.5 * (rho[k - 1] + rho[k]);  /* Iz to face k */
                u[n] = ru[n] / (0.5 * (rc + rc));
                v[n] = rv[n] / (0.5 * (rc + rc));
                w[n] = rw[n] / rf;
            }
}

/* ------------------------------------------------------------------------- */
/* a9: theta, qv, T  (update_atmosphere_model_state.jl:256-292;               */
/*     potential_temperature_formulation.jl:115-145; dynamic_states.jl:31-58) */
/* ------------------------------------------------------------------------- */
void og_compute_thermo(const og_grid *G, double *theta, double *qv, double *T,
                       const double *rtheta, const double *rq)
{
    const double *rho = G->rho_r + G->Hz, *pr = G->p_r + G->Hz;
#pragma omp parallel for collapse(2) schedule(static)
    for (int k = 0; k < G->Nz; ++k)
        for (int j = 0; j < G->Ny; ++j)
            for (int i = 0; i < G->Nx; ++i) {
                size_t n = IDX(G, i, j, k);
                double th = rtheta[n] / rho[k];
                double q = rq[n] / rho[k];
                theta[n] = th;
                qv[n] = q;
                double qd = 1.0 - (q + 0.0 + 0.0);
                double Rm = qd * G->Rd + q * G->Rv;
                double cpm = qd * G->cpd + q * G->cpv + 0.0 + 0.0;
                double Pi = pow(pr[k] / G->p_st, Rm / cpm);
                T[n] = Pi * th + 0.0;
            }
}

/* ------------------------------------------------------------------------- */
/* a1/a5/a6: scalar tendency  G = -div_rhoUc  (src/Advection.jl:20-35)        */
/* ------------------------------------------------------------------------- */
static inline double flux_x_scalar(const og_grid *G, const double *u, const double *c, int i, int j, int k)
{
    size_t n = IDX(G, i, j, k);
    double ut = u[n];
    double cr = biased_face(c + n, 1, left_bias(ut), i, G->Nx, G->tx == BOUNDED);
    double Ax = G->dy * G->dzc[k + G->Hz];
    double rho = G->rho_r[k + G->Hz];
    return (0.5 * (rho + rho)) * (Ax * ut * cr);
}
static inline double flux_y_scalar(const og_grid *G, const double *v, const double *c, int i, int j, int k)
{
    size_t n = IDX(G, i, j, k);
    double vt = v[n];
    double cr = biased_face(c + n, (ptrdiff_t)SX(G), left_bias(vt), j, G->Ny, G->ty == BOUNDED);
    double Ay = G->dx * G->dzc[k + G->Hz];
    double rho = G->rho_r[k + G->Hz];
    return (0.5 * (rho + rho)) * (Ay * vt * cr);
}
static inline double flux_z_scalar(const og_grid *G, const double *w, const double *c, int i, int j, int k)
{
    size_t n = IDX(G, i, j, k);
    double wt = w[n];
    double cr = biased_face(c + n, (ptrdiff_t)(SX(G) * SY(G)), left_bias(wt), k, G->Nz, G->tz == BOUNDED);
    double Az = G->dx * G->dy;
    const double *rho = G->rho_r + G->Hz;
    return (0.5 * (rho[k - 1] + rho[k])) * (Az * wt * cr);
}

void og_scalar_tendency(const og_grid *G, double *Gc, const double *u, const double *v,
                        const double *w, const double *c)
{
#pragma omp parallel for collapse(2) schedule(static)
    for (int k = 0; k < G->Nz; ++k)
        for (int j = 0; j < G->Ny; ++j)
            for (int i = 0; i < G->Nx; ++i) {
                double Vinv = 1.0 / (G->dx * G->dy * G->dzc[k + G->Hz]);
                double dxF = 0.0, dyF = 0.0, dzF = 0.0;
                if (G->tx != FLAT)
                    dxF = flux_x_scalar(G, u, c, i + 1, j, k) - flux_x_scalar(G, u, c, i, j, k);
                if (G->ty != FLAT)
                    dyF = flux_y_scalar(G, v, c, i, j + 1, k) - flux_y_scalar(G, v, c, i, j, k);
                if (G->tz != FLAT)
                    dzF = flux_z_scalar(G, w, c, i, j, k + 1) - flux_z_scalar(G, w, c, i, j, k);
                Gc[IDX(G, i, j, k)] = -(Vinv * (dxF + dyF + dzF));
            }
}


/* ------------------------------------------------------------------------- */
/* f4: bounds-preserving WENO for moisture, WENO(order = 5, bounds = (lo, hi)). */
/* Breeze: div_rhoUc(i, j, k, grid, advection::BoundsPreservingWENO, rho, U, c) =  */
/*   V^-1 (bounded_tracer_flux_divergence_x + _y + _z)   (src/Advection.jl:42-47).  */
/* The three divergences are Oceananigans' (0.110.x, not vendored; PARITY UNPINNED): */
/* the positivity-preserving limiter of Zhang & Shu (2010) as Oceananigans applies  */
/* it to its WENO reconstructions (positivity_preserving_tracer_advection_operators): */
/*   c+L, c+R = left / right biased value at the cell's upper face, c-L, c-R at its   */
/*   lower face;  p~ = (c - w1 c-R - wn c+L) / (1 - 2 w1), w1 = wn = 5/18;            */
/*   M, m = max, min(p~, c+L, c-R);  theta = min(|(hi - c)/(M - c + eps2)|,           */
/*   |(lo - c)/(m - c + eps2)|, 1), eps2 = 1e-20;  the two reconstructions that start */
/*   in this cell are limited, c+L <- theta (c+L - c) + c, c-R likewise;  divergence = */
/*   rho A u (+) (c+L, c+R) at the upper face minus the same at the lower face with    */
/*   upwind_biased_product(u, L, R) = ((u + |u|) L + (u - |u|) R) / 2.                 */
/* The density rides on the face as in tracer_mass_flux_* (src/Advection.jl:20-27).    */
/* Note: theta belongs to the cell, the inflow value of a face is the neighbour's      */
/* unlimited reconstruction: not strictly conservative / bounds-preserving (tests).    */
/* ------------------------------------------------------------------------- */
static inline double upwind_biased_product(double u, double cl, double cr)
{
    return ((u + fabs(u)) * cl + (u - fabs(u)) * cr) / 2.0;
}
static inline double bounded_div_1d(const double *c, ptrdiff_t s, int idx, int N, int bounded, double lo, double hi,
                                    double flux_hi /* rho A u at face idx+1 */, double flux_lo /* at face idx */)
{
    const double w1 = 5.0 / 18.0, eps2 = 1e-20;
    const double cij = c[0];
    double cpL = biased_face(c + s, s, 1, idx + 1, N, bounded);
    double cpR = biased_face(c + s, s, 0, idx + 1, N, bounded);
    double cmL = biased_face(c, s, 1, idx, N, bounded);
    double cmR = biased_face(c, s, 0, idx, N, bounded);
    double pt = (cij - w1 * cmR - w1 * cpL) / (1.0 - 2.0 * w1);
    double M = fmax(pt, fmax(cpL, cmR));
    double m = fmin(pt, fmin(cpL, cmR));
    double th = fmin(fmin(fabs((hi - cij) / (M - cij + eps2)), fabs((lo - cij) / (m - cij + eps2))), 1.0);
    cpL = th * (cpL - cij) + cij;
    cmR = th * (cmR - cij) + cij;
    return upwind_biased_product(flux_hi, cpL, cpR) - upwind_biased_product(flux_lo, cmL, cmR);
}

void og_scalar_tendency_bounded(const og_grid *G, double *Gc, const double *u, const double *v, const double *w,
                                const double *c, double lo, double hi)
{
    const ptrdiff_t sy = (ptrdiff_t)SX(G), sz = (ptrdiff_t)(SX(G) * SY(G));
    const double *rho = G->rho_r + G->Hz;
#pragma omp parallel for collapse(2) schedule(static)
    for (int k = 0; k < G->Nz; ++k)
        for (int j = 0; j < G->Ny; ++j)
            for (int i = 0; i < G->Nx; ++i) {
                size_t n = IDX(G, i, j, k);
                double dzc = G->dzc[k + G->Hz];
                double Vinv = 1.0 / (G->dx * G->dy * dzc);
                double Ax = G->dy * dzc, Ay = G->dx * dzc, Az = G->dx * G->dy;
                double rc = 0.5 * (rho[k] + rho[k]);
                double dxF = 0.0, dyF = 0.0, dzF = 0.0;
                if (G->tx != FLAT)
                    dxF = bounded_div_1d(c + n, 1, i, G->Nx, G->tx == BOUNDED, lo, hi, rc * (Ax * u[n + 1]), rc * (Ax * u[n]));
                if (G->ty != FLAT)
                    dyF = bounded_div_1d(c + n, sy, j, G->Ny, G->ty == BOUNDED, lo, hi, rc * (Ay * v[n + sy]), rc * (Ay * v[n]));
                if (G->tz != FLAT)
                    dzF = bounded_div_1d(c + n, sz, k, G->Nz, G->tz == BOUNDED, lo, hi,
                                         (0.5 * (rho[k] + rho[k + 1])) * (Az * w[n + sz]), (0.5 * (rho[k - 1] + rho[k])) * (Az * w[n]));
                Gc[n] = -(Vinv * (dxF + dyF + dzF));
            }
}

/* ------------------------------------------------------------------------- */
/* a2/a3: momentum tendencies (dynamics_kernel_functions.jl:54-130).          */
/* Advecting field = momentum (rho u), advected = velocity component.        */
/* Each helper returns one advective momentum flux at its natural location.  */
/* ------------------------------------------------------------------------- */
#define STRX ((ptrdiff_t)1)
#define STRY(G) ((ptrdiff_t)SX(G))
#define STRZ(G) ((ptrdiff_t)(SX(G) * SY(G)))

static inline double dzc_at(const og_grid *G, int k) { return G->dzc[k + G->Hz]; }

/* symmetric interpolation along x of q = A*M (A constant along x) */
static inline double symm_x_center(const og_grid *G, const double *M, size_t n, int i, double A)
{   /* to centre i from faces; a Bounded x drops the order next to the walls like z and y */
    if (G->tx == FLAT) return A * M[n];
    if (G->tx == BOUNDED) {
        int B = buffer_at(i, G->Nx, 1, 0);
        if (B >= 4) {
            double q[8]; const int h = B - 1;
            for (int m = 0; m < 2 * h; ++m) q[m] = A * M[n + (m - (h - 1))];
            return symm_wide(q, B);
        }
        if (B == 3) return symm4(A * M[n - 1], A * M[n], A * M[n + 1], A * M[n + 2]);
        return symm2(A * M[n], A * M[n + 1]);
    }
    if (og_weno_R >= 4) {        /* Centered(order 2 (R - 1)): faces i-(R-2) .. i+R-1 */
        double q[8]; const int h = og_weno_R - 1;
        for (int m = 0; m < 2 * h; ++m) q[m] = A * M[n + (m - (h - 1))];
        return symm_wide(q, og_weno_R);
    }
    return symm4(A * M[n - 1], A * M[n], A * M[n + 1], A * M[n + 2]);
}
static inline double symm_x_face(const og_grid *G, const double *M, size_t n, int i, double A)
{   /* to face i from centres */
    if (G->tx == FLAT) return A * M[n];
    if (G->tx == BOUNDED) {
        int B = buffer_at(i, G->Nx, 1, 1);
        if (B >= 4) {
            double q[8]; const int h = B - 1;
            for (int m = 0; m < 2 * h; ++m) q[m] = A * M[n + (m - h)];
            return symm_wide(q, B);
        }
        if (B == 3) return symm4(A * M[n - 2], A * M[n - 1], A * M[n], A * M[n + 1]);
        return symm2(A * M[n - 1], A * M[n]);
    }
    if (og_weno_R >= 4) {        /* centres i-(R-1) .. i+R-2 */
        double q[8]; const int h = og_weno_R - 1;
        for (int m = 0; m < 2 * h; ++m) q[m] = A * M[n + (m - h)];
        return symm_wide(q, og_weno_R);
    }
    return symm4(A * M[n - 2], A * M[n - 1], A * M[n], A * M[n + 1]);
}
/* Bounded y: the order drops with the distance to the wall like the z interpolations below (order 2 (B - 1), B the WENO buffer that fits) */
static inline double symm_y_center(const og_grid *G, const double *M, size_t n, int j, double A)
{
    ptrdiff_t s = STRY(G);
    if (G->ty == FLAT) return A * M[n];
    if (G->ty == BOUNDED) {      /* to centre j from faces j-1 .. j+2 */
        int B = buffer_at(j, G->Ny, 1, 0);
        if (B >= 4) {
            double q[8]; const int h = B - 1;
            for (int m = 0; m < 2 * h; ++m) q[m] = A * M[n + (m - (h - 1)) * s];
            return symm_wide(q, B);
        }
        if (B == 3) return symm4(A * M[n - s], A * M[n], A * M[n + s], A * M[n + 2 * s]);
        return symm2(A * M[n], A * M[n + s]);
    }
    if (og_weno_R >= 4) {
        double q[8]; const int h = og_weno_R - 1;
        for (int m = 0; m < 2 * h; ++m) q[m] = A * M[n + (m - (h - 1)) * s];
        return symm_wide(q, og_weno_R);
    }
    return symm4(A * M[n - s], A * M[n], A * M[n + s], A * M[n + 2 * s]);
}
static inline double symm_y_face(const og_grid *G, const double *M, size_t n, int j, double A)
{
    ptrdiff_t s = STRY(G);
    if (G->ty == FLAT) return A * M[n];
    if (G->ty == BOUNDED) {      /* to face j from centres j-2 .. j+1 */
        int B = buffer_at(j, G->Ny, 1, 1);
        if (B >= 4) {
            double q[8]; const int h = B - 1;
            for (int m = 0; m < 2 * h; ++m) q[m] = A * M[n + (m - h) * s];
            return symm_wide(q, B);
        }
        if (B == 3) return symm4(A * M[n - 2 * s], A * M[n - s], A * M[n], A * M[n + s]);
        return symm2(A * M[n - s], A * M[n]);
    }
    if (og_weno_R >= 4) {
        double q[8]; const int h = og_weno_R - 1;
        for (int m = 0; m < 2 * h; ++m) q[m] = A * M[n + (m - h) * s];
        return symm_wide(q, og_weno_R);
    }
    return symm4(A * M[n - 2 * s], A * M[n - s], A * M[n], A * M[n + s]);
}
/* z is Bounded: order 4 only where the WENO5 buffer fits, else order 2.
 * The area may depend on k (Ax, Ay ~ dz(k)); Az does not. */
static inline double symm_z_face_area(const og_grid *G, const double *M, size_t n, int k, double Ah /* horizontal factor */)
{   /* q(k) = Ah*dzc(k)*M(k), to face k from centres k-2..k+1 */
    ptrdiff_t s = STRZ(G);
    int B = buffer_at(k, G->Nz, G->tz == BOUNDED, 1);
    if (B >= 4) {
        double q[8]; const int h = B - 1;
        for (int m = 0; m < 2 * h; ++m) q[m] = Ah * dzc_at(G, k + m - h) * M[n + (m - h) * s];
        return symm_wide(q, B);
    }
    if (B == 3)
        return symm4(Ah * dzc_at(G, k - 2) * M[n - 2 * s], Ah * dzc_at(G, k - 1) * M[n - s],
                     Ah * dzc_at(G, k) * M[n], Ah * dzc_at(G, k + 1) * M[n + s]);
    return symm2(Ah * dzc_at(G, k - 1) * M[n - s], Ah * dzc_at(G, k) * M[n]);
}
static inline double symm_z_center(const og_grid *G, const double *M, size_t n, int k, double A)
{   /* to centre k from faces k-1..k+2, area constant */
    ptrdiff_t s = STRZ(G);
    int B = buffer_at(k, G->Nz, G->tz == BOUNDED, 0);
    if (B >= 4) {
        double q[8]; const int h = B - 1;
        for (int m = 0; m < 2 * h; ++m) q[m] = A * M[n + (m - (h - 1)) * s];
        return symm_wide(q, B);
    }
    if (B == 3) return symm4(A * M[n - s], A * M[n], A * M[n + s], A * M[n + 2 * s]);
    return symm2(A * M[n], A * M[n + s]);
}

/* ---- fluxes of u-momentum ---- */
static inline double F_Uu(const og_grid *G, const double *ru, const double *u, int i, int j, int k)
{   /* at (c,c,c) index i */
    size_t n = IDX(G, i, j, k);
    double ut = symm_x_center(G, ru, n, i, G->dy * dzc_at(G, k));
    double uR = biased_center(u + n, STRX, left_bias(ut), i, G->Nx, G->tx == BOUNDED);
    return ut * uR;
}
static inline double F_Vu(const og_grid *G, const double *rv, const double *u, int i, int j, int k)
{   /* at (f,f,c) */
    size_t n = IDX(G, i, j, k);
    double vt = symm_x_face(G, rv, n, i, G->dx * dzc_at(G, k));
    double uR = biased_face(u + n, STRY(G), left_bias(vt), j, G->Ny, G->ty == BOUNDED);
    return vt * uR;
}
static inline double F_Wu(const og_grid *G, const double *rw, const double *u, int i, int j, int k)
{   /* at (f,c,f) */
    size_t n = IDX(G, i, j, k);
    double wt = symm_x_face(G, rw, n, i, G->dx * G->dy);
    double uR = biased_face(u + n, STRZ(G), left_bias(wt), k, G->Nz, G->tz == BOUNDED);
    return wt * uR;
}
/* ---- fluxes of v-momentum ---- */
static inline double F_Uv(const og_grid *G, const double *ru, const double *v, int i, int j, int k)
{   /* at (f,f,c) */
    size_t n = IDX(G, i, j, k);
    double ut = symm_y_face(G, ru, n, j, G->dy * dzc_at(G, k));
    double vR = biased_face(v + n, STRX, left_bias(ut), i, G->Nx, G->tx == BOUNDED);
    return ut * vR;
}
static inline double F_Vv(const og_grid *G, const double *rv, const double *v, int i, int j, int k)
{   /* at (c,c,c) index j */
    size_t n = IDX(G, i, j, k);
    double vt = symm_y_center(G, rv, n, j, G->dx * dzc_at(G, k));
    double vR = biased_center(v + n, STRY(G), left_bias(vt), j, G->Ny, G->ty == BOUNDED);
    return vt * vR;
}
static inline double F_Wv(const og_grid *G, const double *rw, const double *v, int i, int j, int k)
{   /* at (c,f,f) */
    size_t n = IDX(G, i, j, k);
    double wt = symm_y_face(G, rw, n, j, G->dx * G->dy);
    double vR = biased_face(v + n, STRZ(G), left_bias(wt), k, G->Nz, G->tz == BOUNDED);
    return wt * vR;
}
/* ---- fluxes of w-momentum ---- */
static inline double F_Uw(const og_grid *G, const double *ru, const double *w, int i, int j, int k)
{   /* at (f,c,f) */
    size_t n = IDX(G, i, j, k);
    double ut = symm_z_face_area(G, ru, n, k, G->dy);
    double wR = biased_face(w + n, STRX, left_bias(ut), i, G->Nx, G->tx == BOUNDED);
    return ut * wR;
}
static inline double F_Vw(const og_grid *G, const double *rv, const double *w, int i, int j, int k)
{   /* at (c,f,f) */
    size_t n = IDX(G, i, j, k);
    double vt = symm_z_face_area(G, rv, n, k, G->dx);
    double wR = biased_face(w + n, STRY(G), left_bias(vt), j, G->Ny, G->ty == BOUNDED);
    return vt * wR;
}
static inline double F_Ww(const og_grid *G, const double *rw, const double *w, int i, int j, int k)
{   /* at (c,c,c) index k */
    size_t n = IDX(G, i, j, k);
    double wt = symm_z_center(G, rw, n, k, G->dx * G->dy);
    double wR = biased_center(w + n, STRZ(G), left_bias(wt), k, G->Nz, G->tz == BOUNDED);
    return wt * wR;
}

void og_u_tendency(const og_grid *G, double *Gu, const double *ru, const double *rv,
                   const double *rw, const double *u)
{
    const int i0 = (G->tx == BOUNDED) ? 1 : 0;      /* Bounded x: the wall face i = 0 is never updated */
#pragma omp parallel for collapse(2) schedule(static)
    for (int k = 0; k < G->Nz; ++k)
        for (int j = 0; j < G->Ny; ++j)
            for (int i = i0; i < G->Nx; ++i) {
                double Vinv = 1.0 / (G->dx * G->dy * dzc_at(G, k));
                double a = 0.0, b = 0.0, c = 0.0;
                if (G->tx != FLAT) a = F_Uu(G, ru, u, i, j, k) - F_Uu(G, ru, u, i - 1, j, k);
                if (G->ty != FLAT) b = F_Vu(G, rv, u, i, j + 1, k) - F_Vu(G, rv, u, i, j, k);
                if (G->tz != FLAT) c = F_Wu(G, rw, u, i, j, k + 1) - F_Wu(G, rw, u, i, j, k);
                Gu[IDX(G, i, j, k)] = -(Vinv * (a + b + c));
            }
}

void og_v_tendency(const og_grid *G, double *Gv, const double *ru, const double *rv,
                   const double *rw, const double *v)
{
    const int j0 = (G->ty == BOUNDED) ? 1 : 0;      /* Bounded y: the wall face j = 0 is never updated (like w at k = 0) */
#pragma omp parallel for collapse(2) schedule(static)
    for (int k = 0; k < G->Nz; ++k)
        for (int j = j0; j < G->Ny; ++j)
            for (int i = 0; i < G->Nx; ++i) {
                double Vinv = 1.0 / (G->dx * G->dy * dzc_at(G, k));
                double a = 0.0, b = 0.0, c = 0.0;
                if (G->tx != FLAT) a = F_Uv(G, ru, v, i + 1, j, k) - F_Uv(G, ru, v, i, j, k);
                if (G->ty != FLAT) b = F_Vv(G, rv, v, i, j, k) - F_Vv(G, rv, v, i, j - 1, k);
                if (G->tz != FLAT) c = F_Wv(G, rw, v, i, j, k + 1) - F_Wv(G, rw, v, i, j, k);
                Gv[IDX(G, i, j, k)] = -(Vinv * (a + b + c));
            }
}

/* a4: anelastic buoyancy at cell centre (anelastic_buoyancy.jl:36-72) */
static inline double buoyancy_ccc(const og_grid *G, const double *T, const double *qv, int i, int j, int k)
{
    size_t n = IDX(G, i, j, k);
    double q = qv[n];
    double rho_r = G->rho_r[k + G->Hz], Tr = G->T_r[k + G->Hz];
    double Rmr = (1.0 - (0.0 + 0.0 + 0.0)) * G->Rd + 0.0 * G->Rv;
    double Rm = (1.0 - (q + 0.0 + 0.0)) * G->Rd + q * G->Rv;
    double rhop = rho_r * (Rmr * Tr / (Rm * T[n]) - 1.0);
    return -G->g * rhop;
}

/* Gw is only defined on interior faces k = 1..Nz-1; wall faces are never
 * updated (the reference overwrites them with 0). */
void og_w_tendency(const og_grid *G, double *Gw, const double *ru, const double *rv,
                   const double *rw, const double *w, const double *T, const double *qv)
{
#pragma omp parallel for collapse(2) schedule(static)
    for (int k = 1; k < G->Nz; ++k)
        for (int j = 0; j < G->Ny; ++j)
            for (int i = 0; i < G->Nx; ++i) {
                double Vinv = 1.0 / (G->dx * G->dy * G->dzf[k + G->Hz]);
                double a = 0.0, b = 0.0, c = 0.0;
                if (G->tx != FLAT) a = F_Uw(G, ru, w, i + 1, j, k) - F_Uw(G, ru, w, i, j, k);
                if (G->ty != FLAT) b = F_Vw(G, rv, w, i, j + 1, k) - F_Vw(G, rv, w, i, j, k);
                c = F_Ww(G, rw, w, i, j, k) - F_Ww(G, rw, w, i, j, k - 1);
                double bf = 0.5 * (buoyancy_ccc(G, T, qv, i, j, k - 1) + buoyancy_ccc(G, T, qv, i, j, k));
                Gw[IDX(G, i, j, k)] = -(Vinv * (a + b + c)) + bf;
            }
}

/* ------------------------------------------------------------------------- */
/* a10: SSP-RK3 substep (ssp_runge_kutta_3.jl:167-173).                       */
/* k range [k0,k1): centres 0..Nz, w-faces 1..Nz (walls never updated).       */
/* ------------------------------------------------------------------------- */
void og_rk3_substep(const og_grid *G, double *u, const double *u0, const double *Gn,
                    double dt, double alpha, int k0, int k1)
{
#pragma omp parallel for collapse(2) schedule(static)
    for (int k = k0; k < k1; ++k)
        for (int j = 0; j < G->Ny; ++j)
            for (int i = 0; i < G->Nx; ++i) {
                size_t n = IDX(G, i, j, k);
                u[n] = (1.0 - alpha) * u0[n] + alpha * (u[n] + dt * Gn[n]);
            }
}

/* ------------------------------------------------------------------------- */
/* a11: Poisson source term (anelastic_pressure_solver.jl:99-105):            */
/*   rhs = dz_c * div(rho U) / dt, written into a halo-free Nx*Ny*Nz array.   */
/* ------------------------------------------------------------------------- */
void og_poisson_source(const og_grid *G, double *rhs, const double *ru, const double *rv,
                       const double *rw, double dt)
{
#pragma omp parallel for collapse(2) schedule(static)
    for (int k = 0; k < G->Nz; ++k)
        for (int j = 0; j < G->Ny; ++j)
            for (int i = 0; i < G->Nx; ++i) {
                double dz = dzc_at(G, k);
                double Ax = G->dy * dz, Ay = G->dx * dz, Az = G->dx * G->dy;
                double Vinv = 1.0 / (G->dx * G->dy * dz);
                size_t n = IDX(G, i, j, k);
                double a = 0.0, b = 0.0, c = 0.0;
                if (G->tx != FLAT) a = Ax * ru[n + 1] - Ax * ru[n];
                if (G->ty != FLAT) b = Ay * rv[n + STRY(G)] - Ay * rv[n];
                if (G->tz != FLAT) c = Az * rw[n + STRZ(G)] - Az * rw[n];
                double div = Vinv * (a + b + c);
                rhs[(size_t)i + (size_t)G->Nx * ((size_t)j + (size_t)G->Ny * (size_t)k)] = dz * div / dt;
            }
}

/* a11: tridiagonal coefficients (anelastic_pressure_solver.jl:39-78).
 * lower[k], k = 0..Nz-2 couples levels k and k+1; diag0[k] is the
 * lambda-independent part of the main diagonal; mass[k] = rho_k*dz_k multiplies
 * (lambda_x + lambda_y). */
void og_poisson_coefficients(const og_grid *G, double *lower, double *diag0, double *mass)
{
    const double *rho = G->rho_r + G->Hz;
    int Nz = G->Nz;
    for (int k = 0; k < Nz - 1; ++k)
        lower[k] = (0.5 * (rho[k] + rho[k + 1])) / G->dzf[k + 1 + G->Hz];
    for (int k = 0; k < Nz; ++k) {
        double up = (k < Nz - 1) ? (0.5 * (rho[k] + rho[k + 1])) / G->dzf[k + 1 + G->Hz] : 0.0;
        double dn = (k > 0) ? (0.5 * (rho[k - 1] + rho[k])) / G->dzf[k + G->Hz] : 0.0;
        if (k == 0) diag0[k] = -up;
        else if (k == Nz - 1) diag0[k] = -dn;
        else diag0[k] = -(up + dn);
        mass[k] = rho[k] * dzc_at(G, k);
    }
}

/* Batched complex Thomas solve along z (Oceananigans BatchedTridiagonalSolver,
 * recalled).  f and phi are complex arrays (re,im interleaved) of
 * shape Nz x Ny x Nx (x fastest).  phi must hold the "stale" values used when
 * the forward update is elided (|beta| <= 10 eps); callers pass zeros.
 * lam[j*Nx+i] = lambda_x[i] + lambda_y[j]. */
void og_tridiagonal_solve(int Nx, int Ny, int Nz, const double *lower, const double *diag0,
                          const double *mass, const double *lam, const double *f, double *phi,
                          double *scratch /* Nz doubles per thread: use Nx*Ny*Nz */)
{
    size_t pl = (size_t)Nx * Ny;
    const double tiny = 10.0 * 2.220446049250313e-16;
#pragma omp parallel for schedule(static)
    for (long c = 0; c < (long)pl; ++c) {
        double *t = scratch + (size_t)c * Nz;
        double l = lam[c];
        double beta = diag0[0] - mass[0] * l;
        phi[2 * c] = f[2 * c] / beta;
        phi[2 * c + 1] = f[2 * c + 1] / beta;
        for (int k = 1; k < Nz; ++k) {
            size_t n = (size_t)c + pl * k, m = (size_t)c + pl * (k - 1);
            double ck = lower[k - 1], ak = lower[k - 1];
            double bk = diag0[k] - mass[k] * l;
            t[k] = ck / beta;
            beta = bk - ak * t[k];
            if (fabs(beta) > tiny) {
                phi[2 * n] = (f[2 * n] - ak * phi[2 * m]) / beta;
                phi[2 * n + 1] = (f[2 * n + 1] - ak * phi[2 * m + 1]) / beta;
            }
        }
        for (int k = Nz - 2; k >= 0; --k) {
            size_t n = (size_t)c + pl * k, p = (size_t)c + pl * (k + 1);
            phi[2 * n] -= t[k + 1] * phi[2 * p];
            phi[2 * n + 1] -= t[k + 1] * phi[2 * p + 1];
        }
    }
}

/* a12: projection (anelastic_time_stepping.jl:45-54), k = 0..Nz-1. */
void og_pressure_correct(const og_grid *G, double *ru, double *rv, double *rw,
                         const double *phi, double dt)
{
    const double *rho = G->rho_r + G->Hz;
#pragma omp parallel for collapse(2) schedule(static)
    for (int k = 0; k < G->Nz; ++k)
        for (int j = 0; j < G->Ny; ++j)
            for (int i = 0; i < G->Nx; ++i) {
                size_t n = IDX(G, i, j, k);
                double rf = 0.5 * (rho[k - 1] + rho[k]);
                double rc = rho[k];
                if (G->tx != FLAT && !(G->tx == BOUNDED && i == 0)) ru[n] -= rc * dt * ((phi[n] - phi[n - 1]) * (1.0 / G->dx));
                if (G->ty != FLAT && !(G->ty == BOUNDED && j == 0)) rv[n] -= rc * dt * ((phi[n] - phi[n - STRY(G)]) * (1.0 / G->dy));
                rw[n] -= rf * dt * ((phi[n] - phi[n - STRZ(G)]) * (1.0 / G->dzf[k + G->Hz]));
            }
}

/* divergence diagnostic used by the restated reference test
 * (test/anelastic_pressure_solver_nonhydrostatic.jl:45-46). */
void og_divergence(const og_grid *G, double *div, const double *ru, const double *rv, const double *rw)
{
#pragma omp parallel for collapse(2) schedule(static)
    for (int k = 0; k < G->Nz; ++k)
        for (int j = 0; j < G->Ny; ++j)
            for (int i = 0; i < G->Nx; ++i) {
                double dz = dzc_at(G, k);
                double Ax = G->dy * dz, Ay = G->dx * dz, Az = G->dx * G->dy;
                double Vinv = 1.0 / (G->dx * G->dy * dz);
                size_t n = IDX(G, i, j, k);
                double a = 0.0, b = 0.0, c = 0.0;
                if (G->tx != FLAT) a = Ax * ru[n + 1] - Ax * ru[n];
                if (G->ty != FLAT) b = Ay * rv[n + STRY(G)] - Ay * rv[n];
                if (G->tz != FLAT) c = Az * rw[n + STRZ(G)] - Az * rw[n];
                div[(size_t)i + (size_t)G->Nx * ((size_t)j + (size_t)G->Ny * (size_t)k)] = Vinv * (a + b + c);
            }
}

/* exposed for unit tests of the reconstruction itself */
double og_weno5(double a, double b, double c, double d, double e) { return weno5(a, b, c, d, e); }
double og_weno3(double a, double b, double c) { return weno3(a, b, c); }
double og_weno_generic(int r, const double *v) { return r == 5 ? weno_r5(v) : r == 4 ? weno_r4(v) : weno_r3(v); }
int og_buffer_at(int idx, int N, int bounded, int at_face) { return buffer_at(idx, N, bounded, at_face); }

/* Compressible split-explicit path. */
#include "breeze_oracle_compressible.inc.c"

/* ------------------------------------------------------------------------- */
/* StaticEnergy formulation                                    */
/* e = rho_e / rho_r;  T = (e - g z + 0 + 0) / c_pm                           */
/*   static_energy_formulation.jl:70-78, dynamic_states.jl:283-298            */
/* G_rho_e = -div_rhoUc(e) - Iz_c(w * Iz_f(buoyancy))                         */
/*   static_energy_tendency.jl:39-72, dynamics_kernel_functions.jl:40-51      */
/* zc: cell-centre heights, halo-inclusive (entry k+Hz)                       */
/* ------------------------------------------------------------------------- */
void og_compute_thermo_energy(const og_grid *G, double *e, double *qv, double *T,
                              const double *re, const double *rq, const double *zc)
{
    const double *rho = G->rho_r + G->Hz;
#pragma omp parallel for collapse(2) schedule(static)
    for (int k = 0; k < G->Nz; ++k)
        for (int j = 0; j < G->Ny; ++j)
            for (int i = 0; i < G->Nx; ++i) {
                size_t n = IDX(G, i, j, k);
                double ev = re[n] / rho[k];
                double q = rq[n] / rho[k];
                e[n] = ev;
                qv[n] = q;
                double qd = 1.0 - (q + 0.0 + 0.0);
                double cpm = qd * G->cpd + q * G->cpv + 0.0 + 0.0;
                T[n] = (ev - G->g * zc[k + G->Hz] + 0.0 + 0.0) / cpm;
            }
}

void og_energy_buoyancy_flux(const og_grid *G, double *Ge, const double *w, const double *T, const double *qv)
{
#pragma omp parallel for collapse(2) schedule(static)
    for (int k = 0; k < G->Nz; ++k)
        for (int j = 0; j < G->Ny; ++j)
            for (int i = 0; i < G->Nx; ++i) {
                size_t n = IDX(G, i, j, k);
                double b_m = buoyancy_ccc(G, T, qv, i, j, k - 1);
                double b_0 = buoyancy_ccc(G, T, qv, i, j, k);
                double b_p = buoyancy_ccc(G, T, qv, i, j, k + 1);
                double f_lo = ((b_0 + b_m) / 2.0) * w[n];
                double f_hi = ((b_p + b_0) / 2.0) * w[n + STRZ(G)];
                Ge[n] = Ge[n] - (f_hi + f_lo) / 2.0;
            }
}

/* ------------------------------------------------------------------------- */
/* Warm-phase SaturationAdjustment                        */
/*   _compute_auxiliary_thermodynamic_variables! with                        */
/*   maybe_adjust_thermodynamic_state  (update_atmosphere_model_state.jl:256-292, */
/*   src/Microphysics/saturation_adjustment.jl:82-86,121-126,168-235),          */
/*   Clausius-Clapeyron (src/Thermodynamics/clausius_clapeyron.jl:59-68),      */
/*   secant_solve (src/Solvers.jl:243-262).  Scalar twin: oracle/thermo.py.    */
/* ------------------------------------------------------------------------- */
typedef struct {
    double Ll, cl;          /* liquid reference latent heat, heat capacity */
    double T_energy;        /* energy_reference_temperature */
    double Ttr, ptr;        /* triple point */
    double abstol;          /* SecantSolver(abstol, reltol = 0) */
    int maxiter;
} og_sa;

static inline double sa_psat(const og_grid *G, const og_sa *S, double T)
{
    double dc = G->cpv - S->cl;
    double L0 = S->Ll - dc * S->T_energy;
    return S->ptr * pow(T / S->Ttr, dc / G->Rv) * exp((1.0 / S->Ttr - 1.0 / T) * L0 / G->Rv);
}
static inline double sa_Rm(const og_grid *G, double qv, double ql) { return (1.0 - (qv + ql + 0.0)) * G->Rd + qv * G->Rv; }
static inline double sa_cpm(const og_grid *G, const og_sa *S, double qv, double ql)
{
    return (1.0 - (qv + ql + 0.0)) * G->cpd + qv * G->cpv + ql * S->cl + 0.0;
}
static inline double sa_T(const og_grid *G, const og_sa *S, double th, double qv, double ql, double pr)
{
    double cpm = sa_cpm(G, S, qv, ql);
    return pow(pr / G->p_st, sa_Rm(G, qv, ql) / cpm) * th + (S->Ll * ql + 0.0) / cpm;
}
static inline void sa_adjust(const og_grid *G, const og_sa *S, double T, double qt, double pr, double *qv, double *ql)
{
    double ps = sa_psat(G, S, T);
    double qs = (G->Rd / G->Rv) * (1.0 - qt) * ps / (pr - ps);
    *ql = fmax(0.0, qt - qs);
    *qv = qt - *ql;
}
static inline double sa_residual(const og_grid *G, const og_sa *S, double T, double th, double qt, double pr)
{
    double qv, ql;
    sa_adjust(G, S, T, qt, pr, &qv, &ql);
    return T - sa_T(G, S, th, qv, ql, pr);
}

void og_compute_thermo_sa(const og_grid *G, const og_sa *S, double *theta, double *qe, double *qv_out, double *ql_out,
                          double *T, const double *rtheta, const double *rq)
{
    const double *rho = G->rho_r + G->Hz, *prc = G->p_r + G->Hz;
#pragma omp parallel for collapse(2) schedule(static)
    for (int k = 0; k < G->Nz; ++k)
        for (int j = 0; j < G->Ny; ++j)
            for (int i = 0; i < G->Nx; ++i) {
                size_t n = IDX(G, i, j, k);
                double th = rtheta[n] / rho[k];
                double qt = rq[n] / rho[k];
                double pr = prc[k];
                theta[n] = th;
                qe[n] = qt;
                double qv = qt, ql = 0.0, Tn;
                if (th == 0.0) {
                    Tn = sa_T(G, S, th, qt, 0.0, pr);
                } else {
                    double T1 = sa_T(G, S, th, qt, 0.0, pr);
                    double rho1 = pr / (sa_Rm(G, qt, 0.0) * T1);
                    double qs1 = sa_psat(G, S, T1) / (rho1 * G->Rv * T1);
                    if (qt <= qs1) {
                        Tn = T1;
                    } else {
                        double qv1, ql1;
                        sa_adjust(G, S, T1, qt, pr, &qv1, &ql1);
                        double dT = (S->Ll * ql1 + 0.0) / sa_cpm(G, S, qv1, ql1);
                        double x1 = T1, x2 = T1 + fmax(0.01, dT / 2.0);
                        double r1 = sa_residual(G, S, x1, th, qt, pr), r2 = sa_residual(G, S, x2, th, qt, pr);
                        int it = 0;
                        while (fabs(r2) > fmax(S->abstol, 0.0 * fabs(x2)) && it < S->maxiter) {
                            double s = (x2 - x1) / (r2 - r1);
                            int valid = isfinite(s);
                            s = valid ? s : 0.0;
                            x1 = x2; r1 = r2;
                            x2 -= r2 * s;
                            r2 = sa_residual(G, S, x2, th, qt, pr);
                            r2 = valid ? r2 : 0.0;
                            ++it;
                        }
                        sa_adjust(G, S, x2, qt, pr, &qv, &ql);
                        Tn = sa_T(G, S, th, qv, ql, pr);
                    }
                }
                qv_out[n] = qv;
                ql_out[n] = ql;
                T[n] = Tn;
            }
}

/* z-momentum tendency with the moist mixture gas constant R_m = q_d Rd + q_v Rv, q_d = 1 - q_v - q_l
 * (anelastic_buoyancy.jl:36-72 with grid_moisture_fractions of SaturationAdjustment, saturation_adjustment.jl:127-131) */
static inline double buoyancy_ccc_moist(const og_grid *G, const double *T, const double *qv, const double *ql, int i, int j, int k)
{
    size_t n = IDX(G, i, j, k);
    double rho_r = G->rho_r[k + G->Hz], Tr = G->T_r[k + G->Hz];
    double Rmr = (1.0 - (0.0 + 0.0 + 0.0)) * G->Rd + 0.0 * G->Rv;
    double Rm = (1.0 - (qv[n] + ql[n] + 0.0)) * G->Rd + qv[n] * G->Rv;
    double rhop = rho_r * (Rmr * Tr / (Rm * T[n]) - 1.0);
    return -G->g * rhop;
}

void og_w_tendency_moist(const og_grid *G, double *Gw, const double *ru, const double *rv, const double *rw,
                         const double *w, const double *T, const double *qv, const double *ql)
{
#pragma omp parallel for collapse(2) schedule(static)
    for (int k = 1; k < G->Nz; ++k)
        for (int j = 0; j < G->Ny; ++j)
            for (int i = 0; i < G->Nx; ++i) {
                double Vinv = 1.0 / (G->dx * G->dy * G->dzf[k + G->Hz]);
                double a = 0.0, b = 0.0, c = 0.0;
                if (G->tx != FLAT) a = F_Uw(G, ru, w, i + 1, j, k) - F_Uw(G, ru, w, i, j, k);
                if (G->ty != FLAT) b = F_Vw(G, rv, w, i, j + 1, k) - F_Vw(G, rv, w, i, j, k);
                c = F_Ww(G, rw, w, i, j, k) - F_Ww(G, rw, w, i, j, k - 1);
                double bf = 0.5 * (buoyancy_ccc_moist(G, T, qv, ql, i, j, k - 1) + buoyancy_ccc_moist(G, T, qv, ql, i, j, k));
                Gw[IDX(G, i, j, k)] = -(Vinv * (a + b + c)) + bf;
            }
}
