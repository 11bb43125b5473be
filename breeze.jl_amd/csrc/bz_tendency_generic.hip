// bz_tendency_generic.hip — advection tendencies for WENO(order = 7) and WENO(order = 9), the order the reference's examples use
// (/root/reference/examples/dry_thermal_bubble.jl:15-25, bomex.jl:204, splitting_supercell.jl:279).
//
// Not a tuned path: one thread per cell, every face flux evaluated by the two cells that share it, stencils read straight from
// global memory (L1 / L2 resident).  The headline path (WENO-5) is bz_tendency5_kernels.h; this file exists so that a model built
// with the examples' scheme runs on the device with the reference's semantics:
//   * reconstruction tables of order 2r-1 (r = 4, 5) from tools/gen_weno_tables.py (exact rational derivation; they reproduce the
//     order-5 table of bz_weno.h and the Balsara & Shu tables), WENO-Z weights with tau_7 = |b0 + 3 b1 - 3 b2 - b3|,
//     tau_9 = |b0 + 2 b1 - 6 b2 + 2 b3 + b4|, eps = 1e-8 — Oceananigans.Advection is not vendored: recalled, PARITY UNPINNED;
//   * buffer cascade next to the Bounded z walls: order 9 -> 7 -> 5 -> 3 -> 1 (the largest stencil that fits);
//   * the advecting mass flux is interpolated with Centered(order 2r-2) and the same cascade (8 -> 6 -> 4 -> 2).
// Fluxes and their order of operations are those of the order-5 kernels (and of the CPU restatement the tests compare with), i.e.
// /root/reference/src/Advection.jl:20-35 and src/AtmosphereModels/dynamics_kernel_functions.jl:54-130.
#include "bz_internal.h"
#include "bz_weno.h"
#include "bz_weno_tables.h"

// Smoothness indicators and candidate values from the first differences of the cells (tables BD / CD of tools/gen_weno_tables.py: the
// quadratic forms of the expanded tables after the exact substitution w_j = w_0 + sum d_i).  Both precisions use this form since
// round 3: it needs 14 instead of 20 multiply-adds per sub-stencil, and it keeps its digits — the expanded form multiplies integer
// coefficients up to 2.5e6 with squares of a 300 K field, which costs Float64 five digits and Float32 all of them.  The six quotients
// of a reconstruction (tau / (beta_s + eps), the normalisation) are reciprocal-multiplies (bz_recip: v_rcp + Newton steps in Float64,
// v_rcp alone in Float32): an IEEE division is ~14 instructions, and the kernels below are bound by instruction issue.
#define BZ_WENO_GENERIC(R)                                                                                       \
    __device__ __forceinline__ double bz_weno_r##R(const double *v)                                             \
    {                                                                                                           \
        double d[2 * R - 2], beta[R], p[R], tau = 0.0, num = 0.0, den = 0.0;                                    \
        _Pragma("unroll") for (int j = 0; j < 2 * R - 2; ++j) d[j] = v[j + 1] - v[j];                           \
        _Pragma("unroll") for (int s = 0; s < R; ++s) {                                                         \
            const double *w = d + (R - 1 - s);                                                                  \
            double b = 0.0, q = 0.0;                                                                            \
            _Pragma("unroll") for (int j = 0; j < R - 1; ++j) {                                                 \
                double in = BZW_BD##R[s][j][j] * w[j];                                                          \
                _Pragma("unroll") for (int l = j + 1; l < R - 1; ++l) in += BZW_BD##R[s][j][l] * w[l];          \
                b = (j == 0) ? w[j] * in : b + w[j] * in;                                                       \
                q = (j == 0) ? BZW_CD##R[s][j] * w[j] : q + BZW_CD##R[s][j] * w[j];                             \
            }                                                                                                   \
            beta[s] = b; p[s] = q;                                                                              \
            tau = (s == 0) ? BZW_T##R[s] * b : tau + BZW_T##R[s] * b;                                           \
        }                                                                                                       \
        tau = fabs(tau);                                                                                        \
        if (BZ_WENO_FT2 == 2) {      /* the FT2 hypothesis (bz_weno.h): quotients, alpha and the normalised weights in Float32 */ \
            float af[R], sum = 0.0f;                                                                            \
            _Pragma("unroll") for (int s = 0; s < R; ++s) {                                                     \
                const float rr = (float)tau / ((float)beta[s] + 1e-8f);                                         \
                af[s] = (float)BZW_D##R[s] * (1.0f + rr * rr);                                                  \
                sum += af[s];                                                                                   \
            }                                                                                                   \
            double acc = 0.0;                                                                                   \
            /* full candidates (v + increment): Float32 weights do not sum to one exactly, and v (1 - sum w) is not small for theta ~ 300 K */ \
            _Pragma("unroll") for (int s = 0; s < R; ++s) acc += (double)(af[s] / sum) * (v[R - 1] + p[s]);     \
            return acc;                                                                                         \
        }                                                                                                       \
        _Pragma("unroll") for (int s = 0; s < R; ++s) {                                                         \
            const double rr = (BZ_WENO_FT2 == 1) ? bz_newton_div(tau, beta[s] + BZ_WENO_EPS) : tau * bz_recip<1>(beta[s] + BZ_WENO_EPS); \
            const double a = BZW_D##R[s] * (1.0 + rr * rr);                                                     \
            num = (s == 0) ? a * p[s] : num + a * p[s];                                                         \
            den = (s == 0) ? a : den + a;                                                                       \
        }                                                                                                       \
        return v[R - 1] + num * bz_recip<2>(den);                                                               \
    }
BZ_WENO_GENERIC(4)
BZ_WENO_GENERIC(5)

// timing experiments only (results are then WRONG): BZ_KOG bit 0: every z-stencil load reads the target level (what a register ring would
// leave of the loads), bit 1: the same for y stencils, bit 2: the order-7 / 9 reconstruction arithmetic is stubbed
#ifndef BZ_GENERIC_CLUSTER
#define BZ_GENERIC_CLUSTER 1
#endif
#ifndef BZ_KOG
#define BZ_KOG 0
#endif
#define BZ_KOG_OFF(off, s) ((((BZ_KOG & 1) && (s) > 8192) || ((BZ_KOG & 2) && (s) > 1 && (s) <= 8192)) ? 0 : (off))

// largest buffer B <= R usable at index idx of the Bounded z direction (face target: B <= idx <= N-B; centre: B-1 <= idx <= N-B)
template <int R>
__device__ __forceinline__ int buf_face(int idx, int N)
{
#pragma unroll
    for (int B = R; B >= 2; --B)
        if (idx >= B && idx <= N - B) return B;
    return 1;
}
template <int R>
__device__ __forceinline__ int buf_center(int idx, int N)
{
#pragma unroll
    for (int B = R; B >= 2; --B)
        if (idx >= B - 1 && idx <= N - B) return B;
    return 1;
}

// Bounded y ((Periodic, Bounded, Bounded)): the buffer of row / y-face jj — wave-uniform, like the level's in z; R on a periodic y
template <int R>
__device__ __forceinline__ int bx_face_g(const DevGrid &g, int ii) { return g.bounded_x ? buf_face<R>(ii, g.Nx) : R; }      // Bounded x: by lane
template <int R>
__device__ __forceinline__ int bx_center_g(const DevGrid &g, int ii) { return g.bounded_x ? buf_center<R>(ii, g.Nx) : R; }
template <int R>
__device__ __forceinline__ int by_face_g(const DevGrid &g, int jj) { return g.bounded_y ? buf_face<R>(jj, g.Ny) : R; }
template <int R>
__device__ __forceinline__ int by_center_g(const DevGrid &g, int jj) { return g.bounded_y ? buf_center<R>(jj, g.Ny) : R; }

// upwind-biased value at FACE idx of centred data (p at cell idx) / at CENTRE idx of face data (p at face idx), stride s, buffer B.
// The 2 B values straddling the target are loaded once; the biased stencil is q[0 .. 2B-2] or its mirror image q[2B-1 .. 1], picked with
// lane-mask selects (bz_sel: the VOP3 select; the ternary form compiled to two loads per value and VCC selects).
// (B is a template parameter of the wide branches: a run-time B in q[2 B - 1 - j] is a dynamic register-array index, i.e. scratch memory)
template <int B>
__device__ __forceinline__ void load_wide_g(double (&q)[2 * B], const double *__restrict__ p, long long s, int first)
{
#pragma unroll
    for (int j = 0; j < 2 * B; ++j) q[j] = p[BZ_KOG_OFF(j + first, s) * s];
}
template <int B>
__device__ __forceinline__ double weno_wide_g(const double (&q)[2 * B], bool left)
{
    const unsigned long long m = bz_lanes(left);
    double v[2 * B - 1];
#pragma unroll
    for (int j = 0; j < 2 * B - 1; ++j) v[j] = bz_sel(m, q[j], q[2 * B - 1 - j]);
    if (BZ_KOG & 4) return q[B - 1] + 0.25 * (q[B] - q[B - 2]);      // reconstruction arithmetic stubbed
    if constexpr (B == 5) return bz_weno_r5(v);
    else return bz_weno_r4(v);
}
template <int B>
__device__ __forceinline__ double biased_wide_g(const double *__restrict__ p, long long s, bool left, int first)
{
    double q[2 * B];
    load_wide_g<B>(q, p, s, first);
    return weno_wide_g<B>(q, left);
}
__device__ __forceinline__ double biased_face_g(const double *__restrict__ p, long long s, bool left, int B)
{
    if (B == 5) return biased_wide_g<5>(p, s, left, -5);
    if (B == 4) return biased_wide_g<4>(p, s, left, -4);
    if (B == 3) return bz_up5(p[-3 * s], p[-2 * s], p[-s], p[0], p[s], p[2 * s], left);
    if (B == 2) return bz_up3(p[-2 * s], p[-s], p[0], p[s], left);
    return bz_sel(bz_lanes(left), p[-s], p[0]);
}
__device__ __forceinline__ double biased_center_g(const double *__restrict__ p, long long s, bool left, int B)
{
    if (B == 5) return biased_wide_g<5>(p, s, left, -4);
    if (B == 4) return biased_wide_g<4>(p, s, left, -3);
    if (B == 3) return bz_up5(p[-2 * s], p[-s], p[0], p[s], p[2 * s], p[3 * s], left);
    if (B == 2) return bz_up3(p[-s], p[0], p[s], p[2 * s], left);
    return bz_sel(bz_lanes(left), p[0], p[s]);
}

// Centered(order 2 (B - 1)) (order 2 for B <= 2) of q(m) = A(m) M(m) along stride s; `first` = offset (in cells) of the first of the
// 2 h values (h = max(B - 1, 1)): -h for a face target from centres (centres idx-h .. idx+h-1), -(h-1) for a centre target from faces.
// A == nullptr: constant factor a0.
// (H = number of values either side is a template parameter of the implementation: with a run-time h the unrolled q[h - 1 - d] is a
// dynamic register-array index, and the table `c` a run-time pointer)
template <int H>
__device__ __forceinline__ double symm_h(const double *__restrict__ M, long long n, long long s, int first, const ColPtr A, int k, double a0)
{
    double q[2 * H];
#pragma unroll
    for (int m = 0; m < 2 * H; ++m) q[m] = (A.p ? A[k + first + m] : a0) * M[n + (long long)BZ_KOG_OFF(first + m, s) * s];
    if constexpr (H == 1) return bz_symm2(q[0], q[1]);
    else if constexpr (H == 2) return bz_symm4(q[0], q[1], q[2], q[3]);
    else {
        double acc = (H == 4 ? BZW_S8[0] : BZW_S6[0]) * (q[H - 1] + q[H]);
#pragma unroll
        for (int d = 1; d < H; ++d) acc += (H == 4 ? BZW_S8[d] : BZW_S6[d]) * (q[H - 1 - d] + q[H + d]);
        return acc;
    }
}
__device__ __forceinline__ double symm_g(const double *__restrict__ M, long long n, long long s, int B, int first, const ColPtr A, int k,
                                         double a0)
{
    if (B == 5) return symm_h<4>(M, n, s, first, A, k, a0);
    if (B == 4) return symm_h<3>(M, n, s, first, A, k, a0);
    if (B == 3) return symm_h<2>(M, n, s, first, A, k, a0);
    return symm_h<1>(M, n, s, first, A, k, a0);
}

// ---- two-pass evaluation ---------------------------------------------------------------------------------------------------------
// A face flux belongs to two cells.  PASS 0 lets both of them evaluate it (one launch, 6 reconstructions per cell and field); the
// shipped path is PASS 1 + PASS 2: every thread evaluates the three fluxes of its own index once and stores them in parent-shaped
// scratch arrays (ctx->d_gflux), a second launch differences them.  Order-9 reconstructions cost ~700 instructions each, the scratch
// traffic (3 words written, 6 read per cell) is noise beside that.  Bit-identical to PASS 0: the same expressions, evaluated once.
struct FluxBuf {
    double *x, *y, *z;
    // optional SSP-RK3 epilogue of the divergence pass (fused-RK tier of the whole-step seam): instead of the tendency G the kernel
    // stores (1 - alpha) u0 + alpha (uold + dt G); uold = the prognostic field the tendency belongs to (E.mode 0: store G)
    RKEpilogue E;
    const double *uold;
    // fused-RK moisture launch only: the moisture scan's word (bz_step.hip: bzi_scan_moisture); where rho q is identically zero its
    // update is 0 -> 0 and both passes of the launch return at once (nothing is read or written; U0 of rho q is then never read either)
    const int *skip_if_dry;
};
__device__ __forceinline__ double rk_out(const FluxBuf &F, double G, long long n)
{
    if (F.E.mode == 0) return G;
    return bz_rk_apply(F.E.mode, F.E.dt, F.E.alpha, F.E.oma, F.E.u0, F.E.u0_out, G, F.uold[n], n);
}

// PASS 1 covers the interior (x is Periodic and y wraps unless the context is a y-slab or Flat: the flux at face N is the flux at face 0,
// bit for bit, because halos are exact periodic images; the Bounded z direction has zero mass flux through its wall faces) plus, on
// y-slabs, one row either side (the fluxes at the slab edges come from exchanged halo rows)
// Block order (round 5).  Consecutive workgroup ids go round-robin to the 8 XCDs, each with its own L2: in launch order (x fastest, then y)
// the rows j - 5 .. j + 5 an order-9 y stencil reads belonged to workgroups of OTHER XCDs, and the marching kernel's grid — 8 tiles wide at
// 512 cells — gave every XCD a tile COLUMN, its x neighbours behind another L2.  Here XCD c owns the band [c gy/8, (c+1) gy/8) of the grid's
// y extent and walks it x fastest, then y, then z (bz_fused.hip: bz_stream_block; bz_compressible.hip: k_ac_column_forward).  Grids whose y
// extent is not a multiple of 8 keep launch order.  The indices are wave-uniform (column tables stay scalar loads).
#ifndef GEN_XCD
#define GEN_XCD 1
#endif
#ifndef GEN_XCD_MARCH
#define GEN_XCD_MARCH 0      // the marching kernels measured equal (momentum) or 1 % slower (theta) in band order: they keep launch order
#endif
__device__ __forceinline__ void generic_block(int &bx, int &by, int &bz, bool bands = GEN_XCD != 0)
{
    bx = blockIdx.x; by = blockIdx.y; bz = blockIdx.z;
    const unsigned gx = gridDim.x, gy = gridDim.y;
    if (bands && (gy & 7u) == 0) {
        const unsigned w = bx + gx * (by + gy * bz), c = w & 7u, rows = gy >> 3;
        unsigned r = w >> 3;
        bx = (int)(r % gx); r /= gx;
        by = (int)(c * rows + r % rows);
        bz = (int)(r / rows);
    }
    bx = __builtin_amdgcn_readfirstlane(bx); by = __builtin_amdgcn_readfirstlane(by); bz = __builtin_amdgcn_readfirstlane(bz);
}
struct GenericWrap { long long xp, xm, yp, ym; bool top, xwall; };      // xwall: the x-face above this cell is a wall (Bounded x): zero flux
template <int PASS>
__device__ __forceinline__ bool generic_index(const DevGrid &g, int &i, int &j, int &k, int k0, GenericWrap &W)
{
    const bool ext_y = !g.wrap_y && !g.flat_y;
    int bx, by, bz;
    generic_block(bx, by, bz);
    i = bx * 256 + threadIdx.x;
    j = by - ((PASS == 1 && ext_y) ? 1 : 0);
    k = (PASS == 1 ? 0 : k0) + bz;
    const long long sy = g.Sx;
    W.xp = (i + 1 < g.Nx || g.bounded_x) ? 1 : 1 - g.Nx;
    W.xm = (i > 0 || g.bounded_x) ? -1 : g.Nx - 1;
    W.xwall = g.bounded_x && i == g.Nx - 1;
    W.yp = (j + 1 < g.Ny || ext_y) ? sy : sy * (1 - g.Ny);
    W.ym = (j > 0 || ext_y) ? -sy : sy * (g.Ny - 1);
    W.top = (k + 1 >= g.Nz);
    return i < g.Nx;
}
#define IN(a, lo, hi) ((a) >= (lo) && (a) <= (hi))

// ---- scalars: G = -div_rhoUc(c) -----------------------------------------------------------------------------------------------------
template <int R, int PASS>
__global__ __launch_bounds__(256) void k_scalar_tendency_g(DevGrid g, double *__restrict__ Gc, const double *__restrict__ u,
                                                           const double *__restrict__ v, const double *__restrict__ w,
                                                           const double *__restrict__ c, FluxBuf F)
{
    if (F.skip_if_dry && __builtin_amdgcn_readfirstlane(*F.skip_if_dry) == 1) return;
    int i, j, k;
    GenericWrap W;
    if (!generic_index<PASS>(g, i, j, k, 0, W)) return;
    const bool ext_y = !g.wrap_y && !g.flat_y;
    const long long sy = g.Sx, sz = g.Sxy, n = g.idx(i, j, k);
    auto fx = [&](long long m, int ii) { const double ut = u[m]; return g.rho[k] * ((g.Ax[k] * ut) * biased_face_g(c + m, 1, ut > 0.0, bx_face_g<R>(g, ii))); };
    auto fy = [&](long long m, int jj) { const double vt = v[m]; return g.rho[k] * ((g.Ay[k] * vt) * biased_face_g(c + m, sy, vt > 0.0, by_face_g<R>(g, jj))); };
    auto fz = [&](long long m, int kf) {
        const double wt = w[m];
        return g.rho_f[kf] * ((g.Az * wt) * biased_face_g(c + m, sz, wt > 0.0, buf_face<R>(kf, g.Nz)));
    };
    if (PASS == 1) {
        const bool cj = IN(j, 0, g.Ny - 1);
#if BZ_GENERIC_CLUSTER
        // interior rows and levels of a grid without walls in x / y (the block is one row of one level: wave-uniform): the thirty stencil
        // values and the three velocities are requested before the first reconstruction starts — one memory latency per cell instead of
        // three (the calls below issue their loads behind the row / level conditions, where the compiler cannot hoist them)
        if (cj && !g.flat_y && !g.bounded_x && !g.bounded_y && k >= R && k <= g.Nz - R) {
            double qx[2 * R], qy[2 * R], qz[2 * R];
            load_wide_g<R>(qx, c + n, 1, -R);
            load_wide_g<R>(qy, c + n, sy, -R);
            load_wide_g<R>(qz, c + n, sz, -R);
            const double ut = u[n], vt = v[n], wt = w[n];
            F.x[n] = g.rho[k] * ((g.Ax[k] * ut) * weno_wide_g<R>(qx, ut > 0.0));
            F.y[n] = g.rho[k] * ((g.Ay[k] * vt) * weno_wide_g<R>(qy, vt > 0.0));
            F.z[n] = g.rho_f[k] * ((g.Az * wt) * weno_wide_g<R>(qz, wt > 0.0));
            return;
        }
#endif
        if (cj) F.x[n] = fx(n, i);
        if (!g.flat_y && IN(j, 0, g.Ny - (ext_y ? 0 : 1))) F.y[n] = fy(n, j);
        if (cj) F.z[n] = fz(n, k);
        return;
    }
    double dx, dy, dz;
    if (PASS == 2) { dx = (W.xwall ? 0.0 : F.x[n + W.xp]) - F.x[n]; dy = g.flat_y ? 0.0 : F.y[n + W.yp] - F.y[n]; dz = (W.top ? 0.0 : F.z[n + sz]) - F.z[n]; }
    else { dx = fx(n + 1, i + 1) - fx(n, i); dy = g.flat_y ? 0.0 : fy(n + sy, j + 1) - fy(n, j); dz = fz(n + sz, k + 1) - fz(n, k); }      // a Flat y has no faces
    Gc[n] = rk_out(F, -(g.Vinv_c[k] * (dx + dy + dz)), n);
}

// ---- momentum -------------------------------------------------------------------------------------------------------------------
template <int R, int PASS>
__global__ __launch_bounds__(256) void k_u_tendency_g(DevGrid g, double *__restrict__ Gu, const double *__restrict__ ru,
                                                      const double *__restrict__ rv, const double *__restrict__ rw,
                                                      const double *__restrict__ u, FluxBuf F)
{
    int i, j, k;
    GenericWrap W;
    if (!generic_index<PASS>(g, i, j, k, 0, W)) return;
    const bool ext_y = !g.wrap_y && !g.flat_y;
    const long long sy = g.Sx, sz = g.Sxy, n = g.idx(i, j, k);
    const ColPtr none(nullptr);
    const int Bfx = bx_face_g<R>(g, i), hfx = Bfx > 2 ? Bfx - 1 : 1;      // Centered in x to the x-face of this lane
    auto FUu = [&](long long m, int ic) {      // at centre ic: advecting flux from faces
        const int B = bx_center_g<R>(g, ic), h = B > 2 ? B - 1 : 1;
        const double ut = symm_g(ru, m, 1, B, -(h - 1), none, 0, g.Ax[k]);
        return ut * biased_center_g(u + m, 1, ut > 0.0, B);
    };
    auto FVu = [&](long long m, int jj) {      // at (f, f, c): Centered in x of Ay rho_v to x-face
        const double vt = symm_g(rv, m, 1, Bfx, -hfx, none, 0, g.Ay[k]);
        return vt * biased_face_g(u + m, sy, vt > 0.0, by_face_g<R>(g, jj));
    };
    auto FWu = [&](long long m, int kf) {
        const double wt = symm_g(rw, m, 1, Bfx, -hfx, none, 0, g.Az);
        return wt * biased_face_g(u + m, sz, wt > 0.0, buf_face<R>(kf, g.Nz));
    };
    if (PASS == 1) {
        const bool cj = IN(j, 0, g.Ny - 1);
        if (cj) F.x[n] = FUu(n, i);
        if (!g.flat_y && IN(j, 0, g.Ny - (ext_y ? 0 : 1))) F.y[n] = FVu(n, j);
        if (cj) F.z[n] = FWu(n, k);
        return;
    }
    if (g.bounded_x && i == 0) return;      // the wall face is never updated
    double a, b, c;
    if (PASS == 2) { a = F.x[n] - F.x[n + W.xm]; b = g.flat_y ? 0.0 : F.y[n + W.yp] - F.y[n]; c = (W.top ? 0.0 : F.z[n + sz]) - F.z[n]; }
    else { a = FUu(n, i) - FUu(n - 1, i - 1); b = g.flat_y ? 0.0 : FVu(n + sy, j + 1) - FVu(n, j); c = FWu(n + sz, k + 1) - FWu(n, k); }
    Gu[n] = rk_out(F, -(g.Vinv_c[k] * (a + b + c)), n);
}

template <int R, int PASS>
__global__ __launch_bounds__(256) void k_v_tendency_g(DevGrid g, double *__restrict__ Gv, const double *__restrict__ ru,
                                                      const double *__restrict__ rv, const double *__restrict__ rw,
                                                      const double *__restrict__ v, FluxBuf F)
{
    int i, j, k;
    GenericWrap W;
    if (!generic_index<PASS>(g, i, j, k, 0, W)) return;
    const bool ext_y = !g.wrap_y && !g.flat_y;
    const long long sy = g.Sx, sz = g.Sxy, n = g.idx(i, j, k);
    const ColPtr none(nullptr);
    // Centered in y to the y-face of this row: order 2 (B - 1) with the buffer that fits (B = R on a periodic y), like the z interpolations of k_w_tendency_g
    const int Bfy = by_face_g<R>(g, j), hfy = Bfy > 2 ? Bfy - 1 : 1;
    auto FUv = [&](long long m, int ii) {
        const double ut = g.flat_y ? g.Ax[k] * ru[m] : symm_g(ru, m, sy, Bfy, -hfy, none, 0, g.Ax[k]);      // Iy of a Flat direction: identity
        return ut * biased_face_g(v + m, 1, ut > 0.0, bx_face_g<R>(g, ii));
    };
    auto FVv = [&](long long m, int jc) {      // at centre jc
        const int B = by_center_g<R>(g, jc), h = B > 2 ? B - 1 : 1;
        const double vt = symm_g(rv, m, sy, B, -(h - 1), none, 0, g.Ay[k]);
        return vt * biased_center_g(v + m, sy, vt > 0.0, B);
    };
    auto FWv = [&](long long m, int kf) {
        const double wt = g.flat_y ? g.Az * rw[m] : symm_g(rw, m, sy, Bfy, -hfy, none, 0, g.Az);
        return wt * biased_face_g(v + m, sz, wt > 0.0, buf_face<R>(kf, g.Nz));
    };
    if (PASS == 1) {
        const bool cj = IN(j, 0, g.Ny - 1);
        if (cj) F.x[n] = FUv(n, i);
        if (!g.flat_y && IN(j, ext_y ? -1 : 0, g.Ny - 1)) F.y[n] = FVv(n, j);
        if (cj) F.z[n] = FWv(n, k);
        return;
    }
    if (g.bounded_y && j == 0) return;      // the wall face is never updated
    double a, b, c;
    if (PASS == 2) { a = (W.xwall ? 0.0 : F.x[n + W.xp]) - F.x[n]; b = g.flat_y ? 0.0 : F.y[n] - F.y[n + W.ym]; c = (W.top ? 0.0 : F.z[n + sz]) - F.z[n]; }
    else { a = FUv(n + 1, i + 1) - FUv(n, i); b = g.flat_y ? 0.0 : FVv(n, j) - FVv(n - sy, j - 1); c = FWv(n + sz, k + 1) - FWv(n, k); }
    Gv[n] = rk_out(F, -(g.Vinv_c[k] * (a + b + c)), n);
}

// faces k = 1 .. Nz-1; + Iz(buoyancy) unless !BUOY (slow tendency of the split-explicit compressible model)
template <int R, int PASS, bool BUOY = true>
__global__ __launch_bounds__(256) void k_w_tendency_g(DevGrid g, double *__restrict__ Gw, const double *__restrict__ ru,
                                                      const double *__restrict__ rv, const double *__restrict__ rw,
                                                      const double *__restrict__ w, const double *__restrict__ T,
                                                      const double *__restrict__ qv, FluxBuf F)
{
    int i, j, k;
    GenericWrap W;
    if (!generic_index<PASS>(g, i, j, k, 1, W)) return;      // PASS 1: k = 0 .. Nz-1 (the centre fluxes are needed from 0 on)
    const bool ext_y = !g.wrap_y && !g.flat_y;
    const long long sy = g.Sx, sz = g.Sxy, n = g.idx(i, j, k);
    const ColPtr none(nullptr);
    auto FUw = [&](long long m, int ii) {      // Centered in z of Ax(k) rho_u to z-face k
        const int Bf = buf_face<R>(k, g.Nz);
        const int h = Bf > 2 ? Bf - 1 : 1;
        const double ut = symm_g(ru, m, sz, Bf, -h, g.Ax, k, 0.0);
        return ut * biased_face_g(w + m, 1, ut > 0.0, bx_face_g<R>(g, ii));
    };
    auto FVw = [&](long long m, int jj) {
        const int Bf = buf_face<R>(k, g.Nz);
        const int h = Bf > 2 ? Bf - 1 : 1;
        const double vt = symm_g(rv, m, sz, Bf, -h, g.Ay, k, 0.0);
        return vt * biased_face_g(w + m, sy, vt > 0.0, by_face_g<R>(g, jj));
    };
    auto FWw = [&](long long m, int kc) {      // at centre kc
        const int B = buf_center<R>(kc, g.Nz);
        const int h = B > 2 ? B - 1 : 1;
        const double wt = symm_g(rw, m, sz, B, -(h - 1), none, 0, g.Az);
        return wt * biased_center_g(w + m, sz, wt > 0.0, B);
    };
    if (PASS == 1) {
        const bool cj = IN(j, 0, g.Ny - 1), ck = k >= 1;
        if (cj && ck) F.x[n] = FUw(n, i);
        if (!g.flat_y && ck && IN(j, 0, g.Ny - (ext_y ? 0 : 1))) F.y[n] = FVw(n, j);
        if (cj) F.z[n] = FWw(n, k);
        return;
    }
    double a, b, c;
    if (PASS == 2) { a = (W.xwall ? 0.0 : F.x[n + W.xp]) - F.x[n]; b = g.flat_y ? 0.0 : F.y[n + W.yp] - F.y[n]; c = F.z[n] - F.z[n - sz]; }
    else { a = FUw(n + 1, i + 1) - FUw(n, i); b = g.flat_y ? 0.0 : FVw(n + sy, j + 1) - FVw(n, j); c = FWw(n, k) - FWw(n - sz, k - 1); }
    const double adv = -(g.Vinv_f[k] * (a + b + c));
    if (BUOY) Gw[n] = rk_out(F, adv + 0.5 * (bz_buoyancy(g, T, qv, n - sz, k - 1) + bz_buoyancy(g, T, qv, n, k)), n);
    else Gw[n] = adv;
}

// scalar tendency with a 3-D carrier density, G = -div(Ix/y/z(rho) U c~) (src/Advection.jl:20-35 with the compressible model's density
// field), and, when Grho != nullptr, the dry-density tendency -div(momentum) (compressible_density_tendency.jl:52-55)
template <int R, int PASS>
__global__ __launch_bounds__(256) void k_scalar_tendency_rho3d_g(DevGrid g, double *__restrict__ Gc, double *__restrict__ Grho,
                                                                 const double *__restrict__ rho, const double *__restrict__ u,
                                                                 const double *__restrict__ v, const double *__restrict__ w,
                                                                 const double *__restrict__ c, const double *__restrict__ ru,
                                                                 const double *__restrict__ rv, const double *__restrict__ rw, FluxBuf F)
{
    int i, j, k;
    GenericWrap W;
    if (!generic_index<PASS>(g, i, j, k, 0, W)) return;
    const bool ext_y = !g.wrap_y && !g.flat_y;
    const long long sy = g.Sx, sz = g.Sxy, n = g.idx(i, j, k);
    auto fx = [&](long long m) { const double ut = u[m]; return ((rho[m] + rho[m - 1]) / 2.0) * ((g.Ax[k] * ut) * biased_face_g(c + m, 1, ut > 0.0, R)); };
    auto fy = [&](long long m) { const double vt = v[m]; return ((rho[m] + rho[m - sy]) / 2.0) * ((g.Ay[k] * vt) * biased_face_g(c + m, sy, vt > 0.0, R)); };
    auto fz = [&](long long m, int kf) {
        const double wt = w[m];
        return ((rho[m] + rho[m - sz]) / 2.0) * ((g.Az * wt) * biased_face_g(c + m, sz, wt > 0.0, buf_face<R>(kf, g.Nz)));
    };
    if (PASS == 1) {
        const bool cj = IN(j, 0, g.Ny - 1);
        if (cj) F.x[n] = fx(n);
        if (!g.flat_y && IN(j, 0, g.Ny - (ext_y ? 0 : 1))) F.y[n] = fy(n);
        if (cj) F.z[n] = fz(n, k);
        return;
    }
    double dx, dy, dz;
    if (PASS == 2) { dx = (W.xwall ? 0.0 : F.x[n + W.xp]) - F.x[n]; dy = g.flat_y ? 0.0 : F.y[n + W.yp] - F.y[n]; dz = (W.top ? 0.0 : F.z[n + sz]) - F.z[n]; }
    else { dx = fx(n + 1) - fx(n); dy = g.flat_y ? 0.0 : fy(n + sy) - fy(n); dz = fz(n + sz, k + 1) - fz(n, k); }
    Gc[n] = -(g.Vinv_c[k] * (dx + dy + dz));
    if (Grho) {
        const double Ax = g.Ax[k], Ay = g.Ay[k];
        const double a = Ax * ru[n + 1] - Ax * ru[n];
        const double b = g.flat_y ? 0.0 : Ay * rv[n + sy] - Ay * rv[n];
        const double cc = g.Az * rw[n + sz] - g.Az * rw[n];
        Grho[n] = -(g.Vinv_c[k] * (a + b + cc));
    }
}
#undef IN

// ---- single pass, marching in z (round 4) ----------------------------------------------------------------------------------------
// The two-pass path above writes three flux arrays per field and reads six values back in a second launch: measured on the reference's
// benchmark case (CBL 512 x 512 x 256, Float32, order 9; tools/gpu_ko_generic.sh) the flux pass with its reconstruction arithmetic AND
// its stencil loads removed still takes 60 % of its time, and the divergence pass another 0.24 - 0.44 ms per field: the skeleton, not
// the arithmetic, is the larger half.  Here a workgroup of 64 x MTY threads owns 64 x MTY columns and walks 64 levels: every thread
// evaluates the three fluxes of its own index ONCE per level with the functions above (same expressions => same bits as the two-pass
// path), and the neighbours' fluxes arrive through
//   x: a lane shuffle; the flux beyond the tile edge is evaluated once per workgroup by all lanes for the 64 levels of the march
//      (lane l: level kbeg + l) and read back with a second shuffle — 1 / 64 extra reconstructions;
//   y: an LDS row exchange of MTY levels at a time; the one row outside the tile is evaluated by a different wave at every level of
//      the group (wave l takes level l), so every wave does MTY * 3 + 1 reconstructions per group — a fixed "top row" wave would do
//      4 per level where the others do 3, and the barrier would wait for it (measured 9 % of k6_w);
//   z: the flux of the lower face / centre rides a register from the previous level.
// No flux arrays, one launch per field, the SSP-RK3 epilogue as in PASS 2.  Periodic x, wrapped y (single GPU) with Nx a multiple of
// 64 and Ny of MTY; everything else (walls, Flat, y-slabs) keeps the two-pass path.  BZ_NO_GENERIC_MARCH=1 restores it everywhere.
#define MTY 4
#ifndef MARCH_W
#define MARCH_W 0
#endif
#ifndef MARCH_WAVES
#define MARCH_WAVES 4
#endif
template <int R, int KIND>      // KIND 0: scalar c advected by (u, v, w) [passed in the ru, rv, rw slots]; 1, 2, 3: u, v, w momentum
struct MarchFlux {
    const DevGrid &g;
    const double *__restrict__ ru, *__restrict__ rv, *__restrict__ rw, *__restrict__ a;
    const long long sy, sz;
    const ColPtr none;
    __device__ __forceinline__ MarchFlux(const DevGrid &g_, const double *ru_, const double *rv_, const double *rw_, const double *a_)
        : g(g_), ru(ru_), rv(rv_), rw(rw_), a(a_), sy(g_.Sx), sz(g_.Sxy), none(nullptr) {}
    // flux through the x-face (KIND 0, 2, 3) / at the x-centre (KIND 1) of index m, level k
    __device__ __forceinline__ double X(long long m, int k) const
    {
        if (KIND == 0) { const double ut = ru[m]; return g.rho[k] * ((g.Ax[k] * ut) * biased_face_g(a + m, 1, ut > 0.0, R)); }
        if (KIND == 1) { const double ut = symm_g(ru, m, 1, R, -(R - 2), none, 0, g.Ax[k]); return ut * biased_center_g(a + m, 1, ut > 0.0, R); }
        if (KIND == 2) { const double ut = symm_g(ru, m, sy, R, -(R - 1), none, 0, g.Ax[k]); return ut * biased_face_g(a + m, 1, ut > 0.0, R); }
        const int Bf = buf_face<R>(k, g.Nz), h = Bf > 2 ? Bf - 1 : 1;
        const double ut = symm_g(ru, m, sz, Bf, -h, g.Ax, k, 0.0);
        return ut * biased_face_g(a + m, 1, ut > 0.0, R);
    }
    // y-face (KIND 0, 1, 3) / y-centre (KIND 2)
    __device__ __forceinline__ double Y(long long m, int k) const
    {
        if (KIND == 0) { const double vt = rv[m]; return g.rho[k] * ((g.Ay[k] * vt) * biased_face_g(a + m, sy, vt > 0.0, R)); }
        if (KIND == 1) { const double vt = symm_g(rv, m, 1, R, -(R - 1), none, 0, g.Ay[k]); return vt * biased_face_g(a + m, sy, vt > 0.0, R); }
        if (KIND == 2) { const double vt = symm_g(rv, m, sy, R, -(R - 2), none, 0, g.Ay[k]); return vt * biased_center_g(a + m, sy, vt > 0.0, R); }
        const int Bf = buf_face<R>(k, g.Nz), h = Bf > 2 ? Bf - 1 : 1;
        const double vt = symm_g(rv, m, sz, Bf, -h, g.Ay, k, 0.0);
        return vt * biased_face_g(a + m, sy, vt > 0.0, R);
    }
    // z-face kf, in two halves (KIND 0, 1, 2): the advecting factor at the face (m = index of the cell above it) and the flux from it and
    // the reconstructed value — Z(m, kf) == Zfin(Zmass(m), biased_face_g(a + m, sz, Zmass(m) > 0, buf_face(kf)), kf), which the marching
    // kernel evaluates with the z stencil in a register ring
    __device__ __forceinline__ double Zmass(long long m) const
    {
        if (KIND == 0) return rw[m];
        if (KIND == 1) return symm_g(rw, m, 1, R, -(R - 1), none, 0, g.Az);
        return symm_g(rw, m, sy, R, -(R - 1), none, 0, g.Az);
    }
    __device__ __forceinline__ double Zfin(double wt, double rec, int kf) const
    {
        if (KIND == 0) return g.rho_f[kf] * ((g.Az * wt) * rec);
        return wt * rec;
    }
    // z-face kf (KIND 0, 1, 2) / z-centre kf (KIND 3)
    __device__ __forceinline__ double Z(long long m, int kf) const
    {
        if (KIND == 0) { const double wt = rw[m]; return g.rho_f[kf] * ((g.Az * wt) * biased_face_g(a + m, sz, wt > 0.0, buf_face<R>(kf, g.Nz))); }
        if (KIND == 1) { const double wt = symm_g(rw, m, 1, R, -(R - 1), none, 0, g.Az); return wt * biased_face_g(a + m, sz, wt > 0.0, buf_face<R>(kf, g.Nz)); }
        if (KIND == 2) { const double wt = symm_g(rw, m, sy, R, -(R - 1), none, 0, g.Az); return wt * biased_face_g(a + m, sz, wt > 0.0, buf_face<R>(kf, g.Nz)); }
        const int B = buf_center<R>(kf, g.Nz), h = B > 2 ? B - 1 : 1;
        const double wt = symm_g(rw, m, sz, B, -(h - 1), none, 0, g.Az);
        return wt * biased_center_g(a + m, sz, wt > 0.0, B);
    }
};

// upwind-biased value at a z face from the ring zr[0 .. 2R-1] = levels kf-R .. kf+R-1 around face kf (biased_face_g(p, sz, left, B) with
// p[(j - R) sz] = zr[j]): buffer B of the wall cascade picks the central 2 B entries
template <int R>
__device__ __forceinline__ double ring_face_g(const double (&zr)[2 * R], bool left, int B)
{
    if (B == R) return weno_wide_g<R>(zr, left);
    if (R == 5 && B == 4) {
        double q[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) q[j] = zr[j + (R - 4)];
        return weno_wide_g<4>(q, left);
    }
    if (B == 3) return bz_up5(zr[R - 3], zr[R - 2], zr[R - 1], zr[R], zr[R + 1], zr[R + 2], left);
    if (B == 2) return bz_up3(zr[R - 2], zr[R - 1], zr[R], zr[R + 1], left);
    return bz_sel(bz_lanes(left), zr[R - 1], zr[R]);
}

// a - b that is never contracted with a product feeding it: the two-pass path rounds every flux when it stores it, so the differences
// below must see rounded products too (hipcc's default -ffp-contract=fast-honor-pragmas would turn nb - ut * r into an fma)
__device__ __forceinline__ double bz_sub_rounded(double a, double b)
{
#pragma clang fp contract(off)
    return a - b;
}

// grid (Nx / 64, Ny / MTY, ceil(levels / 64)), block (64, MTY).  KIND 3: levels = faces 1 .. Nz-1; BUOY as k_w_tendency_g.
// (The level loop inside a group is NOT unrolled and the per-level partial results live in LDS slots of their own thread: unrolled, the
// sixteen inlined order-9 reconstructions of a group made 50 - 90 KB of code and 110 VGPRs, and the kernel was slower than two passes.)
template <int R, int KIND, bool BUOY>
__global__ __launch_bounds__(64 * MTY, MARCH_WAVES) void k_tendency_m(DevGrid g, double *__restrict__ G, const double *__restrict__ ru,
                                                         const double *__restrict__ rv, const double *__restrict__ rw,
                                                         const double *__restrict__ a, const double *__restrict__ T,
                                                         const double *__restrict__ qv, FluxBuf F, int kchunk)
{
    if (F.skip_if_dry && __builtin_amdgcn_readfirstlane(*F.skip_if_dry) == 1) return;
    constexpr bool XC = (KIND == 1), YC = (KIND == 2), ZC = (KIND == 3);
    __shared__ double FY[2][MTY][MTY + 1][64];
    __shared__ double AX[MTY][MTY][64], AZ[MTY][MTY][64];
    const int tx = threadIdx.x, ty = threadIdx.y;
    int bx, by, bz;
    generic_block(bx, by, bz, GEN_XCD_MARCH != 0);
    const int i0 = bx * 64, j0 = by * MTY, i = i0 + tx, j = j0 + ty;
    const int k0 = ZC ? 1 : 0, k1 = g.Nz;                    // levels [k0, k1)
    const int kbeg = k0 + bz * kchunk, kend = min(kbeg + kchunk, k1);      // kchunk <= 64: one edge flux per lane
    if (kbeg >= kend) return;
    const long long sz = g.Sxy;
    const MarchFlux<R, KIND> Fl(g, ru, rv, rw, a);
    long long n = g.idx(i, j, kbeg);
    // x: the flux beyond the tile edge (halo cells are exact periodic images: index Nx is index 0), one level per lane
    const int ie = XC ? i0 - 1 : i0 + 64, le = XC ? 0 : 63;
    double edge;
    {
        const int kk = min(kbeg + tx, kend - 1);
        edge = Fl.X(g.idx(ie, j, kk), kk);
    }
    // y: the row outside the tile
    const int jx = YC ? j0 - 1 : j0 + MTY;
    const long long nx0 = g.idx(i, jx, kbeg);
    const int yown = ty + (YC ? 1 : 0), yext = YC ? 0 : MTY;
    // z: flux of the lower face (centre below for KIND 3) of the first level
    double zlo = ZC ? Fl.Z(n - sz, kbeg - 1) : Fl.Z(n, kbeg);
    // z stencil of the advected field around the upper face of the current level, in registers (KIND 0, 1, 2): one load per level
    // instead of 2 R (the stencil loads are what the L1 path of a CU spends its cycles on: 58 four-byte requests per cell and level in
    // the x-momentum kernel, 64 bytes per clock)
    double zr[2 * R];
    const int ktop = g.Nz + g.Hz - 1;                         // last level a parent array holds
    auto lev = [&](int kk) { return (long long)(max(min(kk, ktop), -g.Hz) - kbeg) * sz; };      // offset of level kk from n at chunk start, clamped to the array
    if (!ZC) {
#pragma unroll
        for (int q = 0; q < 2 * R; ++q) zr[q] = a[n + lev(kbeg + 1 - R + q)];
    }
    int buf = 0;
    for (int k = kbeg; k < kend; k += MTY, n += MTY * sz) {
        const int nl = min(MTY, kend - k);
#pragma unroll 1
        for (int l = 0; l < nl; ++l) {
            const int kl = k + l;
            const long long m = n + l * sz;
            const double fx = Fl.X(m, kl);
            double nb = XC ? __shfl_up(fx, 1) : __shfl_down(fx, 1);
            const double e = __shfl(edge, kl - kbeg);
            if (tx == le) nb = e;
            AX[l][ty][tx] = XC ? bz_sub_rounded(fx, nb) : bz_sub_rounded(nb, fx);
            const int reps = (ty == l) ? 2 : 1;            // wave-uniform: wave l also takes the outside row of level l
#pragma unroll 1
            for (int rep = 0; rep < reps; ++rep)
                FY[buf][l][rep ? yext : yown][tx] = Fl.Y(rep ? nx0 + (long long)(kl - kbeg) * sz : m, kl);
            double zhi;
            if (ZC) zhi = Fl.Z(m, kl);
            else {
                if (kl + 1 >= g.Nz) zhi = 0.0;
                else {
                    const double wt = Fl.Zmass(m + sz);
                    zhi = Fl.Zfin(wt, ring_face_g<R>(zr, wt > 0.0, buf_face<R>(kl + 1, g.Nz)), kl + 1);
                }
                const double znew = a[m + (long long)(min(kl + 1 + R, ktop) - kl) * sz];      // level kl + 1 + R: top of the next face's ring
#pragma unroll
                for (int q = 0; q < 2 * R - 1; ++q) zr[q] = zr[q + 1];
                zr[2 * R - 1] = znew;
            }
            AZ[l][ty][tx] = bz_sub_rounded(zhi, zlo);
            zlo = zhi;
        }
        __syncthreads();
#pragma unroll 1
        for (int l = 0; l < nl; ++l) {
            const int kl = k + l;
            const long long m = n + l * sz;
            const double fy = FY[buf][l][yown][tx];
            const double dy = YC ? bz_sub_rounded(fy, FY[buf][l][ty][tx]) : bz_sub_rounded(FY[buf][l][ty + 1][tx], fy);
            const double adv = -((ZC ? g.Vinv_f[kl] : g.Vinv_c[kl]) * (AX[l][ty][tx] + dy + AZ[l][ty][tx]));
            if (ZC) {
                if (BUOY) G[m] = rk_out(F, adv + 0.5 * (bz_buoyancy(g, T, qv, m - sz, kl - 1) + bz_buoyancy(g, T, qv, m, kl)), m);
                else G[m] = adv;
            } else G[m] = rk_out(F, adv, m);
        }
        buf ^= 1;
    }
}

// scratch of the two-pass path: three parent-shaped arrays with z-face levels
static int generic_flux_buffers(bz_ctx *ctx, FluxBuf &F)
{
    const DevGrid &g = ctx->dg;
    const size_t n = (size_t)g.Sxy * (size_t)(g.Nz + 2 * g.Hz + 1);
    if (!ctx->d_gflux) BZ_HIP(hipMalloc(&ctx->d_gflux, 3 * n * sizeof(double)));
    const long long origin = (long long)g.Sxy * g.Hz + (long long)g.Sx * g.Hy + g.Hx;      // g.idx(0, 0, 0) of a parent array
    (void)origin;
    F.x = ctx->d_gflux; F.y = ctx->d_gflux + n; F.z = ctx->d_gflux + 2 * n;
    return BZ_OK;
}

// launch `kernel<R, PASS>`(args..., F): one pass with BZ_GENERIC_ONEPASS=1, else flux pass over the extended box + divergence pass
#define GENERIC_LAUNCH(KERNEL, TARGS, k0, nk, ...)                                                                                  \
    do {                                                                                                                            \
        const dim3 block(256);                                                                                                      \
        if (onepass) {                                                                                                              \
            hipLaunchKernelGGL((KERNEL<R, 0 TARGS>), dim3((g.Nx + 255) / 256, g.Ny, (nk)), block, 0, ctx->stream, g, __VA_ARGS__, F);  \
        } else {                                                                                                                    \
            hipLaunchKernelGGL((KERNEL<R, 1 TARGS>), dim3((g.Nx + 255) / 256, g.Ny + ((!g.wrap_y && !g.flat_y) ? 2 : 0), g.Nz), block, 0, \
                               ctx->stream, g, __VA_ARGS__, F);                                                                     \
            hipLaunchKernelGGL((KERNEL<R, 2 TARGS>), dim3((g.Nx + 255) / 256, g.Ny, (nk)), block, 0, ctx->stream, g, __VA_ARGS__, F);  \
        }                                                                                                                           \
    } while (0)
#define COMMA_FALSE , false
// the single-pass marching kernel where the grid allows it (see k_tendency_m), else the two passes
static bool generic_march_ok(const bz_ctx *ctx)
{
    const DevGrid &g = ctx->dg;
    return !ctx->tune.generic_onepass && !ctx->tune.no_generic_march && g.wrap_y && !g.flat_y && !g.bounded_x && !g.bounded_y &&
           g.Nx % 64 == 0 && g.Ny % MTY == 0 && g.Nz > 1;
}
// levels per workgroup of the march: 64 where that leaves >= 8 wavefronts per SIMD, else shorter chunks (every chunk pays one edge flux
// and one first z flux per thread: 1 / chunk extra reconstructions; BOMEX 256 x 256 x 128 at 64 levels had two wavefronts per SIMD and
// ran 19 % slower than the two passes)
static int march_chunk(const DevGrid &g)
{
    int kc = 64;
    while (kc > 8 && (long long)(g.Nx / 64) * g.Ny * ((g.Nz + kc - 1) / kc) < 8192) kc >>= 1;
    return kc;
}
#define MARCH_LAUNCH(KIND, BUOY, nlev, ...)                                                                                         \
    hipLaunchKernelGGL((k_tendency_m<R, KIND, BUOY>), dim3(g.Nx / 64, g.Ny / MTY, ((nlev) + mkc - 1) / mkc), dim3(64, MTY), 0, ctx->stream, g, \
                       __VA_ARGS__, F, mkc)

// E != nullptr: the fused-RK tier of the whole-step seam (bz_step.hip) — the divergence pass of every field applies the SSP-RK3 update:
// the predictor momentum goes to the G slots (the prognostic momentum keeps feeding the advecting fluxes of the kernels that follow),
// rho theta / rho q advance in place (the scalar kernels read the specific diagnostics, not the densities)
template <int R>
static int launch_generic(bz_ctx *ctx, const bz_state *s, const bz_prognostic *G, bool buoyancy, bool scalars, const RKEpilogue *E = nullptr,
                          const bz_prognostic *U0 = nullptr)
{
    const DevGrid &g = ctx->dg;
    const bool onepass = ctx->tune.generic_onepass, march = generic_march_ok(ctx);
    const int mkc = march_chunk(g);
    FluxBuf F{};
    int rc;
    if (!onepass && !(march && MARCH_W) && (rc = generic_flux_buffers(ctx, F))) return rc;
    const double *nul = nullptr;
    auto epi = [&](double *u0, const double *uold) {
        if (!E) return;
        F.E = *E; F.E.u0 = u0; F.E.u0_out = u0; F.uold = uold;
    };
    {
        ProfileScope ps(ctx, E ? "x_momentum_tendency+rk3" : "x_momentum_tendency");
        epi(U0 ? U0->rho_u : nullptr, s->rho_u);
        if (march) MARCH_LAUNCH(1, true, g.Nz, G->rho_u, s->rho_u, s->rho_v, s->rho_w, s->u, nul, nul);
        else GENERIC_LAUNCH(k_u_tendency_g, , 0, g.Nz, G->rho_u, s->rho_u, s->rho_v, s->rho_w, s->u);
    }
    {
        ProfileScope ps(ctx, E ? "y_momentum_tendency+rk3" : "y_momentum_tendency");
        epi(U0 ? U0->rho_v : nullptr, s->rho_v);
        if (march) MARCH_LAUNCH(2, true, g.Nz, G->rho_v, s->rho_u, s->rho_v, s->rho_w, s->v, nul, nul);
        else GENERIC_LAUNCH(k_v_tendency_g, , 0, g.Nz, G->rho_v, s->rho_u, s->rho_v, s->rho_w, s->v);
    }
    if (g.Nz > 1) {
        ProfileScope ps(ctx, E ? "z_momentum_tendency+rk3" : "z_momentum_tendency");
        epi(U0 ? U0->rho_w : nullptr, s->rho_w);
        // (the marching kernel of the z-momentum measured slower than the two passes: 2.16 against 1.82 ms per stage on the CBL case — its
        // three fluxes each carry the z-cascade of the Centered interpolation; MARCH_W=1 builds select it)
        if (MARCH_W && march && buoyancy) MARCH_LAUNCH(3, true, g.Nz - 1, G->rho_w, s->rho_u, s->rho_v, s->rho_w, s->w, (const double *)s->T, (const double *)s->q);
        else if (MARCH_W && march) MARCH_LAUNCH(3, false, g.Nz - 1, G->rho_w, s->rho_u, s->rho_v, s->rho_w, s->w, nul, nul);
        else if (buoyancy) GENERIC_LAUNCH(k_w_tendency_g, , 1, g.Nz - 1, G->rho_w, s->rho_u, s->rho_v, s->rho_w, s->w, s->T, s->q);
        else GENERIC_LAUNCH(k_w_tendency_g, COMMA_FALSE, 1, g.Nz - 1, G->rho_w, s->rho_u, s->rho_v, s->rho_w, s->w, (const double *)nullptr,
                            (const double *)nullptr);
    }
    if (scalars) {
        {
            ProfileScope ps(ctx, E ? "potential_temperature_tendency+rk3" : "potential_temperature_tendency");
            epi(U0 ? U0->rho_theta : nullptr, s->rho_theta);
            if (march) MARCH_LAUNCH(0, true, g.Nz, E ? s->rho_theta : G->rho_theta, s->u, s->v, s->w, s->theta, nul, nul);
            else GENERIC_LAUNCH(k_scalar_tendency_g, , 0, g.Nz, E ? s->rho_theta : G->rho_theta, s->u, s->v, s->w, s->theta);
        }
        {
            ProfileScope ps(ctx, E ? "moisture_tendency+rk3" : "moisture_tendency");
            epi(U0 ? U0->rho_q : nullptr, s->rho_q);
            if (E) F.skip_if_dry = bzi_moisture_state(ctx);
            if (march) MARCH_LAUNCH(0, true, g.Nz, E ? s->rho_q : G->rho_q, s->u, s->v, s->w, s->q, nul, nul);
            else GENERIC_LAUNCH(k_scalar_tendency_g, , 0, g.Nz, E ? s->rho_q : G->rho_q, s->u, s->v, s->w, s->q);
            F.skip_if_dry = nullptr;
        }
    }
    BZ_LAUNCH_CHECK();
    return BZ_OK;
}

// tendencies + SSP-RK3 update of the five prognostic fields for WENO(order = 7 / 9): the fused-RK tier of bz_step.hip
int bzi_generic_tendencies_fused_rk(bz_ctx *ctx, const bz_state *s, const bz_prognostic *U0, const bz_prognostic *G, double dt, double alpha,
                                    bool first)
{
    RKEpilogue E;
    E.mode = first ? 1 : 2; E.dt = dt; E.alpha = alpha; E.oma = 1.0 - alpha;
    if (ctx->weno_R == 5) return launch_generic<5>(ctx, s, G, true, true, &E, U0);
    if (ctx->weno_R == 4) return launch_generic<4>(ctx, s, G, true, true, &E, U0);
    return BZ_ERR_INVALID;
}

// compressible split-explicit model with WENO(order = 7 / 9) (examples/splitting_supercell.jl:279): slow momentum tendencies =
// advection alone, scalars with the 3-D carrier density
int bzi_momentum_advection_generic(bz_ctx *ctx, const bz_state *s, const bz_prognostic *G)
{
    if (ctx->weno_R == 5) return launch_generic<5>(ctx, s, G, false, false);
    if (ctx->weno_R == 4) return launch_generic<4>(ctx, s, G, false, false);
    return BZ_ERR_INVALID;
}

template <int R>
static int launch_rho3d(bz_ctx *ctx, double *Gc, double *Grho, const double *rho, const double *u, const double *v, const double *w,
                        const double *c, const double *ru, const double *rv, const double *rw)
{
    const DevGrid &g = ctx->dg;
    const bool onepass = ctx->tune.generic_onepass;
    FluxBuf F{};
    int rc;
    if (!onepass && (rc = generic_flux_buffers(ctx, F))) return rc;
    GENERIC_LAUNCH(k_scalar_tendency_rho3d_g, , 0, g.Nz, Gc, Grho, rho, u, v, w, c, ru, rv, rw);
    BZ_LAUNCH_CHECK();
    return BZ_OK;
}

int bzi_scalar_rho3d_generic(bz_ctx *ctx, double *Gc, double *Grho, const double *rho, const double *u, const double *v, const double *w,
                             const double *c, const double *ru, const double *rv, const double *rw)
{
    if (ctx->weno_R == 5) return launch_rho3d<5>(ctx, Gc, Grho, rho, u, v, w, c, ru, rv, rw);
    if (ctx->weno_R == 4) return launch_rho3d<4>(ctx, Gc, Grho, rho, u, v, w, c, ru, rv, rw);
    return BZ_ERR_INVALID;
}

// G = -div_rhoUc(c) of one more scalar (Kessler species, user tracers)
template <int R>
static int launch_scalar(bz_ctx *ctx, double *Gc, const double *u, const double *v, const double *w, const double *c)
{
    const DevGrid &g = ctx->dg;
    const bool onepass = ctx->tune.generic_onepass, march = generic_march_ok(ctx);
    const int mkc = march_chunk(g);
    FluxBuf F{};
    int rc;
    const double *nul = nullptr;
    if (march) {
        MARCH_LAUNCH(0, true, g.Nz, Gc, u, v, w, c, nul, nul);
        BZ_LAUNCH_CHECK();
        return BZ_OK;
    }
    if (!onepass && (rc = generic_flux_buffers(ctx, F))) return rc;
    GENERIC_LAUNCH(k_scalar_tendency_g, , 0, g.Nz, Gc, u, v, w, c);
    BZ_LAUNCH_CHECK();
    return BZ_OK;
}

int bzi_scalar_tendency_generic(bz_ctx *ctx, double *Gc, const double *u, const double *v, const double *w, const double *c)
{
    if (ctx->scalar_R == 5) return launch_scalar<5>(ctx, Gc, u, v, w, c);
    if (ctx->scalar_R == 4) return launch_scalar<4>(ctx, Gc, u, v, w, c);
    return BZ_ERR_INVALID;
}

// momentum tendencies (advection + buoyancy) alone: the momentum half of a model whose scalars take another order (bz_tendency.hip)
int bzi_momentum_tendencies_generic(bz_ctx *ctx, const bz_state *s, const bz_prognostic *G)
{
    if (ctx->weno_R == 5) return launch_generic<5>(ctx, s, G, true, false);
    if (ctx->weno_R == 4) return launch_generic<4>(ctx, s, G, true, false);
    return BZ_ERR_INVALID;
}

// advection (+ buoyancy) tendencies of the five prognostic fields for ctx->weno_R = 4 (order 7) or 5 (order 9)
int bzi_compute_tendencies_generic(bz_ctx *ctx, const bz_state *s, const bz_prognostic *G)
{
    if (ctx->weno_R == 5) return launch_generic<5>(ctx, s, G, true, true);
    if (ctx->weno_R == 4) return launch_generic<4>(ctx, s, G, true, true);
    ctx->last_error = "bzi_compute_tendencies_generic: WENO order 7 or 9";
    return BZ_ERR_INVALID;
}
