// bz_compressible.hip — CompressibleDynamics + SplitExplicitTimeDiscretization on gfx950:
// the Wicker-Skamarock RK3 outer loop with the linearised acoustic substep loop.
//
//   time_step! / acoustic_rk3_substep!     /root/reference/src/TimeSteppers/acoustic_runge_kutta_3.jl:172-319
//   compute_slow_*_tendencies!             /root/reference/src/TimeSteppers/acoustic_substep_helpers.jl:55-149
//   acoustic_rk3_substep_loop! and kernels /root/reference/src/CompressibleEquations/acoustic_substepping.jl:318-1590
//   update_state! (compressible)           /root/reference/src/AtmosphereModels/update_atmosphere_model_state.jl:41-68
//                                          /root/reference/src/CompressibleEquations/compressible_time_stepping.jl:83-242
//
// Kernel structure of one acoustic substep (the reference launches 5 kernels + 8 halo fills per substep and moves
// ~58 words/cell; here 3 kernels, no halo fills, ~39 words/cell):
//   k_ac_horizontal        pointwise.  Klemp-2018 divergence damping of the PREVIOUS substep (it only needs the two
//                          (rho theta)' levels that are still in memory) followed by the explicit horizontal step of
//                          this substep; accumulates the time-averaged horizontal momentum.  Neighbours by periodic
//                          wrap indexing, so the perturbation fields never need halos inside the loop.
//   k_ac_column_forward    one thread per column marching upward: predictors rho'*, (rho theta)'* of cell k, right-hand
//                          side of face k, forward elimination of the tridiagonal system with the coefficients built on
//                          the fly from registers (the Thomas factors t_k are stored by the first substep of a stage and
//                          reused: they depend only on the linearisation and d tau).
//   k_ac_column_backward   one thread per column marching downward: back substitution for (rho w)', post-solve recovery
//                          of rho', (rho theta)', accumulation of the time-averaged vertical momentum.
// Stage prologue / epilogue are single pointwise kernels (k_ac_stage_init, k_ac_finalize, k_ac_recover) and the whole
// update_state! is k_cmp_diagnose, which also writes every periodic halo image and z-halo copy of what it produces.
#include <cmath>
#include <cstdlib>
#include <cstring>

#include "bz_internal.h"

extern "C" int bz_compressible_kessler_update(bz_ctx *ctx, const bz_compressible_state *s, const bz_compressible_prognostic *G,
                                              const bz_acoustic_substepper *sub, double dt);

// from bz_fused.hip
__device__ __forceinline__ void cst_img(double *__restrict__ f, long long n, double v, long long ox, long long oy)
{
    f[n] = v;
    if (ox) f[n + ox] = v;
    if (oy) {
        f[n + oy] = v;
        if (ox) f[n + ox + oy] = v;
    }
}
__device__ __forceinline__ void cst_img_only(double *__restrict__ f, long long n, double v, long long ox, long long oy)
{
    if (ox) f[n + ox] = v;
    if (oy) {
        f[n + oy] = v;
        if (ox) f[n + ox + oy] = v;
    }
}

struct WrapIdx {
    long long im, ip, jm, jp;    // offsets to the periodic x / y neighbours of (i, j)
    long long ox, oy;            // offsets of this cell's periodic halo images (0: none)
};
__device__ __forceinline__ WrapIdx wrap_of(const DevGrid &g, int i, int j)
{
    WrapIdx w;
    // Bounded x (lateral walls of the acoustic loop, round 6): columns -1 and Nx are halo columns — filled by k_ac_fill_walls for the
    // substepper's own fields (their default zero-gradient boundary condition), by the caller for the model's
    w.im = (i > 0 || g.bounded_x) ? -1 : g.Nx - 1;
    w.ip = (i + 1 < g.Nx || g.bounded_x) ? 1 : 1 - g.Nx;
    // y-slab mode (wrap_y == 0): rows -1 and Ny are halo rows delivered by the caller's neighbour exchange (Bounded y: as Bounded x above)
    w.jm = (j > 0 || !g.wrap_y) ? -(long long)g.Sx : (long long)g.Sx * (g.Ny - 1);
    w.jp = (j + 1 < g.Ny || !g.wrap_y) ? (long long)g.Sx : (long long)g.Sx * (1 - g.Ny);
    w.ox = g.bounded_x ? 0 : (i < g.Hx) ? g.Nx : (i >= g.Nx - g.Hx) ? -(long long)g.Nx : 0;
    w.oy = !g.wrap_y ? 0 : (j < g.Hy) ? (long long)g.Ny * g.Sx : (j >= g.Ny - g.Hy) ? -(long long)g.Ny * g.Sx : 0;
    return w;
}

// ---------------------------------------------------------------------------------------------------------------------
// update_state!: total density, halos, velocities, theta, q, T (Newton), p   [+ linearisation when LIN]
// ---------------------------------------------------------------------------------------------------------------------
struct DiagFields {
    double *rho_d, *rho, *ru, *rv, *rw, *rth, *rq;
    double *u, *v, *w, *theta, *q, *T, *p;
    double *Pi, *thL, *gR, *Clin;       // LIN
    int st32;                           // LIN: the four linearisation arrays are stored as float (substep_floattype = Float32)
};
// store into a working array of the substepper in its storage type (wave-uniform branch)
__device__ __forceinline__ void st_store(double *p, long long n, double v, int st32)
{
    if (st32) ((float *)p)[n] = (float)v;
    else p[n] = v;
}

// FULL: everything; !FULL: halos of rho_d, rho_theta, momentum + velocities only (tail of acoustic_rk3_substep_loop!)
// KES: DCMIP2016 Kessler species — total density includes rho q^cl + rho q^r, q = (q^v, q^cl + q^r) in R_m, c_pm and the
// latent term of the temperature inversion, q^cl / q^r / q^v diagnosed (dcmip2016_kessler.jl:222-227,298-303,860-865)
// MP = 1: SaturationAdjustment(WarmPhaseEquilibrium) on the density-based state — rho q is the total moisture, q^v / q^l are
// diagnosed by bz_ds_adjust at the cell's own total density and the temperature is the same Newton inversion with the latent
// term (compressible_time_stepping.jl:191-250; saturation_adjustment.jl:236-301)
template <bool FULL, bool LIN, int MP = 0>
__global__ __launch_bounds__(256) void k_cmp_diagnose(DevGrid g, DiagFields F, double abstol, int maxiter)
{
    constexpr bool KES = (MP == 2), SA = (MP == 1);
    const int i = blockIdx.x * 256 + threadIdx.x, j = blockIdx.y, k = blockIdx.z;
    if (i >= g.Nx) return;
    const long long sz = g.Sxy;
    const WrapIdx W = wrap_of(g, i, j);
    const long long ox = W.ox, oy = W.oy;
    const long long n = g.idx(i, j, k);
    const bool bot = (k == 0), top = (k == g.Nz - 1);

    const double rd = F.rho_d[n];
    const double rdx = (rd + F.rho_d[n + W.im]) / 2.0;
    const double rdy = (rd + F.rho_d[n + W.jm]) / 2.0;
    const double ru = F.ru[n], rv = F.rv[n];
    const double u = ru / rdx, v = rv / rdy;
    const double rth = F.rth[n];
    cst_img_only(F.rho_d, n, rd, ox, oy);
    cst_img_only(F.ru, n, ru, ox, oy);
    cst_img_only(F.rv, n, rv, ox, oy);
    cst_img_only(F.rth, n, rth, ox, oy);
    cst_img(F.u, n, u, ox, oy);
    cst_img(F.v, n, v, ox, oy);
    if (!bot) {
        const double rw = F.rw[n];
        const double rdz = (rd + F.rho_d[n - sz]) / 2.0;
        cst_img_only(F.rw, n, rw, ox, oy);
        cst_img(F.w, n, rw / rdz, ox, oy);
    } else {
        cst_img(F.rw, n, 0.0, ox, oy);      // impenetrable walls
        cst_img(F.w, n, 0.0, ox, oy);
    }
    if (top) {
        cst_img(F.rw, n + sz, 0.0, ox, oy);
        cst_img(F.w, n + sz, 0.0, ox, oy);
    }
    double r = 0.0, q = 0.0, th = 0.0, T = 0.0, p = 0.0, rq = 0.0, qcl_v = 0.0, qr_v = 0.0, sa_qv = 0.0, sa_ql = 0.0;
    if (FULL) {
        rq = F.rq[n];
        double rqcl = 0.0, rqr = 0.0, ql = 0.0;
        if (KES) {
            rqcl = g.rqcl_field[n];
            rqr = g.rqr_field[n];
            r = rd + (rq + (rqcl + (rqr + 0.0)));
        } else {
            r = rd + (rq + 0.0);
        }
        th = rth / rd;
        q = rq / r;
        if (KES) {
            qcl_v = rqcl / r;
            qr_v = rqr / r;
            ql = qcl_v + qr_v;
        }
        double qvap = q;        // vapour fraction of the mixture constants (q itself unless the adjustment partitions it)
        if (SA) {
            T = bz_ds_adjust(g, th, q, r, abstol, maxiter, qvap, ql);
            cst_img(g.qv_field, n, qvap, ox, oy);
            cst_img(g.ql_field, n, ql, ox, oy);
            sa_qv = qvap; sa_ql = ql;
        }
        const double qd = 1.0 - (qvap + ql);
        const double Rm = qd * g.Rd + qvap * g.Rv;
        const double cpm = (KES || SA) ? qd * g.cpd + qvap * g.cpv + ql * g.sa_cl : qd * g.cpd + qvap * g.cpv;
        if (!SA) {
            const double kap = Rm / cpm;
            const double gam = cpm / (cpm - Rm);
            const double Lt = KES ? (g.sa_Ll * ql) / cpm : 0.0;
            T = pow(th, gam) * pow(r * Rm / g.pst, gam - 1.0) + Lt;
            double dT = T;
            for (int it = 0; it < maxiter && fabs(dT) > abstol; ++it) {
                const double Phi = pow(r * Rm * T / g.pst, kap) * th;
                dT = -(T - Phi - Lt) / (1.0 - kap * Phi / T);
                T += dT;
            }
        }
        p = r * Rm * T;
        if (KES) {
            cst_img_only(g.rqcl_field, n, rqcl, ox, oy);
            cst_img_only(g.rqr_field, n, rqr, ox, oy);
            cst_img(g.qcl_field, n, qcl_v, ox, oy);
            cst_img(g.qr_field, n, qr_v, ox, oy);
            cst_img(g.qv_field, n, q, ox, oy);
        }
        cst_img_only(F.rq, n, rq, ox, oy);
        cst_img(F.rho, n, r, ox, oy);
        cst_img(F.theta, n, th, ox, oy);
        cst_img(F.q, n, q, ox, oy);
        cst_img(F.T, n, T, ox, oy);
        cst_img(F.p, n, p, ox, oy);
        if (LIN) {
            const double Pi = pow(p / g.pst, g.Rd / g.cpd);
            const double thl = rth / ((rd == 0.0) ? 1.0 : rd);
            const double gr = cpm * Rm / (cpm - Rm);
            st_store(F.Pi, n, Pi, F.st32);
            st_store(F.thL, n, thl, F.st32);
            st_store(F.gR, n, gr, F.st32);
            st_store(F.Clin, n, gr * Pi, F.st32);
        }
    }
    if (bot || top) {     // first z-halo cell of the no-flux centre fields
        const long long h = bot ? -sz : sz;
        cst_img(F.rho_d, n + h, rd, ox, oy);
        cst_img(F.ru, n + h, ru, ox, oy);
        cst_img(F.rv, n + h, rv, ox, oy);
        cst_img(F.rth, n + h, rth, ox, oy);
        cst_img(F.u, n + h, u, ox, oy);
        cst_img(F.v, n + h, v, ox, oy);
        if (FULL) {
            cst_img(F.rq, n + h, rq, ox, oy);
            cst_img(F.rho, n + h, r, ox, oy);
            cst_img(F.theta, n + h, th, ox, oy);
            cst_img(F.q, n + h, q, ox, oy);
            cst_img(F.T, n + h, T, ox, oy);
            cst_img(F.p, n + h, p, ox, oy);
            if (SA) {
                cst_img(g.qv_field, n + h, sa_qv, ox, oy);
                cst_img(g.ql_field, n + h, sa_ql, ox, oy);
            }
            if (KES) {
                cst_img(g.rqcl_field, n + h, g.rqcl_field[n], ox, oy);
                cst_img(g.rqr_field, n + h, g.rqr_field[n], ox, oy);
                cst_img(g.qcl_field, n + h, qcl_v, ox, oy);
                cst_img(g.qr_field, n + h, qr_v, ox, oy);
                cst_img(g.qv_field, n + h, q, ox, oy);
            }
        }
    }
}

// refresh_linearization_basic_state! (acoustic_substepping.jl:318-399)
__global__ __launch_bounds__(256) void k_cmp_linearization(DevGrid g, double *__restrict__ Pi, double *__restrict__ thL,
                                                           double *__restrict__ gR, double *__restrict__ Clin,
                                                           const double *__restrict__ p, const double *__restrict__ rho_d,
                                                           const double *__restrict__ rth, const double *__restrict__ qv, int st32, int hrows,
                                                           int hcols)
{
    // y-slab mode: hrows halo rows on each side are linearised locally (their inputs arrive with the state's halo exchange): one for
    // the substep kernels, two with DirectDivergenceDamping, whose delta of row -1 averages theta_L of rows -2 and -1
    // Bounded x / y: one halo column / row each side holds the zero-gradient copy of the adjacent interior cell — the fill_halo_regions! of
    // acoustic_substepping.jl:365-367 on fields with default boundary conditions: the value is formed from the state of that interior cell
    const int i = (int)(blockIdx.x * 256 + threadIdx.x) - hcols, j = (int)blockIdx.y - hrows, k = blockIdx.z;
    if (i >= g.Nx + hcols) return;
    const long long nd = g.idx(i, j, k);
    const int is = g.bounded_x ? min(max(i, 0), g.Nx - 1) : i, js = g.bounded_y ? min(max(j, 0), g.Ny - 1) : j;
    const long long n = g.idx(is, js, k);
    const double rd = rho_d[n];
    const double q = (g.microphysics == 1) ? g.qv_field[n] : qv[n];
    const double ql = (g.microphysics == 2) ? g.qcl_field[n] + g.qr_field[n] : ((g.microphysics == 1) ? g.ql_field[n] : 0.0);
    const double qd = 1.0 - q - ql;
    const double Rm = qd * g.Rd + q * g.Rv;
    const double cpm = g.microphysics ? qd * g.cpd + q * g.cpv + ql * g.sa_cl : qd * g.cpd + q * g.cpv;
    const double P = pow(p[n] / g.pst, g.Rd / g.cpd);
    const double gr = cpm * Rm / (cpm - Rm);
    st_store(Pi, nd, P, st32);
    st_store(thL, nd, rth[n] / ((rd == 0.0) ? 1.0 : rd), st32);
    st_store(gR, nd, gr, st32);
    st_store(Clin, nd, gr * P, st32);
}

// ---------------------------------------------------------------------------------------------------------------------
// slow scalar tendencies: G_rho_theta = -div_rhoUc(theta) with the 3-D carrier density (src/Advection.jl:20-35) and,
// when Grho != nullptr, G_rho_d = -div(momentum) (compressible_density_tendency.jl:52-55)
// ---------------------------------------------------------------------------------------------------------------------
#include "bz_weno.h"

#define CTY 4
__global__ __launch_bounds__(64 * CTY) void k_scalar_tendency_rho3d(DevGrid g, double *__restrict__ Gc, double *__restrict__ Grho,
                                                                   const double *__restrict__ rho, const double *__restrict__ u,
                                                                   const double *__restrict__ v, const double *__restrict__ w,
                                                                   const double *__restrict__ c, const double *__restrict__ ru,
                                                                   const double *__restrict__ rv, const double *__restrict__ rw,
                                                                   int kchunk, const int *__restrict__ zero_if_dry)
{
    const int i = blockIdx.x * 64 + threadIdx.x;
    const int j = blockIdx.y * CTY + threadIdx.y;
    if (i >= g.Nx || j >= g.Ny) return;
    const int k0 = blockIdx.z * kchunk;
    const int k1 = min(k0 + kchunk, g.Nz);
    const long long sy = g.Sx, sz = g.Sxy;
    long long n = g.idx(i, j, k0);
    // moisture launch of a dry model (the moisture scan's word, bz_step.hip: bzi_scan_moisture): the advected field is identically zero,
    // every flux an exact zero — the tendency is written as such without reading anything
    if (zero_if_dry && __builtin_amdgcn_readfirstlane(*zero_if_dry) == 1) {
        for (int k = k0; k < k1; ++k, n += sz) Gc[n] = 0.0;
        return;
    }

    double zm3 = c[n - 3 * sz], zm2 = c[n - 2 * sz], zm1 = c[n - sz], z0 = c[n], zp1 = c[n + sz], zp2 = c[n + 2 * sz];
    double r_lo = rho[n - sz], r0 = rho[n];
    double Fz_lo;
    {
        const double wt = w[n];
        const double cR = bz_upB(zm3, zm2, zm1, z0, zp1, zp2, wt > 0.0, bz_buffer_face(k0, g.Nz));
        Fz_lo = ((r0 + r_lo) / 2.0) * ((g.Az * wt) * cR);
    }
    for (int k = k0; k < k1; ++k, n += sz) {
        const double zp3 = c[n + 3 * sz];
        const double r_hi = rho[n + sz];
        double Fz_hi;
        {
            const double wt = w[n + sz];
            const double cR = bz_upB(zm2, zm1, z0, zp1, zp2, zp3, wt > 0.0, bz_buffer_face(k + 1, g.Nz));
            Fz_hi = ((r_hi + r0) / 2.0) * ((g.Az * wt) * cR);
        }
        const double Ax = g.Ax[k], Ay = g.Ay[k];
        const double xm3 = c[n - 3], xm2 = c[n - 2], xm1 = c[n - 1], xp1 = c[n + 1], xp2 = c[n + 2], xp3 = c[n + 3];
        const double u0 = u[n], u1 = u[n + 1];
        const double Fx_lo = ((r0 + rho[n - 1]) / 2.0) * ((Ax * u0) * bz_up5(xm3, xm2, xm1, z0, xp1, xp2, u0 > 0.0));
        const double Fx_hi = ((rho[n + 1] + r0) / 2.0) * ((Ax * u1) * bz_up5(xm2, xm1, z0, xp1, xp2, xp3, u1 > 0.0));
        double Fy_lo = 0.0, Fy_hi = 0.0;
        if (!g.flat_y) {
            const double ym3 = c[n - 3 * sy], ym2 = c[n - 2 * sy], ym1 = c[n - sy], yp1 = c[n + sy], yp2 = c[n + 2 * sy], yp3 = c[n + 3 * sy];
            const double v0 = v[n], v1 = v[n + sy];
            Fy_lo = ((r0 + rho[n - sy]) / 2.0) * ((Ay * v0) * bz_up5(ym3, ym2, ym1, z0, yp1, yp2, v0 > 0.0));
            Fy_hi = ((rho[n + sy] + r0) / 2.0) * ((Ay * v1) * bz_up5(ym2, ym1, z0, yp1, yp2, yp3, v1 > 0.0));
        }
        Gc[n] = -(g.Vinv_c[k] * ((Fx_hi - Fx_lo) + (Fy_hi - Fy_lo) + (Fz_hi - Fz_lo)));
        if (Grho) {
            const double a = Ax * ru[n + 1] - Ax * ru[n];
            const double b = g.flat_y ? 0.0 : Ay * rv[n + sy] - Ay * rv[n];
            const double cc = g.Az * rw[n + sz] - g.Az * rw[n];
            Grho[n] = -(g.Vinv_c[k] * (a + b + cc));
        }
        zm3 = zm2; zm2 = zm1; zm1 = z0; z0 = zp1; zp1 = zp2; zp2 = zp3;
        Fz_lo = Fz_hi;
        r_lo = r0; r0 = r_hi;
    }
}

// The same tendency with every face flux evaluated ONCE (round 4).  The kernel above evaluates both x faces, both y faces and the upper
// z face of its cell: five order-5 reconstructions per cell where three are needed.  Here a thread evaluates the fluxes through its
// low x face and low y face and through the upper z face; the high x-face flux comes from the next lane (the one beyond the tile edge:
// one evaluation per lane for the 64 levels of the march, read back with a second shuffle), the high y-face flux through an LDS row
// exchange of CTY levels at a time in which wave l evaluates the row outside the tile for level l (every wave: 3 CTY + 1 reconstructions
// per group).  Same expressions per flux; the differences see rounded fluxes (bz_sub_rounded_c), where the kernel above lets the compiler
// contract one of each pair into an fma — results differ from it by an ulp of a flux, both within 1e-12 of the oracle.
// grid (Nx / 64, Ny / CTY, ceil(Nz / 64)): rows of a multiple of 64 cells, Ny a multiple of CTY, not Flat; halo rows in y are read as
// they are (periodic images or a slab neighbour's rows).
// a value one lane up / down the 64-lane wavefront (lane l receives lane l - 1 / l + 1) as two v_mov_b32_dpp wave_shr:1 / wave_shl:1 — 4 cycles of
// the vector ALU each where __shfl_up / __shfl_down are ds_bpermute_b32 at 10 ns of the CU's LDS pipe (instruction costs;
// tools/dpp_check.hip: the same values, also with the upper lanes of a ragged row gone)
template <int CTRL, class T>
__device__ __forceinline__ T ac_lane_shift(T v)
{
    if constexpr (sizeof(T) == 8) {
        const long long b = __builtin_bit_cast(long long, v);
        int lo = (int)b, hi = (int)(b >> 32);
        lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xf, 0xf, false);
        hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xf, 0xf, false);
        return __builtin_bit_cast(T, ((long long)hi << 32) | (long long)(unsigned)lo);
    } else {
        return __builtin_bit_cast(T, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, false));
    }
}
template <bool DPP>
__device__ __forceinline__ double ac_lane_up(double v) { return DPP ? ac_lane_shift<0x138>(v) : __shfl_up(v, 1); }
template <bool DPP>
__device__ __forceinline__ double ac_lane_down(double v) { return DPP ? ac_lane_shift<0x130>(v) : __shfl_down(v, 1); }

__device__ __forceinline__ double bz_sub_rounded_c(double a, double b)
{
#pragma clang fp contract(off)
    return a - b;
}
__global__ __launch_bounds__(64 * CTY) void k_scalar_tendency_rho3d_x(DevGrid g, double *__restrict__ Gc, double *__restrict__ Grho,
                                                                     const double *__restrict__ rho, const double *__restrict__ u,
                                                                     const double *__restrict__ v, const double *__restrict__ w,
                                                                     const double *__restrict__ c, const double *__restrict__ ru,
                                                                     const double *__restrict__ rv, const double *__restrict__ rw,
                                                                     const int *__restrict__ zero_if_dry, int kchunk)
{
    __shared__ double FY[2][CTY][CTY + 1][64];
    __shared__ double AX[CTY][CTY][64], AZ[CTY][CTY][64];
    const int tx = threadIdx.x, ty = threadIdx.y;
    // Round 6: every XCD owns a band of tile rows (gridDim.y a multiple of 8).  A 64 x 4 tile reads nine rows of c for its four (y stencil)
    // and the row below of rho; in launch order (x fastest, round-robin over the eight XCDs) the tiles above and below sit behind other
    // L2s and every tile fetched its frame itself: PMC 1.4 x the compulsory bytes at 5.4 TB/s of real traffic — the kernel was
    // bandwidth-bound on re-reads.  (A pipelined form — next level's loads in flight — was measured equal and removed.)
    int bxr = blockIdx.x, byr = blockIdx.y;
    if ((gridDim.y & 7u) == 0) {
        const unsigned wv = blockIdx.y * gridDim.x + blockIdx.x, cx = wv & 7u, rr = wv >> 3;
        bxr = (int)(rr % gridDim.x);
        byr = (int)(cx * (gridDim.y >> 3) + rr / gridDim.x);
    }
    const int i0 = bxr * 64, j0 = byr * CTY, i = i0 + tx, j = j0 + ty;
    const int k0 = blockIdx.z * kchunk, k1 = min(k0 + kchunk, g.Nz);      // kchunk <= 64: one edge flux per lane
    const long long sy = g.Sx, sz = g.Sxy;
    long long n = g.idx(i, j, k0);
    if (zero_if_dry && __builtin_amdgcn_readfirstlane(*zero_if_dry) == 1) {
        for (int k = k0; k < k1; ++k, n += sz) Gc[n] = 0.0;
        return;
    }
    auto FX = [&](long long m, int k) {      // flux through the low x face of cell m
        const double u0 = u[m];
        return ((rho[m] + rho[m - 1]) / 2.0) * ((g.Ax[k] * u0) * bz_up5(c[m - 3], c[m - 2], c[m - 1], c[m], c[m + 1], c[m + 2], u0 > 0.0));
    };
    auto FYf = [&](long long m, int k) {     // low y face
        const double v0 = v[m];
        return ((rho[m] + rho[m - sy]) / 2.0) *
               ((g.Ay[k] * v0) * bz_up5(c[m - 3 * sy], c[m - 2 * sy], c[m - sy], c[m], c[m + sy], c[m + 2 * sy], v0 > 0.0));
    };
    double edge;
    {
        const int kk = min(k0 + tx, k1 - 1);
        edge = FX(g.idx(i0 + 64, j, kk), kk);
    }
    const long long nx0 = g.idx(i, j0 + CTY, k0);
    double zm3 = c[n - 3 * sz], zm2 = c[n - 2 * sz], zm1 = c[n - sz], z0 = c[n], zp1 = c[n + sz], zp2 = c[n + 2 * sz];
    double r_lo = rho[n - sz], r0 = rho[n];
    double Fz_lo;
    {
        const double wt = w[n];
        const double cR = bz_upB(zm3, zm2, zm1, z0, zp1, zp2, wt > 0.0, bz_buffer_face(k0, g.Nz));
        Fz_lo = ((r0 + r_lo) / 2.0) * ((g.Az * wt) * cR);
    }
    int buf = 0;
    for (int k = k0; k < k1; k += CTY, n += CTY * sz) {
        const int nl = min(CTY, k1 - k);
#pragma unroll 1
        for (int l = 0; l < nl; ++l) {
            const int kl = k + l;
            const long long m = n + l * sz;
            // Round 6: every load of the level first.  The reconstructions branch on the wave's upwind direction (bz_up5), so a load written
            // inside a flux expression stays in that flux's basic block: the ISA of the round-4 form drained the memory counter four
            // times per level (z, x, y, outside row).  Same expressions, same bits.
            const double zp3 = c[m + 3 * sz];
            const double r_hi = rho[m + sz];
            const double wt = w[m + sz];
            const double u0 = u[m], rxm = rho[m - 1];
            const double xm3 = c[m - 3], xm2 = c[m - 2], xm1 = c[m - 1], xp1 = c[m + 1], xp2 = c[m + 2];
            const double v0 = v[m], rym = rho[m - sy];
            const double ym3 = c[m - 3 * sy], ym2 = c[m - 2 * sy], ym1 = c[m - sy], yp1 = c[m + sy], yp2 = c[m + 2 * sy];
            const double cR = bz_upB(zm2, zm1, z0, zp1, zp2, zp3, wt > 0.0, bz_buffer_face(kl + 1, g.Nz));
            const double Fz_hi = ((r_hi + r0) / 2.0) * ((g.Az * wt) * cR);
            AZ[l][ty][tx] = bz_sub_rounded_c(Fz_hi, Fz_lo);
            // (z0 = c[m] and r0 = rho[m]: the ring values of this level)
            const double fx = ((r0 + rxm) / 2.0) * ((g.Ax[kl] * u0) * bz_up5(xm3, xm2, xm1, z0, xp1, xp2, u0 > 0.0));
            double nb = __shfl_down(fx, 1);
            const double e = __shfl(edge, kl - k0);
            if (tx == 63) nb = e;
            AX[l][ty][tx] = bz_sub_rounded_c(nb, fx);
            FY[buf][l][ty][tx] = ((r0 + rym) / 2.0) * ((g.Ay[kl] * v0) * bz_up5(ym3, ym2, ym1, z0, yp1, yp2, v0 > 0.0));
            if (ty == l) FY[buf][l][CTY][tx] = FYf(nx0 + (long long)(kl - k0) * sz, kl);      // the row outside the tile: one wave per level
            zm3 = zm2; zm2 = zm1; zm1 = z0; z0 = zp1; zp1 = zp2; zp2 = zp3;
            Fz_lo = Fz_hi;
            r_lo = r0; r0 = r_hi;
        }
        __syncthreads();
#pragma unroll 1
        for (int l = 0; l < nl; ++l) {
            const int kl = k + l;
            const long long m = n + l * sz;
            const double dy = bz_sub_rounded_c(FY[buf][l][ty + 1][tx], FY[buf][l][ty][tx]);
            Gc[m] = -(g.Vinv_c[kl] * (AX[l][ty][tx] + dy + AZ[l][ty][tx]));
            if (Grho) {
                const double Ax = g.Ax[kl], Ay = g.Ay[kl];
                const double a = Ax * ru[m + 1] - Ax * ru[m];
                const double b = Ay * rv[m + sy] - Ay * rv[m];
                const double cc = g.Az * rw[m + sz] - g.Az * rw[m];
                Grho[m] = -(g.Vinv_c[kl] * (a + b + cc));
            }
        }
        buf ^= 1;
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// Round 6: the same tendency with every stencil read from LDS tiles (the structure of k6_u, bz_tendency5_kernels.h).  The kernel above
// issues 22 vector-memory instructions per wave and level (stencils straight from the L1) and keeps a CU's address path 84 - 95 % busy
// (tools/gpu_sq_one_kernel.sh: ~32 TA cycles per 64-lane 8-byte load) at 1.3 x its compulsory bytes; here a 64 x SLT tile stages c with
// its three-cell frame and rho with its low-side frame once per level (double-buffered; every global load of an iteration is a prefetch
// for the NEXT level: ring tops, the thread's frame cells, its u, v, w), the x and y stencils are ds_reads, the high x-face flux comes
// from the next lane (beyond the tile: one evaluation per lane for 64 levels), the high y-face flux from the row above through LDS (the
// row outside the tile: wave 0).  7 loads per thread and level (13 with G_rho).  Same expressions per flux as k_scalar_tendency_rho3d_x:
// same bits.  grid (Nx / 64, Ny / SLT, chunks): Nx a multiple of 64, Ny of SLT; XCD bands of tile rows where gridDim.y is a multiple of 8.
// ---------------------------------------------------------------------------------------------------------------------
#ifndef SLT
#define SLT 8
#endif
template <bool GRHO>
__global__ __launch_bounds__(64 * SLT, 2) void k_scalar_rho3d_lds(DevGrid g, double *__restrict__ Gc, double *__restrict__ Grho,
                                                                const double *__restrict__ rho, const double *__restrict__ u,
                                                                const double *__restrict__ v, const double *__restrict__ w,
                                                                const double *__restrict__ c, const double *__restrict__ ru,
                                                                const double *__restrict__ rv, const double *__restrict__ rw,
                                                                const int *__restrict__ zero_if_dry, int kchunk)
{
    constexpr int TY = SLT, TR = TY + 6, TC = 72, RR = TY + 2, RC = 68, NT = 64 * TY;
    constexpr int NHC = TR * 70 - TY * 64;      // frame cells of the c tile (468 for TY = 8)
    constexpr int NHR = RR * 65 - TY * 64;      // frame cells of the rho tile: row -1, row TY (cols -1 .. 63), column -1 of rows 0 .. TY-1 (138)
    static_assert(NHC <= NT && NHR <= NT, "one frame cell of each tile per thread");
    __shared__ double C[2][TR][TC];             // c:   tile row r (-3 .. TY+2) at [r + 3], column q (-3 .. 66) at [q + 3]
    __shared__ double R[2][RR][RC];             // rho: tile row r (-1 .. TY)   at [r + 1], column q (-1 .. 63) at [q + 1]
    __shared__ double FY[2][TY + 1][64];
    const int tx = threadIdx.x, ty = threadIdx.y, t = ty * 64 + tx, tc = tx + 3;
    int bxr = blockIdx.x, byr = blockIdx.y;
    if ((gridDim.y & 7u) == 0) {
        const unsigned wv = blockIdx.y * gridDim.x + blockIdx.x, cx = wv & 7u, rr = wv >> 3;
        bxr = (int)(rr % gridDim.x);
        byr = (int)(cx * (gridDim.y >> 3) + rr / gridDim.x);
    }
    const int i0 = bxr * 64, j0 = byr * TY, i = i0 + tx, j = j0 + ty;
    const int k0 = blockIdx.z * kchunk, k1 = min(k0 + kchunk, g.Nz);      // kchunk <= 64: one edge flux per lane
    if (k0 >= k1) return;
    const long long sy = g.Sx, sz = g.Sxy;
    long long n = g.idx(i, j, k0);
    if (zero_if_dry && __builtin_amdgcn_readfirstlane(*zero_if_dry) == 1) {
        for (int k = k0; k < k1; ++k, n += sz) Gc[n] = 0.0;
        return;
    }
    // frame cell of the c tile
    const bool hc = t < NHC;
    int hcr = 0, hcc = 0;
    {
        const int h = hc ? t : 0;
        if (h < 6 * 70) { const int rr = h / 70; hcc = h - rr * 70; hcr = (rr < 3) ? rr : TY + rr; }      // rows -3 .. -1 and TY .. TY+2, whole width
        else { const int hh = h - 6 * 70, rr = hh / 6, cc = hh - rr * 6; hcr = 3 + rr; hcc = (cc < 3) ? cc : 64 + cc; }      // side columns of the interior rows
    }
    const long long hcn = g.idx(i0 - 3 + hcc, j0 - 3 + hcr, k0);
    // frame cell of the rho tile
    const bool hr = t < NHR;
    int hrr = 0, hrc = 0;
    {
        const int h = hr ? t : 0;
        if (h < 2 * 65) { const int rr = h / 65; hrc = h - rr * 65; hrr = rr ? TY + 1 : 0; }      // rows -1 and TY
        else { hrr = 1 + (h - 2 * 65); hrc = 0; }                                                  // column -1 of rows 0 .. TY-1
    }
    const long long hrn = g.idx(i0 - 1 + hrc, j0 - 1 + hrr, k0);
    // x flux beyond the tile (column i0 + 64), one level per lane, straight from memory
    auto FXg = [&](long long m, int k) {
        const double u0 = u[m];
        return ((rho[m] + rho[m - 1]) / 2.0) * ((g.Ax[k] * u0) * bz_up5(c[m - 3], c[m - 2], c[m - 1], c[m], c[m + 1], c[m + 2], u0 > 0.0));
    };
    double edge;
    {
        const int kk = min(k0 + tx, k1 - 1);
        edge = FXg(g.idx(i0 + 64, j, kk), kk);
    }
    // z ring of the own column, densities of the levels k-1, k, k+1, lower z flux
    double zm3 = c[n - 3 * sz], zm2 = c[n - 2 * sz], zm1 = c[n - sz], z0 = c[n], zp1 = c[n + sz], zp2 = c[n + 2 * sz];
    double r0 = rho[n], r_hi = rho[n + sz];
    double Fz_lo;
    {
        const double wt = w[n], r_lo = rho[n - sz];
        const double cR = bz_upB(zm3, zm2, zm1, z0, zp1, zp2, wt > 0.0, bz_buffer_face(k0, g.Nz));
        Fz_lo = ((r0 + r_lo) / 2.0) * ((g.Az * wt) * cR);
    }
    double u0 = u[n], v0 = v[n], wt = w[n + sz];
    double grw_lo = GRHO ? rw[n] : 0.0;
    const long long nvt = g.idx(i, j0 + TY, k0);      // v of the row outside the tile (wave 0)
    double vT = (ty == 0) ? v[nvt] : 0.0;
    // tiles of level k0
    C[0][ty + 3][tc] = z0;
    R[0][ty + 1][tx + 1] = r0;
    if (hc) C[0][hcr][hcc] = c[hcn];
    if (hr) R[0][hrr][hrc] = rho[hrn];
    __syncthreads();
    int buf = 0;
    for (int k = k0; k < k1; ++k, n += sz) {
        const long long lev1 = (long long)(k + 1 - k0) * sz;
        // ---- prefetch for level k + 1 (consumed at the end of this iteration) ----
        const double p_zp3 = c[n + 3 * sz];
        const double p_rn = rho[n + 2 * sz];
        const double p_w = w[n + 2 * sz];
        const double p_u = u[n + sz], p_v = v[n + sz];
        const double p_hc = hc ? c[hcn + lev1] : 0.0;
        const double p_hr = hr ? rho[hrn + lev1] : 0.0;
        const double p_vT = (ty == 0) ? v[nvt + lev1] : 0.0;
        const double Ax = g.Ax[k], Ay = g.Ay[k];
        if (GRHO) {      // G_rho = -div(rho u) of the cell: the x neighbour from the next lane, the lower z face carried from the level below
            const double gru0 = ru[n], grv0 = rv[n], grv1 = rv[n + sy], grw1 = rw[n + sz];
            double gru1 = ac_lane_down<true>(gru0);
            if (tx == 63) gru1 = ru[n + 1];
            const double a = Ax * gru1 - Ax * gru0;
            const double b = Ay * grv1 - Ay * grv0;
            const double cc = g.Az * grw1 - g.Az * grw_lo;
            Grho[n] = -(g.Vinv_c[k] * (a + b + cc));
            grw_lo = grw1;
        }
        const double(*Ck)[TC] = C[buf];
        const double(*Rk)[RC] = R[buf];
        // ---- z: upper face ----
        const double cR = bz_upB(zm2, zm1, z0, zp1, zp2, p_zp3, wt > 0.0, bz_buffer_face(k + 1, g.Nz));
        const double Fz_hi = ((r_hi + r0) / 2.0) * ((g.Az * wt) * cR);
        const double dz = bz_sub_rounded_c(Fz_hi, Fz_lo);
        // ---- x: low face of the own cell ----
        const double *cr = Ck[ty + 3] + tc;
        const double fx = ((r0 + Rk[ty + 1][tx]) / 2.0) * ((Ax * u0) * bz_up5(cr[-3], cr[-2], cr[-1], z0, cr[1], cr[2], u0 > 0.0));
        // ---- y: low face of the own cell; wave 0 also takes the row outside the tile ----
        const double fy = ((r0 + Rk[ty][tx + 1]) / 2.0) * ((Ay * v0) * bz_up5(Ck[ty][tc], Ck[ty + 1][tc], Ck[ty + 2][tc], z0, Ck[ty + 4][tc], Ck[ty + 5][tc], v0 > 0.0));
        FY[buf][ty][tx] = fy;
        if (ty == 0)
            FY[buf][TY][tx] = ((Rk[TY + 1][tx + 1] + Rk[TY][tx + 1]) / 2.0) *
                              ((Ay * vT) * bz_up5(Ck[TY][tc], Ck[TY + 1][tc], Ck[TY + 2][tc], Ck[TY + 3][tc], Ck[TY + 4][tc], Ck[TY + 5][tc], vT > 0.0));
        // ---- stage level k + 1 ----
        C[buf ^ 1][ty + 3][tc] = zp1;
        R[buf ^ 1][ty + 1][tx + 1] = r_hi;
        if (hc) C[buf ^ 1][hcr][hcc] = p_hc;
        if (hr) R[buf ^ 1][hrr][hrc] = p_hr;
        __syncthreads();
        {
            double nb = __shfl_down(fx, 1);
            const double e = __shfl(edge, k - k0);
            if (tx == 63) nb = e;
            const double dx = bz_sub_rounded_c(nb, fx);
            const double dy = bz_sub_rounded_c(FY[buf][ty + 1][tx], fy);
            Gc[n] = -(g.Vinv_c[k] * (dx + dy + dz));
        }
        zm3 = zm2; zm2 = zm1; zm1 = z0; z0 = zp1; zp1 = zp2; zp2 = p_zp3;
        Fz_lo = Fz_hi;
        r0 = r_hi; r_hi = p_rn;
        u0 = p_u; v0 = p_v; wt = p_w; vT = p_vT;
        buf ^= 1;
    }
}

static int pick_kchunk_c(const DevGrid &g, int nlev)
{
    long long tiles = (long long)((g.Nx + 63) / 64) * ((g.Ny + CTY - 1) / CTY);
    long long want = (4096 + tiles - 1) / tiles;
    if (want < 1) want = 1;
    long long maxchunks = nlev / 8 > 0 ? nlev / 8 : 1;
    if (want > maxchunks) want = maxchunks;
    return (int)((nlev + want - 1) / want);
}

// ---------------------------------------------------------------------------------------------------------------------
// acoustic stage kernels
// ---------------------------------------------------------------------------------------------------------------------
struct AcParams {
    double dtau, dtn, dto;        // substep size, omega*dtau, (1-omega)*dtau
    double d_new, d_old;          // implicit vertical damping prefactors (0 unless damp_vertical)
    double f_theta, f_w;          // thermodynamic / vertical-momentum tendency factors
    double gate;                  // 1: perturbation horizontal PGF applied this substep, 0: skipped (first small step)
    double kdamp;                 // alpha * min(dx,dy)^2 / dtau   (0: no damping)
    double inv_N;                 // 1 / N_tau
    int xcd;                      // forward sweep: 1 = every XCD owns a band of tile rows (see k_ac_column_forward), 0 = launch order
    // Round 6, dry runs inside bz_time_step_compressible: the time-averaged velocities of a stage feed one thing, the moisture (tracer)
    // tendency the NEXT stage's update uses (acoustic_runge_kutta_3.jl:189-192) — which a model whose rho q is identically zero skips
    // (bz_step.hip: moisture scan; exact zeros either way).  Where this points at the scan's word and the word says "identically zero,
    // verified by the scan that opened this call", the substep kernels of stages 1 and 2 neither read nor write the three accumulators
    // (6 of a substep's 32 words) and the stage epilogue does not form the averages; stage 3 accumulates as ever, so after the step the
    // substepper holds the averages the reference leaves.  nullptr: always accumulate (per-operator entry points, slabs, moist models).
    const int *skip_avg_if_dry;
    // every stage of a dry whole step (same word): rho q and q are identically zero and stay so (the skipped moisture tendency is an exact
    // zero): the stage epilogue neither reads U0_rho_q, G_rho_q nor writes rho q, q (4 of its 41 words)
    const int *dry_q;
    // Bounded lateral topology (round 6): 1 = the west / south face of (rho u)' / (rho v)' is an impenetrable wall (the model's momentum boundary
    // condition there is the default one) and is held at zero — enforce_wall_impenetrability! (acoustic_substepping.jl:1378-1395); 0 = an
    // active open boundary, whose face the substep kernels advance like any other.  The east / north wall face (index N + 1 of the
    // reference's face field) is written by no kernel of the reference's loop — all of them are launched over :xyz — and keeps the zero the
    // field was built with: here it is an exact zero in every flux that reads it, whatever the boundary condition.
    int wall_w, wall_s;
    // Round 6: <u>, <v> accumulated two substeps at a time (k_ac_forward2).  The forward sweep of substep n reads the stored (rho u)' of
    // substep n - 1 anyway (it advances it), so a pair (n - 1, n) is added in substep n as  a += (u'_{n-1} + u'_n)  and substep n - 1 neither
    // reads nor writes the accumulators: 4 of a sweep's 22 words in every other substep.  0: a += u'_n (the reference's order); 1: this substep
    // leaves the accumulators alone (the next one takes the pair); 2: this substep adds the pair.  The sum differs from the reference's
    // ((a + u'_{n-1}) + u'_n) by one rounding of a per pair.  Thermal or no damping only (the stored u'_{n-1} is then the accumulated value:
    // its damping is applied by the sweep that reads it; DirectDivergenceDamping changes the stored field after the accumulation), working
    // fields in the grid's type (a Float32-stored u' is the rounded value of what was accumulated).
    int acc_mode;
    // Round 6: the stage epilogue writes the recovered state into ANOTHER set of arrays than it reads (compressible_step_body: buffer rotation
    // — the state arrays stay intact as U0, nothing is copied into U0): rho_d goes out with the other fields (no thread reads the output set)
    // and rho q is stored also on the dry path (the output set may hold anything)
    int out_of_place;
};
// _zero_x_wall_face! / _zero_y_wall_face! (acoustic_substepping.jl:1367-1375) as a mask on the four faces a column's predictor reads: the
// reference zeroes the plane after the horizontal step and again after the damping; nothing reads the face between the kernel that writes it and
// the zeroing, so forming the zero where the face value is formed gives the same fields
__device__ __forceinline__ void ac_wall_faces(const DevGrid &g, const AcParams &P, int i, int j, double &up0, double &up1, double &vp0, double &vp1)
{
    if (g.bounded_x) {
        if (i == 0 && P.wall_w) up0 = 0.0;
        if (i == g.Nx - 1) up1 = 0.0;
    }
    if (g.bounded_y) {
        if (j == 0 && P.wall_s) vp0 = 0.0;
        if (j == g.Ny - 1) vp1 = 0.0;
    }
}
__device__ __forceinline__ bool ac_accumulate(const AcParams &P)
{
    return !(P.skip_avg_if_dry && __builtin_amdgcn_readfirstlane(*P.skip_avg_if_dry) == 1);
}

// ST = substep_floattype (acoustic_substepping.jl:199-235): the storage type of the acoustic perturbation / predictor / linearisation
// working fields.  Kernels read ST, promote to the grid's real, compute there and store ST; (rho w)', the tridiagonal right-hand side and
// factors, the time-averaged velocities and every model field stay in the grid's type.  ST = float inside a Float64 model halves the
// bytes of 12 of the arrays the substep kernels stream.
template <class ST>
struct AcFieldsT {
    // model state (stage-entry U^L; untouched by the loop)
    double *rho_d, *rth, *ru, *rv, *rw, *rq;
    const double *rho, *p;
    // outer-step start and slow tendencies
    const double *U0_rho_d, *U0_rth, *U0_ru, *U0_rv, *U0_rw, *U0_rq;
    const double *G_rho_d, *G_rth, *G_ru, *G_rv, *G_rw, *G_rq;
    double *Gp_ru, *Gp_rv;        // G_ru - dx p^L, G_rv - dy p^L of the stage (k_ac_stage_init<.., PF>; read by k_ac_forward2<.., PF>)
    // substepper
    const ST *thL, *Clin;
    ST *rp, *rthp, *rup, *rvp;
    double *rwp;
    ST *rs, *rths, *rth_old;
    // fused substep (k_ac_column_forward<.., FUSED = true>): (rho u)', (rho v)' ping-pong between rup_in (read) and rup
    // (written); (rho theta)' ping-pongs between rthp (current, read) / rth_old (previous, read) and rthp_out (written
    // by the backward sweep), so no thread reads a location another thread of the same launch writes.
    const ST *rup_in, *rvp_in;
    ST *rthp_out;
    double *au, *av, *aw;
    double *rqcl, *rqr;           // Kessler species (k_ac_recover<2>)
    const double *U0_rqcl, *U0_rqr, *G_rqcl, *G_rqr;
    double *Gs, *phi;             // slow vertical momentum tendency; forward-eliminated right-hand side
    double *tfac;                 // Thomas factors t_k
    const double *sponge;         // UpperSponge: damping_rate * ramp(z_face) per face k = 0 .. Nz (all zero without a sponge)
};
typedef AcFieldsT<double> AcFields;
#define COMMA ,
// the same fields with the working arrays seen as ST (the host allocated them in that type; every other member is copied)
template <class ST>
static AcFieldsT<ST> ac_cast(const AcFields &F)
{
    AcFieldsT<ST> R;
    R.rho_d = F.rho_d; R.rth = F.rth; R.ru = F.ru; R.rv = F.rv; R.rw = F.rw; R.rq = F.rq; R.rho = F.rho; R.p = F.p;
    R.U0_rho_d = F.U0_rho_d; R.U0_rth = F.U0_rth; R.U0_ru = F.U0_ru; R.U0_rv = F.U0_rv; R.U0_rw = F.U0_rw; R.U0_rq = F.U0_rq;
    R.G_rho_d = F.G_rho_d; R.G_rth = F.G_rth; R.G_ru = F.G_ru; R.G_rv = F.G_rv; R.G_rw = F.G_rw; R.G_rq = F.G_rq;
    R.Gp_ru = F.Gp_ru; R.Gp_rv = F.Gp_rv;
    R.thL = (const ST *)F.thL; R.Clin = (const ST *)F.Clin;
    R.rp = (ST *)F.rp; R.rthp = (ST *)F.rthp; R.rup = (ST *)F.rup; R.rvp = (ST *)F.rvp; R.rwp = F.rwp;
    R.rs = (ST *)F.rs; R.rths = (ST *)F.rths; R.rth_old = (ST *)F.rth_old;
    R.rup_in = (const ST *)F.rup_in; R.rvp_in = (const ST *)F.rvp_in; R.rthp_out = (ST *)F.rthp_out;
    R.au = F.au; R.av = F.av; R.aw = F.aw; R.rqcl = F.rqcl; R.rqr = F.rqr;
    R.U0_rqcl = F.U0_rqcl; R.U0_rqr = F.U0_rqr; R.G_rqcl = F.G_rqcl; R.G_rqr = F.G_rqr;
    R.Gs = F.Gs; R.phi = F.phi; R.tfac = F.tfac; R.sponge = F.sponge;
    return R;
}
// launch an acoustic kernel in the context's substep storage type: KERNEL is the template name, TA its leading template arguments
// with a trailing comma (or empty)
#define AC_LAUNCH(KERNEL, TA, GRID, BLOCK, FIELDS, ...)                                                                         \
    do {                                                                                                                         \
        if (ctx->substep_f32) hipLaunchKernelGGL((KERNEL<TA float>), GRID, BLOCK, 0, ctx->stream, g, ac_cast<float>(FIELDS), __VA_ARGS__);   \
        else hipLaunchKernelGGL((KERNEL<TA double>), GRID, BLOCK, 0, ctx->stream, g, FIELDS, __VA_ARGS__);                       \
    } while (0)
#define AC_LAUNCH0(KERNEL, TA, GRID, BLOCK, FIELDS)                                                                             \
    do {                                                                                                                         \
        if (ctx->substep_f32) hipLaunchKernelGGL((KERNEL<TA float>), GRID, BLOCK, 0, ctx->stream, g, ac_cast<float>(FIELDS));    \
        else hipLaunchKernelGGL((KERNEL<TA double>), GRID, BLOCK, 0, ctx->stream, g, FIELDS);                                    \
    } while (0)


// Block order of the pointwise stage kernels (grid (Nx / 256, Ny, Nz)).  They read the row below (j - 1) and the level below (k - 1) of one to
// three of their 19 - 41 arrays; in launch order row j - 1 belonged to another XCD and level k - 1 had been read a whole plane of all the arrays
// earlier (PMC r05: stage-init 12.2 GB per launch for 10.2 GB of words, stage-end 23.0 / 25.1 for 19.9 / 22.0).  Here XCD c owns the groups of
// eight rows G = c, c + 8, ... and walks a group x fastest, then its eight rows, then z: the level below is eight rows of traffic away (~0.7 MB),
// the row below inside the group too.  Ny a multiple of 64, else launch order.
#ifndef AC_PENCIL
#define AC_PENCIL 1
#endif
__device__ __forceinline__ void ac_pencil_block(int &bx, int &j, int &k)
{
    bx = blockIdx.x; j = blockIdx.y; k = blockIdx.z;
    const unsigned gx = gridDim.x, gy = gridDim.y, gz = gridDim.z;
    if (AC_PENCIL && (gy & 63u) == 0) {
        const unsigned w = bx + gx * (j + gy * k), c = w & 7u;
        unsigned r = w >> 3;
        bx = (int)(r % gx); r /= gx;
        const unsigned jj = r & 7u; r >>= 3;
        k = (int)(r % gz); r /= gz;
        j = (int)(((r * 8 + c) * 8) + jj);
    }
    bx = __builtin_amdgcn_readfirstlane(bx); j = __builtin_amdgcn_readfirstlane(j); k = __builtin_amdgcn_readfirstlane(k);
}

// assemble_slow_vertical_momentum_tendency! + initialize_stage_perturbations! (acoustic_substepping.jl:727-752,793-838)
// ZERO: the three time-average accumulators are zeroed here (unfused substeps); with the fused substep kernels the first substep of a
// stage assigns them instead (three words per cell and stage written only to be read back once).
// STORE0: first stage of a whole step — the state IS U0, so the perturbations are exact zeros and store_initial_state! rides along
// (the six copies of bzi_compressible_store_initial_state: 12 words per cell; here 6, and the five state reads are shared).
// PF (round 6): the horizontal gradient of the stage's p^L, which the reference's explicit horizontal step re-evaluates in every substep
// (acoustic_substepping.jl:859-876), is folded into the slow tendencies once per stage: Gp_ru = G_ru - dx p^L, Gp_rv = G_rv - dy p^L.
// NOPERT (round 6): the stage's initial perturbations are not stored — the first forward / backward sweep of the stage forms them from
// U0 - U itself (k_ac_forward2<.., INIT>: 1) or knows them to be exact zeros (first stage of a whole step: 2)
template <bool ZERO, bool STORE0, bool PF, bool NOPERT, class ST>
__global__ __launch_bounds__(256) void k_ac_stage_init(DevGrid g, AcFieldsT<ST> F)
{
    int bx, j, k;
    ac_pencil_block(bx, j, k);
    const int i = bx * 256 + threadIdx.x;
    if (i >= g.Nx) return;
    const long long sz = g.Sxy;
    const long long n = g.idx(i, j, k);
    if (j >= g.Ny) {
        // y-slab: the row above the slab.  The north face of the slab's last row reads its folded tendency; G_rv and p of this row arrived
        // with the halo exchanges, and the rank that owns the row forms the same value from the same numbers
        if (PF) F.Gp_rv[n] = F.G_rv[n] - (F.p[n] - F.p[n - g.Sx]) * g.rdy;
        return;
    }
    if (PF) {
        const WrapIdx W = wrap_of(g, i, j);
        const double p0 = F.p[n];
        F.Gp_ru[n] = F.G_ru[n] - (p0 - F.p[n + W.im]) * g.rdx;
        F.Gp_rv[n] = F.G_rv[n] - (p0 - F.p[n + W.jm]) * g.rdy;
    }
    if (STORE0) {
        // U0 - U with U0 := U: (+0) for every finite value, as the subtraction of the stored copy gives
        ((double *)F.U0_rho_d)[n] = F.rho_d[n];
        ((double *)F.U0_rth)[n] = F.rth[n];
        ((double *)F.U0_ru)[n] = F.ru[n];
        ((double *)F.U0_rv)[n] = F.rv[n];
        ((double *)F.U0_rw)[n] = F.rw[n];
        ((double *)F.U0_rq)[n] = F.rq[n];
        if (!NOPERT) { F.rp[n] = 0.0; F.rthp_out[n] = 0.0; F.rup[n] = 0.0; F.rvp[n] = 0.0; F.rwp[n] = 0.0; }
    } else if (!NOPERT) {
    F.rp[n] = F.U0_rho_d[n] - F.rho_d[n];
    F.rthp_out[n] = F.U0_rth[n] - F.rth[n];     // start buffers of the ping-pong fields (set by the launcher)
    F.rup[n] = F.U0_ru[n] - F.ru[n];
    F.rvp[n] = F.U0_rv[n] - F.rv[n];
    F.rwp[n] = F.U0_rw[n] - F.rw[n];
    }
    if (ZERO) {
    F.au[n] = 0.0;
    F.av[n] = 0.0;
    F.aw[n] = 0.0;
    }
    if (k == 0) {
        F.Gs[n] = 0.0;
    } else {
        const long long m = n - sz;
        const double dp = ((F.p[n] - g.p_r[k]) - (F.p[m] - g.p_r[k - 1])) * g.rdzf[k];
        const double rf = ((F.rho[n] - g.rho[k]) + (F.rho[m] - g.rho[k - 1])) / 2.0;
        F.Gs[n] = F.G_rw[n] - dp - g.g * rf;
    }
    if (k == g.Nz - 1) {
        if (STORE0) { ((double *)F.U0_rw)[n + sz] = F.rw[n + sz]; if (!NOPERT) F.rwp[n + sz] = 0.0; }
        else if (!NOPERT) F.rwp[n + sz] = F.U0_rw[n + sz] - F.rw[n + sz];
        F.Gs[n + sz] = 0.0;
    }
}

// Klemp-Skamarock-Ha damping of the previous substep (acoustic_substepping.jl:1123-1139) followed by the explicit
// horizontal step of this substep (:860-881) and the time-average accumulation (:999-1000).
template <bool DAMP, bool STEP, class ST>
__global__ __launch_bounds__(256) void k_ac_horizontal(DevGrid g, AcFieldsT<ST> F, AcParams P)
{
    const int i = blockIdx.x * 256 + threadIdx.x, j = blockIdx.y, k = blockIdx.z;
    if (i >= g.Nx) return;
    const WrapIdx W = wrap_of(g, i, j);
    const long long n = g.idx(i, j, k), mx = n + W.im, my = n + W.jm;
    double up = F.rup[n], vp = F.rvp[n];
    const double rt = F.rthp[n], rtx = F.rthp[mx], rty = F.rthp[my];
    if (DAMP) {
        const double d0 = rt - F.rth_old[n];
        const double ddx = (d0 - (rtx - F.rth_old[mx])) * g.rdx;
        const double ddy = (d0 - (rty - F.rth_old[my])) * g.rdy;
        const double th = F.thL[n];
        up -= P.kdamp * ddx / ((th + F.thL[mx]) / 2.0);
        vp -= P.kdamp * ddy / ((th + F.thL[my]) / 2.0);
    }
    if (STEP) {
        const double p0 = F.p[n];
        double dpx = (p0 - F.p[mx]) * g.rdx;
        double dpy = (p0 - F.p[my]) * g.rdy;
        if (P.gate != 0.0) {
            const double c0 = F.Clin[n] * rt;
            dpx = dpx + P.gate * ((c0 - F.Clin[mx] * rtx) * g.rdx);
            dpy = dpy + P.gate * ((c0 - F.Clin[my] * rty) * g.rdy);
        }
        up += P.dtau * (F.G_ru[n] - dpx);
        vp += P.dtau * (F.G_rv[n] - dpy);
        if (g.bounded_x && i == 0 && P.wall_w) up = 0.0;
        if (g.bounded_y && j == 0 && P.wall_s) vp = 0.0;
        F.au[n] += up;
        F.av[n] += vp;
    }
    if (g.bounded_x && i == 0 && P.wall_w) up = 0.0;
    if (g.bounded_y && j == 0 && P.wall_s) vp = 0.0;
    F.rup[n] = up;
    F.rvp[n] = vp;
}

// boundary-aware centre->face interpolation of theta^L (acoustic_substepping.jl:539-550) from the two cell values
__device__ __forceinline__ double ibz(double fp, double fm, bool p_per, bool m_per)
{
    fp = p_per ? fm : fp;
    fm = m_per ? fp : fm;
    return (fp + fm) / 2.0;
}

// one face of k_ac_horizontal: cells a (low side) and b (high side) of the face, rd = 1/spacing
template <bool DAMP>
__device__ __forceinline__ double ac_face_update(double up, double G, double rt_b, double rt_a, double rto_b, double rto_a,
                                                 double th_b, double th_a, double C_b, double C_a, double p_b, double p_a,
                                                 double rd, const AcParams &P)
{
    if (DAMP) {
        const double dd = ((rt_b - rto_b) - (rt_a - rto_a)) * rd;
        up -= P.kdamp * dd / ((th_b + th_a) / 2.0);
    }
    double dp = (p_b - p_a) * rd;
    if (P.gate != 0.0) dp = dp + P.gate * ((C_b * rt_b - C_a * rt_a) * rd);
    return up + P.dtau * (G - dp);
}

// block shapes of the column kernels (measured on MI355X, 512x512x256: forward sweep 64x4 99 ms/step vs 256x1 102;
// backward sweep 256x1 32.5 vs 64x4 37.8)
#ifndef ACX
#define ACX 64
#endif
#ifndef ACY
#define ACY 4
#endif
#ifndef ABX
#define ABX 256
#endif
#ifndef ABY
#define ABY 1
#endif
#ifndef AC_MINW
#define AC_MINW 1
#endif
// _build_predictors! + _build_vertical_rhs! + forward sweep of the BatchedTridiagonalSolver
// (acoustic_substepping.jl:605-659,907-970).  FUSED: the horizontal step (k_ac_horizontal) of the four faces of the
// column is evaluated in place of loading (rho u)', (rho v)': each face is computed by its two adjacent columns with
// identical arithmetic and stored by its owner, which removes one read + one write of (rho u)', (rho v)' and the separate
// pass over (rho theta)', theta^L, C per substep.
template <bool FIRST, bool FUSED, bool DAMP, class ST>
__global__ __launch_bounds__(ACX * ACY, AC_MINW) void k_ac_column_forward(DevGrid g, AcFieldsT<ST> F, AcParams P)
{
    // Workgroups go round-robin to the 8 XCDs in launch order (x fastest), so in launch order XCD c owns the tile COLUMN bx = c (mod 8): the
    // x neighbours of every tile live behind another XCD's L2, and the one value a row's edge lane needs of them (eight arrays) costs a
    // 128-byte line through the fabric each.  With P.xcd XCD c owns the BAND of tile rows [c gy/8, (c+1) gy/8) instead and walks it x fastest:
    // both x neighbours and (but at the band edges) both y neighbours are tiles of the same L2.
    int bx = blockIdx.x, by = blockIdx.y;
    if (P.xcd) {
        const unsigned w = blockIdx.y * gridDim.x + blockIdx.x, c = w & 7u, r = w >> 3;
        bx = (int)(r % gridDim.x);
        by = (int)(c * (gridDim.y >> 3) + r / gridDim.x);
    }
    const int i = bx * ACX + threadIdx.x, j = by * ACY + threadIdx.y;
    if (i >= g.Nx || j >= g.Ny) return;
    const WrapIdx W = wrap_of(g, i, j);
    const long long sz = g.Sxy;
    const int Nz = g.Nz;
    long long n = g.idx(i, j, 0);
    const double dtn2 = P.dtn * P.dtn;

    // rings: theta^L at k, k+1 (own column) and at faces k-1, k; C at k-1, k; old (rho w)' at faces k-1, k, k+1
    double th_0 = F.thL[n], th_p = F.thL[n + sz];
    double C_m = 0.0, C_0 = F.Clin[n];
    // Neighbours of the two linearisation fields (round 4).  The own-column values ride rings that are loaded one (C) or two (theta_L)
    // levels ahead; the neighbours used to be loaded at the level that needs them — the same cache lines, one or two level fronts of
    // the whole grid later (47 KB per block and level x 1024 resident blocks: long evicted), so every line of theta_L and C crossed
    // the fabric twice.  Now the x neighbours come from the neighbouring lanes' ring values (a row of the block is one wavefront; only
    // its two edge lanes load), and the y neighbours are requested together with the own-column value of their level and carried.
#ifndef AC_FX
#define AC_FX 1
#endif
    const bool FX = AC_FX && FUSED && (ACX == 64) && sizeof(ST) == 8;      // (Float32 working fields: measured slower — 74 -> 101 ms per step of forward sweeps — and stay on the loads)
    const int lane = threadIdx.x;
    double thy_m0 = 0.0, thy_p0 = 0.0, thy_m1 = 0.0, thy_p1 = 0.0;      // theta_L of rows j-1 / j+1 at levels k and k+1
    double cy_m0 = 0.0, cy_p0 = 0.0;                                    // C of rows j-1 / j+1 at level k
    if (FX) {
        thy_m0 = F.thL[n + W.jm]; thy_p0 = F.thL[n + W.jp];
        thy_m1 = F.thL[n + sz + W.jm]; thy_p1 = F.thL[n + sz + W.jp];
        cy_m0 = F.Clin[n + W.jm]; cy_p0 = F.Clin[n + W.jp];
    }
    double w_m = 0.0, w_0 = F.rwp[n], w_p = F.rwp[n + sz];
    double rs_m = 0.0, rths_m = 0.0, rp_m = 0.0, rthp_m = 0.0;
    double beta = 1.0, phi_m = 0.0, c_m = 0.0;     // row 0: b = 1, c = 0, f = 0
    double thf_0 = th_0, thf_m = th_0;               // theta at faces k (k = 0: one-sided) and k-1

    for (int k = 0; k < Nz; ++k, n += sz) {
        const double rdc = g.rdzc[k];
        const double Ax = g.Ax[k], Ay = g.Ay[k], Vinv = g.Vinv_c[k];
        const double rp = F.rp[n], rthp = F.rthp[n];
        double thxm, thxp, thym, thyp;
        if (FX) {
            // lanes 0 and 63 of the row hold the block's edge columns: their outer neighbours belong to another block (or wrap around)
            const double e_m = (lane == 0) ? F.thL[n + W.im] : 0.0, e_p = (lane == ACX - 1 || i == g.Nx - 1) ? F.thL[n + W.ip] : 0.0;
            thxm = __shfl_up(th_0, 1); thxp = __shfl_down(th_0, 1);
            if (lane == 0) thxm = e_m;
            if (lane == ACX - 1 || i == g.Nx - 1) thxp = e_p;
            thym = thy_m0; thyp = thy_p0;
        } else {
            thxm = F.thL[n + W.im]; thxp = F.thL[n + W.ip]; thym = F.thL[n + W.jm]; thyp = F.thL[n + W.jp];
        }
        double up0, up1, vp0, vp1;
        if (FUSED) {
            const long long nxm = n + W.im, nxp = n + W.ip, nym = n + W.jm, nyp = n + W.jp;
            const double rt_xm = F.rthp[nxm], rt_xp = F.rthp[nxp], rt_ym = F.rthp[nym], rt_yp = F.rthp[nyp];
            double o0 = 0.0, o_xm = 0.0, o_xp = 0.0, o_ym = 0.0, o_yp = 0.0;
            if (DAMP) { o0 = F.rth_old[n]; o_xm = F.rth_old[nxm]; o_xp = F.rth_old[nxp]; o_ym = F.rth_old[nym]; o_yp = F.rth_old[nyp]; }
            double c_xm = 0.0, c_xp = 0.0, c_ym = 0.0, c_yp = 0.0;
            if (FX) {
                const double e_m = (lane == 0) ? F.Clin[nxm] : 0.0, e_p = (lane == ACX - 1 || i == g.Nx - 1) ? F.Clin[nxp] : 0.0;
                c_xm = __shfl_up(C_0, 1); c_xp = __shfl_down(C_0, 1);
                if (lane == 0) c_xm = e_m;
                if (lane == ACX - 1 || i == g.Nx - 1) c_xp = e_p;
                c_ym = cy_m0; c_yp = cy_p0;
            } else if (P.gate != 0.0) { c_xm = F.Clin[nxm]; c_xp = F.Clin[nxp]; c_ym = F.Clin[nym]; c_yp = F.Clin[nyp]; }
            const double p0 = F.p[n], p_xm = F.p[nxm], p_xp = F.p[nxp], p_ym = F.p[nym], p_yp = F.p[nyp];
            up0 = ac_face_update<DAMP>(F.rup_in[n], F.G_ru[n], rthp, rt_xm, o0, o_xm, th_0, thxm, C_0, c_xm, p0, p_xm, g.rdx, P);
            up1 = ac_face_update<DAMP>(F.rup_in[nxp], F.G_ru[nxp], rt_xp, rthp, o_xp, o0, thxp, th_0, c_xp, C_0, p_xp, p0, g.rdx, P);
            vp0 = ac_face_update<DAMP>(F.rvp_in[n], F.G_rv[n], rthp, rt_ym, o0, o_ym, th_0, thym, C_0, c_ym, p0, p_ym, g.rdy, P);
            vp1 = ac_face_update<DAMP>(F.rvp_in[nyp], F.G_rv[nyp], rt_yp, rthp, o_yp, o0, thyp, th_0, c_yp, C_0, p_yp, p0, g.rdy, P);
            ac_wall_faces(g, P, i, j, up0, up1, vp0, vp1);
            F.rup[n] = up0;
            F.rvp[n] = vp0;
            if (FIRST) { F.au[n] = 0.0 + up0; F.av[n] = 0.0 + vp0; }      // first substep of the stage: the accumulators start here (0 + x keeps the bits of the zeroed array)
            else { F.au[n] += up0; F.av[n] += vp0; }
        } else {
            up0 = F.rup[n]; up1 = F.rup[n + W.ip]; vp0 = F.rvp[n]; vp1 = F.rvp[n + W.jp];
            ac_wall_faces(g, P, i, j, up0, up1, vp0, vp1);
            F.rth_old[n] = rthp;
        }
        // theta face k+1 (top face: one-sided)
        const double thf_p = (k + 1 < Nz) ? (th_p + th_0) / 2.0 : th_0;

        const double dxM = Ax * up1 - Ax * up0;
        const double dxT = Ax * ((thxp + th_0) / 2.0) * up1 - Ax * ((th_0 + thxm) / 2.0) * up0;
        const double dyM = Ay * vp1 - Ay * vp0;
        const double dyT = Ay * ((thyp + th_0) / 2.0) * vp1 - Ay * ((th_0 + thym) / 2.0) * vp0;
        const double divM = Vinv * (dxM + dyM);
        const double divT = Vinv * (dxT + dyT);
        const double dzW = (w_p - w_0) * rdc;
        const double dzT = (thf_p * w_p - thf_0 * w_0) * rdc;
        const double rs = rp + P.dtau * (F.G_rho_d[n] - divM) - P.dto * dzW;
        const double rths = rthp + P.dtau * (P.f_theta * F.G_rth[n] - divT) - P.dto * dzT;
        F.rs[n] = rs;
        F.rths[n] = rths;

        double phi = 0.0;
        if (k > 0) {
            const double rdf = g.rdzf[k], rdm = g.rdzc[k - 1];
            // right-hand side at face k
            const double dps = (C_0 * rths - C_m * rths_m) * rdf;
            const double dpo = (C_0 * rthp - C_m * rthp_m) * rdf;
            const double Gp = P.dto * dpo + P.dtn * dps;
            const double Gb = g.g * (P.dto * ((rp + rp_m) / 2.0) + P.dtn * ((rs + rs_m) / 2.0));
            const double d2 = ((w_p - w_0) * rdc - (w_0 - w_m) * rdm) * rdf;
            const double Gd = -P.d_old * d2;
            const double sp = F.sponge[k];                // sponge_rhs / sponge_term_diag (acoustic_substepping.jl:591-602)
            const double f = w_0 + P.dtau * P.f_w * F.Gs[n] - Gp - Gb - Gd - fabs(P.dto) * sp * w_0;
            // coefficients of row k
            const double a = -dtn2 * C_m * thf_m * rdm * rdf + dtn2 * g.g * rdm / 2.0 + (-P.d_new * rdm * rdf);
            const double b = 1.0 + (dtn2 * thf_0 * (C_0 * rdc + C_m * rdm) * rdf + dtn2 * g.g * (rdc - rdm) / 2.0 +
                                    P.d_new * (rdc + rdm) * rdf + fabs(P.dtn) * sp);
            const double t = c_m / beta;
            beta = b - a * t;
            phi = (f - a * phi_m) / beta;
            if (FIRST) F.tfac[n] = t;
            // upper coefficient of this row, used by the next one
            c_m = -dtn2 * C_0 * thf_p * rdc * rdf + (-dtn2 * g.g * rdc / 2.0) + (-P.d_new * rdc * rdf);
        } else if (FIRST) {
            F.tfac[n] = 0.0;
        }
        F.phi[n] = phi;
        phi_m = phi;

        // advance the rings
        rs_m = rs; rths_m = rths; rp_m = rp; rthp_m = rthp;
        C_m = C_0;
        th_0 = th_p; thf_m = thf_0; thf_0 = thf_p;
        w_m = w_0; w_0 = w_p;
        if (k + 1 < Nz) {
            C_0 = F.Clin[n + sz];
            th_p = (k + 2 < Nz) ? F.thL[n + 2 * sz] : th_0;
            w_p = F.rwp[n + 2 * sz];
            if (FX) {      // the y neighbours of the levels just requested, with them
                cy_m0 = F.Clin[n + sz + W.jm]; cy_p0 = F.Clin[n + sz + W.jp];
                thy_m0 = thy_m1; thy_p0 = thy_p1;
                if (k + 2 < Nz) { thy_m1 = F.thL[n + 2 * sz + W.jm]; thy_p1 = F.thL[n + 2 * sz + W.jp]; }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// Round 6: the fused forward sweep written for the memory pipe.  The kernel above spends 77 % of its wave cycles parked
// (profiles/r05_pmc_compressible.json) for two reasons its ISA shows:
//   * its stores (rup, rvp, au, av, rs, rths, phi) cannot be proven not to alias the loads that follow them in program order (au / av are
//     read-modify-write, the ring loads of the next level sit below the stores), so hipcc keeps that order and — the vector-memory counter
//     returns in order — drains everything outstanding four times per level (`s_waitcnt vmcnt(0)`; store, one load, drain, store, ...):
//     four to five memory round trips per level instead of one;
//   * every `F.x[n + d]` is a 64-bit per-lane address: 45 base pairs = 90 of its 214 VGPRs hold addresses (two waves per SIMD).
// Here every load of a level — the level's own words, the neighbours' and the ring words of the levels above — is issued before the
// first store of the level, all stores close the level, and every access is `uniform base + 32-bit per-lane byte offset` (the saddr form
// of global_load: one offset register per neighbour displacement).  The arithmetic is the text of the kernel above, so the results carry
// its bits; PF folds the stage-constant horizontal gradient of p^L into the slow tendencies (Gp_ru, Gp_rv written by k_ac_stage_init:
// one array and four neighbour loads less per level and substep; the only deviation, one rounding of G - dp).
// Neighbours in x of theta_L, C, (rho theta)' and its previous value come from the neighbouring lanes; the two edge lanes of a row fetch
// theirs with one predicated load per array.
// ---------------------------------------------------------------------------------------------------------------------
template <class T>
__device__ __forceinline__ T ac_ld(const T *base, unsigned e)
{
    return *(const T *)((const char *)base + (size_t)(e * (unsigned)sizeof(T)));
}
template <class T, class V>
__device__ __forceinline__ void ac_st(T *base, unsigned e, V v)
{
    *(T *)((char *)base + (size_t)(e * (unsigned)sizeof(T))) = (T)v;
}
// the same for words no other column reads (or that are read once): non-temporal, so that they do not displace the neighbour-shared
// lines (theta_L, C, (rho theta)', the momentum perturbations and their tendencies) from the L2 before the neighbour row asks for them
#ifndef AC2_NT
#define AC2_NT 0
#endif
#ifndef AC2_BARRIER
#define AC2_BARRIER 0
#endif
template <class T>
__device__ __forceinline__ T ac_ld_nt(const T *base, unsigned e)
{
    if (AC2_NT & 1) return __builtin_nontemporal_load((const T *)((const char *)base + (size_t)(e * (unsigned)sizeof(T))));
    return ac_ld(base, e);
}
template <class T, class V>
__device__ __forceinline__ void ac_st_nt(T *base, unsigned e, V v)
{
    if (AC2_NT & 2) __builtin_nontemporal_store((T)v, (T *)((char *)base + (size_t)(e * (unsigned)sizeof(T))));
    else ac_st(base, e, v);
}

template <bool DAMP, bool PF>
__device__ __forceinline__ double ac_face_update2(double up, double G, double rt_b, double rt_a, double rto_b, double rto_a,
                                                  double th_b, double th_a, double C_b, double C_a, double p_b, double p_a,
                                                  double rd, const AcParams &P)
{
    if (DAMP) {
        const double dd = ((rt_b - rto_b) - (rt_a - rto_a)) * rd;
        up -= P.kdamp * dd / ((th_b + th_a) / 2.0);
    }
    if (PF) {      // G already carries -(p_b - p_a) rd
        if (P.gate != 0.0) G = G - P.gate * ((C_b * rt_b - C_a * rt_a) * rd);
        return up + P.dtau * G;
    }
    double dp = (p_b - p_a) * rd;
    if (P.gate != 0.0) dp = dp + P.gate * ((C_b * rt_b - C_a * rt_a) * rd);
    return up + P.dtau * (G - dp);
}

// CFG: bit 0: 512 threads per block instead of 256; bit 1: register budget for three waves per SIMD instead of two; bit 2: the rows of a
// block advance level by level together (one s_barrier per level); bit 3: the loads of level k + 1 are issued before level k is worked on;
// bit 4: x neighbours by DPP wavefront shifts instead of ds_bpermute
// INIT (first substep of a stage only): 0 the stage's initial perturbations are read from their arrays (k_ac_stage_init stored them);
// 1 they are formed here from U0 - U (rounded through the storage type, as the stored ones are) — the initialisation pass neither writes
// them (5 words per cell) nor does this sweep read them back (5): it reads U0 and U (10) and stores the initial (rho theta)' (1), which
// the next substep's damping and the stage epilogue read; 2 first stage of a whole step: the state IS U0, the perturbations are exact zeros
template <bool FIRST, bool DAMP, bool PF, int CFG, class ST, int INIT = 0>
__global__ __launch_bounds__((CFG & 1) ? 512 : 256, (CFG & 2) ? 3 : 2) void k_ac_forward2(DevGrid g, AcFieldsT<ST> F, AcParams P)
{
    static_assert(FIRST || INIT == 0, "initial perturbations belong to the first substep of a stage");
    int bx = blockIdx.x, by = blockIdx.y;
    if (P.xcd) {      // XCD c owns the band of tile rows [c gy/8, (c+1) gy/8) (see k_ac_column_forward)
        const unsigned w = blockIdx.y * gridDim.x + blockIdx.x, c = w & 7u, r = w >> 3;
        bx = (int)(r % gridDim.x);
        by = (int)(c * (gridDim.y >> 3) + r / gridDim.x);
    }
    // 256 threads as 64 x 4, 128 x 2 or 256 x 1 columns (BZ_AC_BX): a wavefront is always 64 consecutive cells of one row
    const int i = bx * (int)blockDim.x + threadIdx.x, j = by * (int)blockDim.y + threadIdx.y;
    if (i >= g.Nx || j >= g.Ny) return;
    const WrapIdx W = wrap_of(g, i, j);
    const unsigned sz = (unsigned)g.Sxy;
    const unsigned dxp = (unsigned)(int)W.ip, dym = (unsigned)(int)W.jm, dyp = (unsigned)(int)W.jp;
    const int Nz = g.Nz;
    unsigned e = (unsigned)g.idx(i, j, 0);
    const double dtn2 = P.dtn * P.dtn;
    const int lane = threadIdx.x & 63;
    const bool edge_m = (lane == 0), edge_p = (lane == 63 || i == g.Nx - 1);
    const bool edge = edge_m || edge_p;
    const unsigned dedge = edge_m ? (unsigned)(int)W.im : dxp;
    const ColPtr::cptr sponge = (ColPtr::cptr)F.sponge;
    const double *Gu = PF ? F.Gp_ru : F.G_ru, *Gv = PF ? F.Gp_rv : F.G_rv;
    const bool acc = ac_accumulate(P);      // uniform: the time-average accumulators of this stage are wanted

    // rings (see k_ac_column_forward): theta_L at k, k+1 of the own column and of rows j-1 / j+1, C at k-1, k (rows j-1 / j+1: k),
    // old (rho w)' at faces k-1, k, k+1
    double th_0 = ac_ld(F.thL, e), th_p = ac_ld(F.thL, e + sz);
    double C_m = 0.0, C_0 = ac_ld(F.Clin, e);
    double thy_m0 = ac_ld(F.thL, e + dym), thy_p0 = ac_ld(F.thL, e + dyp);
    double thy_m1 = ac_ld(F.thL, e + sz + dym), thy_p1 = ac_ld(F.thL, e + sz + dyp);
    double cy_m0 = ac_ld(F.Clin, e + dym), cy_p0 = ac_ld(F.Clin, e + dyp);
    // initial perturbation of a prognostic field at element ee, as k_ac_stage_init stores it (the working fields in the storage type)
    auto pert = [&](const double *U0, const double *U, unsigned ee) { return (double)(ST)(ac_ld(U0, ee) - ac_ld(U, ee)); };
    auto pert_w = [&](unsigned ee) { return ac_ld(F.U0_rw, ee) - ac_ld((const double *)F.rw, ee); };      // (rho w)' stays in the grid's type
    double w_m = 0.0, w_0, w_p;
    if (INIT == 2) { w_0 = 0.0; w_p = 0.0; }
    else if (INIT == 1) { w_0 = pert_w(e); w_p = pert_w(e + sz); }
    else { w_0 = ac_ld(F.rwp, e); w_p = ac_ld(F.rwp, e + sz); }
    double rs_m = 0.0, rths_m = 0.0, rp_m = 0.0, rthp_m = 0.0;
    double beta = 1.0, phi_m = 0.0, c_m = 0.0;     // row 0: b = 1, c = 0, f = 0
    double thf_0 = th_0, thf_m = th_0;               // theta at faces k (k = 0: one-sided) and k-1

    // every word a level asks the memory for, as one record: `level(e, k)` issues the loads (own level k, the neighbours', the ring words of the
    // levels above); with CFG bit 3 the record of level k + 1 is requested before level k is worked on (its loads fly under the ~280 VALU
    // instructions, the stores and the barrier of level k: the rows of a block that advance in lock step would otherwise leave the memory
    // pipe idle between a level's stores and the next level's loads)
    struct AcLevel {
        double rp, rthp, rt_ym, rt_yp, o0, o_ym, o_yp, ru0, ru1, rv0, rv1, Gu0, Gu1, Gv0, Gv1, p0, p_xm, p_xp, p_ym, p_yp, au_o, av_o, Grho, Grth, Gs_k;
        double e_th, e_C, e_rt, e_o;      // the row's outer neighbour: lane 0 the left one, the last lane the right one
        double C_n, cy_mn, cy_pn, th_n, thy_mn, thy_pn, w_n;
    };
    auto level = [&](const unsigned e, const int k) {
        AcLevel L;
        const unsigned exp_ = e + dxp, eym = e + dym, eyp = e + dyp;
        if (INIT == 2) { L.rp = 0.0; L.rthp = 0.0; L.rt_ym = 0.0; L.rt_yp = 0.0; }
        else if (INIT == 1) {
            L.rp = pert(F.U0_rho_d, F.rho_d, e); L.rthp = pert(F.U0_rth, F.rth, e);
            L.rt_ym = pert(F.U0_rth, F.rth, eym); L.rt_yp = pert(F.U0_rth, F.rth, eyp);
        } else {
        L.rp = ac_ld_nt(F.rp, e); L.rthp = ac_ld(F.rthp, e);
        L.rt_ym = ac_ld(F.rthp, eym); L.rt_yp = ac_ld(F.rthp, eyp);
        }
        L.o0 = 0.0; L.o_ym = 0.0; L.o_yp = 0.0;
        if (DAMP) { L.o0 = ac_ld(F.rth_old, e); L.o_ym = ac_ld(F.rth_old, eym); L.o_yp = ac_ld(F.rth_old, eyp); }
        if (INIT == 2) { L.ru0 = 0.0; L.ru1 = 0.0; L.rv0 = 0.0; L.rv1 = 0.0; }
        else if (INIT == 1) {
            L.ru0 = pert(F.U0_ru, F.ru, e); L.ru1 = pert(F.U0_ru, F.ru, exp_); L.rv0 = pert(F.U0_rv, F.rv, e); L.rv1 = pert(F.U0_rv, F.rv, eyp);
        } else {
        L.ru0 = ac_ld(F.rup_in, e); L.ru1 = ac_ld(F.rup_in, exp_); L.rv0 = ac_ld(F.rvp_in, e); L.rv1 = ac_ld(F.rvp_in, eyp);
        }
        L.Gu0 = ac_ld(Gu, e); L.Gu1 = ac_ld(Gu, exp_); L.Gv0 = ac_ld(Gv, e); L.Gv1 = ac_ld(Gv, eyp);
        L.p0 = 0.0; L.p_xm = 0.0; L.p_xp = 0.0; L.p_ym = 0.0; L.p_yp = 0.0;
        if (!PF) {
            L.p0 = ac_ld(F.p, e); L.p_xm = ac_ld(F.p, e + (unsigned)(int)W.im); L.p_xp = ac_ld(F.p, exp_);
            L.p_ym = ac_ld(F.p, eym); L.p_yp = ac_ld(F.p, eyp);
        }
        L.au_o = 0.0; L.av_o = 0.0;
        if (!FIRST && acc && P.acc_mode != 1) { L.au_o = ac_ld_nt(F.au, e); L.av_o = ac_ld_nt(F.av, e); }
        L.Grho = ac_ld_nt(F.G_rho_d, e); L.Grth = ac_ld_nt(F.G_rth, e); L.Gs_k = ac_ld_nt(F.Gs, e);
        L.e_th = 0.0; L.e_C = 0.0; L.e_rt = 0.0; L.e_o = 0.0;
        if (edge) {
            const unsigned ee = e + dedge;
            L.e_th = ac_ld(F.thL, ee); L.e_C = ac_ld(F.Clin, ee);
            L.e_rt = INIT == 2 ? 0.0 : INIT == 1 ? pert(F.U0_rth, F.rth, ee) : ac_ld(F.rthp, ee);
            if (DAMP) L.e_o = ac_ld(F.rth_old, ee);
        }
        // ring words of the levels above (the last levels re-read an in-range level; those values are never used)
        const unsigned e1 = (k + 1 < Nz) ? e + sz : e, e2 = (k + 2 < Nz) ? e1 + sz : e1;
        L.C_n = ac_ld(F.Clin, e1); L.cy_mn = ac_ld(F.Clin, e1 + dym); L.cy_pn = ac_ld(F.Clin, e1 + dyp);
        L.th_n = ac_ld(F.thL, e2); L.thy_mn = ac_ld(F.thL, e2 + dym); L.thy_pn = ac_ld(F.thL, e2 + dyp);
        L.w_n = INIT == 2 ? 0.0 : INIT == 1 ? pert_w(e1 + sz) : ac_ld_nt(F.rwp, e1 + sz);
        return L;
    };
    constexpr bool PIPE = (CFG & 8) != 0;
    auto work = [&](const AcLevel &L, const int k, const unsigned e) {
        const double rdc = g.rdzc[k];
        const double Ax = g.Ax[k], Ay = g.Ay[k], Vinv = g.Vinv_c[k];
        const double rp = L.rp, rthp = L.rthp, rt_ym = L.rt_ym, rt_yp = L.rt_yp, o0 = L.o0, o_ym = L.o_ym, o_yp = L.o_yp;
        const double ru0 = L.ru0, ru1 = L.ru1, rv0 = L.rv0, rv1 = L.rv1, Gu0 = L.Gu0, Gu1 = L.Gu1, Gv0 = L.Gv0, Gv1 = L.Gv1;
        const double p0 = L.p0, p_xm = L.p_xm, p_xp = L.p_xp, p_ym = L.p_ym, p_yp = L.p_yp, au_o = L.au_o, av_o = L.av_o;
        const double Grho = L.Grho, Grth = L.Grth, Gs_k = L.Gs_k, e_th = L.e_th, e_C = L.e_C, e_rt = L.e_rt, e_o = L.e_o;
        const double C_n = L.C_n, cy_mn = L.cy_mn, cy_pn = L.cy_pn, th_n = L.th_n, thy_mn = L.thy_mn, thy_pn = L.thy_pn, w_n = L.w_n;
        const double sp = sponge[k];                // sponge_rhs / sponge_term_diag (acoustic_substepping.jl:591-602)

        // ---- neighbours in x from the neighbouring lanes ----
        constexpr bool DPP = (CFG & 16) != 0;
        double thxm = ac_lane_up<DPP>(th_0), thxp = ac_lane_down<DPP>(th_0);
        double c_xm = ac_lane_up<DPP>(C_0), c_xp = ac_lane_down<DPP>(C_0);
        double rt_xm = ac_lane_up<DPP>(rthp), rt_xp = ac_lane_down<DPP>(rthp);
        double o_xm = 0.0, o_xp = 0.0;
        if (DAMP) { o_xm = ac_lane_up<DPP>(o0); o_xp = ac_lane_down<DPP>(o0); }
        if (edge_m) { thxm = e_th; c_xm = e_C; rt_xm = e_rt; o_xm = e_o; }
        if (edge_p) { thxp = e_th; c_xp = e_C; rt_xp = e_rt; o_xp = e_o; }
        const double thym = thy_m0, thyp = thy_p0, c_ym = cy_m0, c_yp = cy_p0;

        const double up0 = ac_face_update2<DAMP, PF>(ru0, Gu0, rthp, rt_xm, o0, o_xm, th_0, thxm, C_0, c_xm, p0, p_xm, g.rdx, P);
        const double up1 = ac_face_update2<DAMP, PF>(ru1, Gu1, rt_xp, rthp, o_xp, o0, thxp, th_0, c_xp, C_0, p_xp, p0, g.rdx, P);
        const double vp0 = ac_face_update2<DAMP, PF>(rv0, Gv0, rthp, rt_ym, o0, o_ym, th_0, thym, C_0, c_ym, p0, p_ym, g.rdy, P);
        const double vp1 = ac_face_update2<DAMP, PF>(rv1, Gv1, rt_yp, rthp, o_yp, o0, thyp, th_0, c_yp, C_0, p_yp, p0, g.rdy, P);
        const double au_n = FIRST ? 0.0 + up0 : (P.acc_mode == 2) ? au_o + (ru0 + up0) : au_o + up0;
        const double av_n = FIRST ? 0.0 + vp0 : (P.acc_mode == 2) ? av_o + (rv0 + vp0) : av_o + vp0;
        // theta face k+1 (top face: one-sided)
        const double thf_p = (k + 1 < Nz) ? (th_p + th_0) / 2.0 : th_0;

        const double dxM = Ax * up1 - Ax * up0;
        const double dxT = Ax * ((thxp + th_0) / 2.0) * up1 - Ax * ((th_0 + thxm) / 2.0) * up0;
        const double dyM = Ay * vp1 - Ay * vp0;
        const double dyT = Ay * ((thyp + th_0) / 2.0) * vp1 - Ay * ((th_0 + thym) / 2.0) * vp0;
        const double divM = Vinv * (dxM + dyM);
        const double divT = Vinv * (dxT + dyT);
        const double dzW = (w_p - w_0) * rdc;
        const double dzT = (thf_p * w_p - thf_0 * w_0) * rdc;
        const double rs = rp + P.dtau * (Grho - divM) - P.dto * dzW;
        const double rths = rthp + P.dtau * (P.f_theta * Grth - divT) - P.dto * dzT;

        double phi = 0.0, t = 0.0;
        if (k > 0) {
            const double rdf = g.rdzf[k], rdm = g.rdzc[k - 1];
            // right-hand side at face k
            const double dps = (C_0 * rths - C_m * rths_m) * rdf;
            const double dpo = (C_0 * rthp - C_m * rthp_m) * rdf;
            const double Gp = P.dto * dpo + P.dtn * dps;
            const double Gb = g.g * (P.dto * ((rp + rp_m) / 2.0) + P.dtn * ((rs + rs_m) / 2.0));
            const double d2 = ((w_p - w_0) * rdc - (w_0 - w_m) * rdm) * rdf;
            const double Gd = -P.d_old * d2;
            const double f = w_0 + P.dtau * P.f_w * Gs_k - Gp - Gb - Gd - fabs(P.dto) * sp * w_0;
            // coefficients of row k
            const double a = -dtn2 * C_m * thf_m * rdm * rdf + dtn2 * g.g * rdm / 2.0 + (-P.d_new * rdm * rdf);
            const double b = 1.0 + (dtn2 * thf_0 * (C_0 * rdc + C_m * rdm) * rdf + dtn2 * g.g * (rdc - rdm) / 2.0 +
                                    P.d_new * (rdc + rdm) * rdf + fabs(P.dtn) * sp);
            t = c_m / beta;
            beta = b - a * t;
            phi = (f - a * phi_m) / beta;
            // upper coefficient of this row, used by the next one
            c_m = -dtn2 * C_0 * thf_p * rdc * rdf + (-dtn2 * g.g * rdc / 2.0) + (-P.d_new * rdc * rdf);
        }
        // ---- every store of the level ----
        if (INIT) ac_st(F.rthp, e, rthp);      // the initial (rho theta)': the next substep's damping and the stage epilogue read it
        if (INIT && !g.wrap_y) {               // y-slab: and its halo rows (the exchange before this sweep carried a buffer nobody had filled)
            if (j == 0) ac_st(F.rthp, e + dym, rt_ym);
            if (j == g.Ny - 1) ac_st(F.rthp, e + dyp, rt_yp);
        }
        ac_st_nt(F.rup, e, up0);
        ac_st_nt(F.rvp, e, vp0);
        if (acc && (FIRST || P.acc_mode != 1)) {
            ac_st_nt(F.au, e, au_n);
            ac_st_nt(F.av, e, av_n);
        }
        ac_st_nt(F.rs, e, rs);
        ac_st_nt(F.rths, e, rths);
        if (FIRST) ac_st_nt(F.tfac, e, t);
        ac_st_nt(F.phi, e, phi);
        phi_m = phi;
        if (AC2_BARRIER || (CFG & 4)) __builtin_amdgcn_s_barrier();      // the rows of a block advance together: a row's y neighbours are requested while their lines are near

        // advance the rings
        rs_m = rs; rths_m = rths; rp_m = rp; rthp_m = rthp;
        C_m = C_0; C_0 = C_n;
        thf_m = thf_0; thf_0 = thf_p;
        th_0 = th_p; th_p = th_n;
        w_m = w_0; w_0 = w_p; w_p = w_n;
        cy_m0 = cy_mn; cy_p0 = cy_pn;
        thy_m0 = thy_m1; thy_p0 = thy_p1; thy_m1 = thy_mn; thy_p1 = thy_pn;
    };
    if (!PIPE) {
        for (int k = 0; k < Nz; ++k, e += sz) work(level(e, k), k, e);
    } else {
        // two levels per trip so that the two records live in fixed registers (a rotation `current = next` would have to wait for the
        // next level's loads before it could copy them); the last level re-reads itself
        AcLevel LA = level(e, 0), LB;
#pragma unroll 1
        for (int k = 0; k < Nz; k += 2, e += 2 * sz) {
            const int k1 = (k + 1 < Nz) ? k + 1 : k;
            LB = level(e + (unsigned)(k1 - k) * sz, k1);
            work(LA, k, e);
            if (k + 1 < Nz) {
                const int k2 = (k + 2 < Nz) ? k + 2 : k + 1;
                LA = level(e + (unsigned)(k2 - k) * sz, k2);
                work(LB, k + 1, e + sz);
            }
        }
    }
}

// back substitution + _post_solve_recovery! (acoustic_substepping.jl:993-1002)
// INIT (FIRST only; see k_ac_forward2): 1 / 2 — the initial (rho w)' of the held top face is formed here (U0 - U / zero) and stored
template <bool FIRST, class ST, int INIT = 0>      // FIRST: first substep of a stage with the fused forward sweep: <rho w> starts here
__global__ __launch_bounds__(ABX * ABY) void k_ac_column_backward(DevGrid g, AcFieldsT<ST> F, AcParams P)
{
    const int i = blockIdx.x * ABX + threadIdx.x, j = blockIdx.y * ABY + threadIdx.y;
    if (i >= g.Nx || j >= g.Ny) return;
    const long long sz = g.Sxy;
    const int Nz = g.Nz;
    const bool acc = ac_accumulate(P);
    long long n = g.idx(i, j, Nz - 1);
    double w_hi = F.rwp[n + sz];                 // top face: held at its (zero) rewind value
    if (INIT) {
        w_hi = (INIT == 2) ? 0.0 : F.U0_rw[n + sz] - F.rw[n + sz];
        F.rwp[n + sz] = w_hi;
    }
    double th_0 = F.thL[n];                      // theta at cell k
    double thf_hi = th_0;                        // face Nz: one-sided
    double t_hi = 0.0;                           // t_{k+1}
    for (int k = Nz - 1; k >= 0; --k, n -= sz) {
        const double th_m = (k > 0) ? F.thL[n - sz] : th_0;
        const double thf_lo = (k > 0) ? (th_0 + th_m) / 2.0 : th_0;
        double w_lo = F.phi[n];
        if (k < Nz - 1) w_lo -= t_hi * w_hi;
        const double rdc = g.rdzc[k];
        const double dzW = (w_hi - w_lo) * rdc;
        const double dzT = (thf_hi * w_hi - thf_lo * w_lo) * rdc;
        F.rp[n] = F.rs[n] - P.dtn * dzW;
        F.rthp_out[n] = F.rths[n] - P.dtn * dzT;
        // <w> two substeps at a time (AcParams::acc_mode): the second substep of a pair adds (w'_{n-1} + w'_n) with w'_{n-1} read from the field
        // it is about to overwrite (one word instead of the accumulator's two in the pair's first substep)
        const double w_old = (!FIRST && acc && P.acc_mode == 2) ? F.rwp[n] : 0.0;
        F.rwp[n] = w_lo;
        if (acc) {
            if (FIRST) F.aw[n] = 0.0 + w_lo;
            else if (P.acc_mode == 2) F.aw[n] += (w_old + w_lo);
            else if (P.acc_mode == 0) F.aw[n] += w_lo;
        }
        t_hi = F.tfac[n];
        w_hi = w_lo;
        thf_hi = thf_lo;
        th_0 = th_m;
    }
}

// last substep's damping + _finalize_time_averaged_velocity! (acoustic_substepping.jl:1225-1250); writes the periodic
// halo images and z-halo copies of the averaged velocities (they feed WENO stencils of the moisture tendency).
template <bool DAMP, class ST>
__global__ __launch_bounds__(256) void k_ac_finalize(DevGrid g, AcFieldsT<ST> F, AcParams P)
{
    const int i = blockIdx.x * 256 + threadIdx.x, j = blockIdx.y, k = blockIdx.z;
    if (i >= g.Nx) return;
    const WrapIdx W = wrap_of(g, i, j);
    const long long sz = g.Sxy;
    const long long n = g.idx(i, j, k), mx = n + W.im, my = n + W.jm;
    if (DAMP) {
        const double d0 = F.rthp[n] - F.rth_old[n];
        const double ddx = (d0 - (F.rthp[mx] - F.rth_old[mx])) * g.rdx;
        const double ddy = (d0 - (F.rthp[my] - F.rth_old[my])) * g.rdy;
        const double th = F.thL[n];
        double up = F.rup[n] - P.kdamp * ddx / ((th + F.thL[mx]) / 2.0);
        double vp = F.rvp[n] - P.kdamp * ddy / ((th + F.thL[my]) / 2.0);
        if (g.bounded_x && i == 0 && P.wall_w) up = 0.0;
        if (g.bounded_y && j == 0 && P.wall_s) vp = 0.0;
        F.rup[n] = up;
        F.rvp[n] = vp;
    }
    const double r0 = F.rho_d[n];
    double rx = (r0 + F.rho_d[mx]) / 2.0, ry = (r0 + F.rho_d[my]) / 2.0;
    rx = (rx == 0.0) ? 1.0 : rx;
    ry = (ry == 0.0) ? 1.0 : ry;
    const double ua = (F.ru[n] + F.au[n] * P.inv_N) / rx;
    const double va = (F.rv[n] + F.av[n] * P.inv_N) / ry;
    cst_img(F.au, n, ua, W.ox, W.oy);
    cst_img(F.av, n, va, W.ox, W.oy);
    double wa = 0.0;
    if (k > 0) {
        double rz = (r0 + F.rho_d[n - sz]) / 2.0;
        rz = (rz == 0.0) ? 1.0 : rz;
        wa = (F.rw[n] + F.aw[n] * P.inv_N) / rz;
    }
    cst_img(F.aw, n, wa, W.ox, W.oy);
    if (k == 0 || k == g.Nz - 1) {
        const long long h = (k == 0) ? -sz : sz;
        cst_img(F.au, n + h, ua, W.ox, W.oy);
        cst_img(F.av, n + h, va, W.ox, W.oy);
        if (k == g.Nz - 1) cst_img(F.aw, n + sz, 0.0, W.ox, W.oy);
    }
}

// _recover_full_state! (acoustic_substepping.jl:1274-1292) [+ WS-RK3 update of the moisture density,
// acoustic_runge_kutta_3.jl:189-192, when dt_stage_q != 0 pointer-wise]
template <int MOIST, class ST>      // MOIST 0: acoustic prognostics only; 1: + rho q^v; 2: + rho q^v and the Kessler species
__global__ __launch_bounds__(256) void k_ac_recover(DevGrid g, AcFieldsT<ST> F, double dt_stage)
{
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
    if (t >= (long long)g.Ny * g.Sx) return;
    const int k = blockIdx.y;
    const long long n = g.Sxy * (k + g.Hz) + (long long)g.Hy * g.Sx + t;
    if (g.bounded_x) {      // the halo columns of a Bounded x hold the caller's boundary values of the model fields, not periodic images
        const int c = (int)(t % g.Sx);
        if (c < g.Hx || c >= g.Hx + g.Nx) return;
    }
    F.rho_d[n] = F.rho_d[n] + F.rp[n];
    F.rth[n] = F.rth[n] + F.rthp[n];
    F.ru[n] = F.ru[n] + F.rup[n];
    F.rv[n] = F.rv[n] + F.rvp[n];
    F.rw[n] = F.rw[n] + F.rwp[n];
    if (MOIST) F.rq[n] = F.U0_rq[n] + dt_stage * F.G_rq[n];
    if (MOIST == 2) {
        F.rqcl[n] = F.U0_rqcl[n] + dt_stage * F.G_rqcl[n];
        F.rqr[n] = F.U0_rqr[n] + dt_stage * F.G_rqr[n];
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// Stage epilogue of the whole-step seam in ONE pass (round 4): k_ac_finalize + k_ac_recover<MOIST> +
// k_cmp_diagnose<true, LIN, MP> — three passes that re-read each other's output (17 + 18 + 18 words per cell) as one (38) plus the
// three-word density pass below.  Same expressions in the same order as the three kernels, so the results carry their bits:
//   * the last substep's divergence damping of (rho u)', (rho v)' (the damped values are consumed here and not stored: nothing reads
//     the perturbation fields between the recovery and the next stage's initialisation);
//   * time-averaged velocities from the accumulators and the STAGE-ENTRY densities, with their halo images;
//   * recovery U = U^L + U' of rho theta, momentum (stored in place: pointwise) and of rho_d — which is NOT stored here, because the
//     averages above and the velocities below read rho_d of the x / y / z neighbours: every thread forms the neighbours' new density
//     from the old one and rho' itself, and k_ac_recover_density writes the field afterwards;
//   * WS-RK3 update of the moisture density (and the Kessler species);
//   * update_state!: total density, velocities, theta, q, Newton temperature, pressure, every halo image, and (LIN) the next
//     stage's linearisation — theta_L goes to `thL_out`, a second buffer, because the damping of this pass reads the old theta_L
//     of the neighbours (bz_ctx::d_thL2; the stage driver alternates the two).
// Single-device contexts only (a y-slab has no rho' in its halo rows).
template <bool DAMP, bool LIN, int MP, class ST>
__global__ __launch_bounds__(256) void k_ac_stage_end(DevGrid g, AcFieldsT<ST> F, DiagFields D, AcParams P, double dt_stage, ST *__restrict__ thL_out,
                                                      double abstol, int maxiter)
{
    constexpr bool KES = (MP == 2), SA = (MP == 1);
    int bx, j, k;
    ac_pencil_block(bx, j, k);
    const int i = bx * 256 + threadIdx.x;
    if (i >= g.Nx) return;
    const WrapIdx W = wrap_of(g, i, j);
    const long long ox = W.ox, oy = W.oy;
    const long long sz = g.Sxy;
    const long long n = g.idx(i, j, k), mx = n + W.im, my = n + W.jm, mz = n - sz;
    const bool bot = (k == 0), top = (k == g.Nz - 1);

    // ---- k_ac_finalize: damping of the last substep ----
    ST up_s = F.rup[n], vp_s = F.rvp[n];
    if (DAMP) {
        const double d0 = F.rthp[n] - F.rth_old[n];
        const double ddx = (d0 - (F.rthp[mx] - F.rth_old[mx])) * g.rdx;
        const double ddy = (d0 - (F.rthp[my] - F.rth_old[my])) * g.rdy;
        const double th = F.thL[n];
        up_s -= P.kdamp * ddx / ((th + F.thL[mx]) / 2.0);
        vp_s -= P.kdamp * ddy / ((th + F.thL[my]) / 2.0);
    }
    // ---- k_ac_finalize: time-averaged velocities (stage-entry densities and momentum) ----
    const double r0 = F.rho_d[n], r_mx = F.rho_d[mx], r_my = F.rho_d[my];
    const double ru0 = F.ru[n], rv0 = F.rv[n];
    double r_mz = 0.0, rw0 = 0.0;
    if (!bot) {
        r_mz = F.rho_d[mz];
        rw0 = F.rw[n];
    }
    if (ac_accumulate(P)) {      // (a dry stage 1 / 2 of a whole step carries no accumulators: AcParams::skip_avg_if_dry)
        double rx = (r0 + r_mx) / 2.0, ry = (r0 + r_my) / 2.0;
        rx = (rx == 0.0) ? 1.0 : rx;
        ry = (ry == 0.0) ? 1.0 : ry;
        const double ua = (ru0 + F.au[n] * P.inv_N) / rx;
        const double va = (rv0 + F.av[n] * P.inv_N) / ry;
        cst_img(F.au, n, ua, ox, oy);
        cst_img(F.av, n, va, ox, oy);
        double wa = 0.0;
        if (!bot) {
            double rz = (r0 + r_mz) / 2.0;
            rz = (rz == 0.0) ? 1.0 : rz;
            wa = (rw0 + F.aw[n] * P.inv_N) / rz;
        }
        cst_img(F.aw, n, wa, ox, oy);
        if (bot || top) {
            const long long h = bot ? -sz : sz;
            cst_img(F.au, n + h, ua, ox, oy);
            cst_img(F.av, n + h, va, ox, oy);
            if (top) cst_img(F.aw, n + sz, 0.0, ox, oy);
        }
    }
    // ---- k_ac_recover ----
    const double rd = r0 + F.rp[n];
    const double rth = F.rth[n] + F.rthp[n];
    const double ru = ru0 + up_s, rv = rv0 + vp_s;
    const bool dryq = (MP == 0) && P.dry_q && __builtin_amdgcn_readfirstlane(*P.dry_q) == 1;
    const double rq = dryq ? 0.0 : F.U0_rq[n] + dt_stage * F.G_rq[n];
    double rqcl = 0.0, rqr = 0.0;
    if (KES) {
        rqcl = F.U0_rqcl[n] + dt_stage * F.G_rqcl[n];
        rqr = F.U0_rqr[n] + dt_stage * F.G_rqr[n];
    }
    // ---- k_cmp_diagnose<true, LIN, MP> on the recovered state ----
    const double rdx = (rd + (r_mx + F.rp[mx])) / 2.0;
    const double rdy = (rd + (r_my + F.rp[my])) / 2.0;
    const double u = ru / rdx, v = rv / rdy;
    if (P.out_of_place) cst_img(D.rho_d, n, rd, ox, oy);
    cst_img(D.rth, n, rth, ox, oy);
    cst_img(D.ru, n, ru, ox, oy);
    cst_img(D.rv, n, rv, ox, oy);
    cst_img(D.u, n, u, ox, oy);
    cst_img(D.v, n, v, ox, oy);
    if (!bot) {
        const double rw = rw0 + F.rwp[n];
        const double rdz = (rd + (r_mz + F.rp[mz])) / 2.0;
        cst_img(D.rw, n, rw, ox, oy);
        cst_img(D.w, n, rw / rdz, ox, oy);
    } else {
        cst_img(D.rw, n, 0.0, ox, oy);      // impenetrable walls
        cst_img(D.w, n, 0.0, ox, oy);
    }
    if (top) {
        cst_img(D.rw, n + sz, 0.0, ox, oy);
        cst_img(D.w, n + sz, 0.0, ox, oy);
    }
    double r, q, th, T, p, qcl_v = 0.0, qr_v = 0.0, sa_qv = 0.0, sa_ql = 0.0;
    {
        double ql = 0.0;
        if (KES) r = rd + (rq + (rqcl + (rqr + 0.0)));
        else r = rd + (rq + 0.0);
        th = rth / rd;
        q = rq / r;
        if (KES) {
            qcl_v = rqcl / r;
            qr_v = rqr / r;
            ql = qcl_v + qr_v;
        }
        double qvap = q;
        if (SA) {
            T = bz_ds_adjust(g, th, q, r, abstol, maxiter, qvap, ql);
            cst_img(g.qv_field, n, qvap, ox, oy);
            cst_img(g.ql_field, n, ql, ox, oy);
            sa_qv = qvap; sa_ql = ql;
        }
        const double qd = 1.0 - (qvap + ql);
        const double Rm = qd * g.Rd + qvap * g.Rv;
        const double cpm = (KES || SA) ? qd * g.cpd + qvap * g.cpv + ql * g.sa_cl : qd * g.cpd + qvap * g.cpv;
        if (!SA) {
            const double kap = Rm / cpm;
            const double gam = cpm / (cpm - Rm);
            const double Lt = KES ? (g.sa_Ll * ql) / cpm : 0.0;
            T = pow(th, gam) * pow(r * Rm / g.pst, gam - 1.0) + Lt;
            double dT = T;
            for (int it = 0; it < maxiter && fabs(dT) > abstol; ++it) {
                const double Phi = pow(r * Rm * T / g.pst, kap) * th;
                dT = -(T - Phi - Lt) / (1.0 - kap * Phi / T);
                T += dT;
            }
        }
        p = r * Rm * T;
        if (KES) {
            cst_img(F.rqcl, n, rqcl, ox, oy);
            cst_img(F.rqr, n, rqr, ox, oy);
            cst_img(g.qcl_field, n, qcl_v, ox, oy);
            cst_img(g.qr_field, n, qr_v, ox, oy);
            cst_img(g.qv_field, n, q, ox, oy);
        }
        if (!dryq || P.out_of_place) cst_img(D.rq, n, rq, ox, oy);
        cst_img(D.rho, n, r, ox, oy);
        cst_img(D.theta, n, th, ox, oy);
        if (!dryq) cst_img(D.q, n, q, ox, oy);
        cst_img(D.T, n, T, ox, oy);
        cst_img(D.p, n, p, ox, oy);
        if (LIN) {
            const double Pi = pow(p / g.pst, g.Rd / g.cpd);
            const double thl = rth / ((rd == 0.0) ? 1.0 : rd);
            const double gr = cpm * Rm / (cpm - Rm);
            st_store(D.Pi, n, Pi, D.st32);
            thL_out[n] = (ST)thl;
            st_store(D.gR, n, gr, D.st32);
            st_store(D.Clin, n, gr * Pi, D.st32);
        }
    }
    if (bot || top) {     // first z-halo cell of the no-flux centre fields (rho_d: k_ac_recover_density)
        const long long h = bot ? -sz : sz;
        if (P.out_of_place) cst_img(D.rho_d, n + h, rd, ox, oy);
        cst_img(D.ru, n + h, ru, ox, oy);
        cst_img(D.rv, n + h, rv, ox, oy);
        cst_img(D.rth, n + h, rth, ox, oy);
        cst_img(D.u, n + h, u, ox, oy);
        cst_img(D.v, n + h, v, ox, oy);
        if (!dryq || P.out_of_place) cst_img(D.rq, n + h, rq, ox, oy);
        cst_img(D.rho, n + h, r, ox, oy);
        cst_img(D.theta, n + h, th, ox, oy);
        if (!dryq) cst_img(D.q, n + h, q, ox, oy);
        cst_img(D.T, n + h, T, ox, oy);
        cst_img(D.p, n + h, p, ox, oy);
        if (SA) {
            cst_img(g.qv_field, n + h, sa_qv, ox, oy);
            cst_img(g.ql_field, n + h, sa_ql, ox, oy);
        }
        if (KES) {
            cst_img(F.rqcl, n + h, rqcl, ox, oy);
            cst_img(F.rqr, n + h, rqr, ox, oy);
            cst_img(g.qcl_field, n + h, qcl_v, ox, oy);
            cst_img(g.qr_field, n + h, qr_v, ox, oy);
            cst_img(g.qv_field, n + h, q, ox, oy);
        }
    }
}

// rho_d = rho_d^L + rho' with its halo images and z-halo copies: the one field of the recovery whose neighbours k_ac_stage_end reads
template <class ST>
__global__ __launch_bounds__(256) void k_ac_recover_density(DevGrid g, double *__restrict__ rho_d, const ST *__restrict__ rp)
{
    const int i = blockIdx.x * 256 + threadIdx.x, j = blockIdx.y, k = blockIdx.z;
    if (i >= g.Nx) return;
    const WrapIdx W = wrap_of(g, i, j);
    const long long n = g.idx(i, j, k);
    const double rd = rho_d[n] + rp[n];
    cst_img(rho_d, n, rd, W.ox, W.oy);
    if (k == 0 || k == g.Nz - 1) cst_img(rho_d, n + ((k == 0) ? -g.Sxy : g.Sxy), rd, W.ox, W.oy);
}

__global__ __launch_bounds__(256) void k_ws_rk3_scalar(DevGrid g, double *__restrict__ u, const double *__restrict__ u0,
                                                       const double *__restrict__ G, double dt_stage)
{
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
    if (t >= (long long)g.Ny * g.Sx) return;
    const int k = blockIdx.y;
    const long long n = g.Sxy * (k + g.Hz) + (long long)g.Hy * g.Sx + t;
    u[n] = u0[n] + dt_stage * G[n];
}

// ---------------------------------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------------------------------
static bool valid_state(const bz_compressible_state *s)
{
    return s && s->rho_d && s->rho && s->rho_u && s->rho_v && s->rho_w && s->rho_theta && s->rho_q && s->u && s->v &&
           s->w && s->theta && s->q && s->T && s->p;
}
static bool valid_prog(const bz_compressible_prognostic *P)
{
    return P && P->rho_d && P->rho_u && P->rho_v && P->rho_w && P->rho_theta && P->rho_q;
}
static bool valid_sub(const bz_acoustic_substepper *a)
{
    return a && a->exner && a->potential_temperature && a->gamma_R_mixture && a->density_perturbation &&
           a->density_potential_temperature_perturbation && a->momentum_perturbation_u && a->momentum_perturbation_v &&
           a->momentum_perturbation_w && a->density_predictor && a->density_potential_temperature_predictor &&
           a->previous_density_potential_temperature_perturbation && a->time_averaged_u && a->time_averaged_v &&
           a->time_averaged_w && a->slow_vertical_momentum_tendency && a->vertical_solver_source_term;
}
// Contexts with a Bounded x or y run the acoustic loop (bz_refresh_linearization, bz_acoustic_substep_loop, bz_acoustic_stage_begin /
// _substep / _stage_end); the rest of the compressible model on lateral walls — wall-aware slow tendencies, update_state! with the model's
// boundary conditions — is not built
#define BZ_REJECT_WALLS(what)                                                                                                          \
    do {                                                                                                                               \
        if (ctx->dg.bounded_x || ctx->dg.bounded_y) {                                                                                  \
            ctx->last_error = what ": not implemented on a Bounded x or y (compressible contexts with lateral walls run the acoustic substep loop only)"; \
            return BZ_ERR_UNSUPPORTED;                                                                                                 \
        }                                                                                                                              \
    } while (0)
#define BZ_REQUIRE_COMPRESSIBLE()                                                      \
    do {                                                                               \
        if (!ctx) return BZ_ERR_INVALID;                                               \
        if (!ctx->compressible) {                                                      \
            ctx->last_error = "context was not created by bz_create_compressible";    \
            return BZ_ERR_INVALID;                                                     \
        }                                                                              \
    } while (0)

static int bzi_create_compressible(bz_ctx **out, const bz_grid *grid, const bz_constants *constants,
                                   const bz_exner_reference_state *ref, const bz_split_explicit *td, int weno_order,
                                   int y_nranks, int y_rank, bool slab);

extern "C" int bz_create_compressible(bz_ctx **out, const bz_grid *grid, const bz_constants *constants,
                                      const bz_exner_reference_state *ref, const bz_split_explicit *td, int weno_order)
{
    return bzi_create_compressible(out, grid, constants, ref, td, weno_order, 1, 0, false);
}

extern "C" int bz_create_compressible_slab(bz_ctx **out, const bz_grid *local_grid, const bz_constants *constants,
                                           const bz_exner_reference_state *ref, const bz_split_explicit *td, int weno_order,
                                           int y_nranks, int y_rank)
{
    // substep_floattype = Float32 inside a Float64 model: the per-substep halo messages carry the Float32 rows as Sx / 2 doubles (bz_comm.hip:
    // halo_exchange, half) — the row length must be even; library-owned communicators only (the host-driven exchange moves the grid's real)
    if (td && td->substep_float_bytes == 4 && sizeof(double) == 8 && local_grid && ((local_grid->Nx + 2 * local_grid->Hx) & 1)) return BZ_ERR_UNSUPPORTED;
    if (y_nranks < 1 || y_rank < 0 || y_rank >= y_nranks) return BZ_ERR_INVALID;
    return bzi_create_compressible(out, local_grid, constants, ref, td, weno_order, y_nranks, y_rank, true);
}

static int bzi_create_compressible(bz_ctx **out, const bz_grid *grid, const bz_constants *constants,
                                   const bz_exner_reference_state *ref, const bz_split_explicit *td, int weno_order,
                                   int y_nranks, int y_rank, bool slab)
{
    if (!out || !grid || !constants || !ref || !td) return BZ_ERR_INVALID;
    if ((ref->pressure == nullptr) != (ref->density == nullptr)) return BZ_ERR_INVALID;
    if (td->substeps < 0 || !(td->acoustic_cfl > 0.0) || td->newton_maxiter < 0) return BZ_ERR_INVALID;
    if (td->sponge_ramp < 0 || td->sponge_ramp > 3 || (td->sponge_ramp && !(td->sponge_depth > 0.0))) return BZ_ERR_INVALID;
    if (td->substep_distribution < 0 || td->substep_distribution > 2) return BZ_ERR_INVALID;
    // substep_floattype: 0 = eltype(grid); 4 = Float32 working fields (inside a Float64 model: half the bytes; in the Float32 library: eltype)
    if (td->substep_float_bytes != 0 && td->substep_float_bytes != 4 && td->substep_float_bytes != (int32_t)sizeof(double)) return BZ_ERR_UNSUPPORTED;
    if (td->substep_float_bytes == 4 && sizeof(double) == 8 && td->direct_divergence_damping) return BZ_ERR_UNSUPPORTED;   // its two kernels read the working fields as the grid's real
    const int nc = grid->Nz + 2 * grid->Hz;
    std::vector<double> zeros((size_t)nc, 0.0);
    bz_reference_state r;
    r.surface_pressure = 0.0;
    r.potential_temperature = 0.0;
    r.standard_pressure = ref->standard_pressure;
    r.density = ref->density ? ref->density : zeros.data();
    r.pressure = ref->pressure ? ref->pressure : zeros.data();
    r.temperature = zeros.data();
    int rc = bzi_create(out, grid, constants, &r, weno_order, y_nranks, y_rank, slab, true);
    if (rc != BZ_OK) return rc;
    bz_ctx *ctx = *out;
    ctx->se = *td;
    ctx->has_reference = ref->density != nullptr;
    const size_t ncell = (size_t)ctx->dg.Sxy * (size_t)nc;
    ctx->ac_fused = !ctx->tune.no_ac_fuse;
    ctx->substep_f32 = (td->substep_float_bytes == 4) && sizeof(double) == 8;
    if (hipMalloc(&ctx->d_Clin, ncell * sizeof(double)) != hipSuccess ||
        hipMalloc(&ctx->d_tfac_ac, ncell * sizeof(double)) != hipSuccess ||
        hipMalloc(&ctx->d_up2, ncell * sizeof(double)) != hipSuccess ||
        hipMalloc(&ctx->d_thL2, ncell * sizeof(double)) != hipSuccess ||
        hipMalloc(&ctx->d_vp2, ncell * sizeof(double)) != hipSuccess ||
        (ctx->tune.ac_pfold && (hipMalloc(&ctx->d_Gp_ru, ncell * sizeof(double)) != hipSuccess ||
                                hipMalloc(&ctx->d_Gp_rv, ncell * sizeof(double)) != hipSuccess))) {
        bz_destroy(ctx);
        *out = nullptr;
        return BZ_ERR_ALLOC;
    }
    if (ctx->d_Gp_ru) {
        (void)hipMemset(ctx->d_Gp_ru, 0, ncell * sizeof(double));
        (void)hipMemset(ctx->d_Gp_rv, 0, ncell * sizeof(double));
    }
    (void)hipMemset(ctx->d_Clin, 0, ncell * sizeof(double));
    (void)hipMemset(ctx->d_tfac_ac, 0, ncell * sizeof(double));
    (void)hipMemset(ctx->d_up2, 0, ncell * sizeof(double));
    (void)hipMemset(ctx->d_thL2, 0, ncell * sizeof(double));
    (void)hipMemset(ctx->d_vp2, 0, ncell * sizeof(double));
    // UpperSponge profile on the faces: rate * ramp(z, grid.Lz, depth), ramp = 0 below Lz - depth and 1 at z = Lz
    // (time_discretizations.jl:398-433; the reference passes grid.Lz as the sponge top, whatever z[0] is)
    std::vector<double> sp((size_t)grid->Nz + 1, 0.0);
    if (td->sponge_ramp) {
        const double Lz = grid->zf[grid->Nz] - grid->zf[0], depth = td->sponge_depth, pi = 3.14159265358979323846;
        for (int k = 0; k <= grid->Nz; ++k) {
            double sN = (grid->zf[k] - (Lz - depth)) / depth;
            sN = sN < 0.0 ? 0.0 : (sN > 1.0 ? 1.0 : sN);
            const double ramp = td->sponge_ramp == 1 ? sN : td->sponge_ramp == 2 ? sN * sN * (3.0 - 2.0 * sN) : std::sin(pi / 2.0 * sN) * std::sin(pi / 2.0 * sN);
            sp[k] = td->sponge_damping_rate * ramp;
        }
    }
    if (hipMalloc(&ctx->d_sponge, sp.size() * sizeof(double)) != hipSuccess) {
        bz_destroy(ctx);
        *out = nullptr;
        return BZ_ERR_ALLOC;
    }
    BZ_HIP(hipMemcpy(ctx->d_sponge, sp.data(), sp.size() * sizeof(double), hipMemcpyHostToDevice));
    return BZ_OK;
}

void bzi_compressible_teardown(bz_ctx *ctx)
{
    if (ctx->d_Clin) (void)hipFree(ctx->d_Clin);
    if (ctx->d_tfac_ac) (void)hipFree(ctx->d_tfac_ac);
    if (ctx->d_up2) (void)hipFree(ctx->d_up2);
    if (ctx->d_thL2) (void)hipFree(ctx->d_thL2);
    ctx->d_thL2 = nullptr;
    if (ctx->d_vp2) (void)hipFree(ctx->d_vp2);
    if (ctx->d_sponge) (void)hipFree(ctx->d_sponge);
    if (ctx->d_Gp_ru) (void)hipFree(ctx->d_Gp_ru);
    if (ctx->d_Gp_rv) (void)hipFree(ctx->d_Gp_rv);
    ctx->d_Gp_ru = ctx->d_Gp_rv = nullptr;
    ctx->d_Clin = ctx->d_tfac_ac = ctx->d_up2 = ctx->d_vp2 = ctx->d_sponge = nullptr;
}

static DiagFields diag_fields(bz_ctx *ctx, const bz_compressible_state *s, const bz_acoustic_substepper *sub)
{
    DiagFields F;
    F.rho_d = s->rho_d; F.rho = s->rho; F.ru = s->rho_u; F.rv = s->rho_v; F.rw = s->rho_w; F.rth = s->rho_theta; F.rq = s->rho_q;
    F.u = s->u; F.v = s->v; F.w = s->w; F.theta = s->theta; F.q = s->q; F.T = s->T; F.p = s->p;
    F.Pi = sub ? sub->exner : nullptr;
    F.thL = sub ? sub->potential_temperature : nullptr;
    F.gR = sub ? sub->gamma_R_mixture : nullptr;
    F.Clin = ctx->d_Clin;
    F.st32 = ctx->substep_f32 ? 1 : 0;
    return F;
}

static int launch_scalar_rho3d(bz_ctx *ctx, const char *name, double *Gc, double *Grho, const double *rho, const double *u,
                               const double *v, const double *w, const double *c, const double *ru, const double *rv,
                               const double *rw, const int *zero_if_dry = nullptr)
{
    const DevGrid &g = ctx->dg;
    ProfileScope ps(ctx, name);
    if (ctx->weno_R != 3) return bzi_scalar_rho3d_generic(ctx, Gc, Grho, rho, u, v, w, c, ru, rv, rw);
    if (!ctx->tune.no_rho3d_exchange && ctx->tune.scalar_lds && !g.flat_y && !g.bounded_x && !g.bounded_y && g.Nx % 64 == 0 && g.Ny % SLT == 0 && g.Hx >= 3 &&
        g.Hy >= 3 && g.Hz >= 3) {
        int kc = 64;
        while (kc > 8 && (long long)(g.Nx / 64) * (g.Ny / SLT) * ((g.Nz + kc - 1) / kc) < 2048) kc >>= 1;
        dim3 block(64, SLT), grid(g.Nx / 64, g.Ny / SLT, (g.Nz + kc - 1) / kc);
        if (Grho) hipLaunchKernelGGL(k_scalar_rho3d_lds<true>, grid, block, 0, ctx->stream, g, Gc, Grho, rho, u, v, w, c, ru, rv, rw, zero_if_dry, kc);
        else hipLaunchKernelGGL(k_scalar_rho3d_lds<false>, grid, block, 0, ctx->stream, g, Gc, Grho, rho, u, v, w, c, ru, rv, rw, zero_if_dry, kc);
        BZ_LAUNCH_CHECK();
        return BZ_OK;
    }
    if (!ctx->tune.no_rho3d_exchange && !g.flat_y && !g.bounded_x && !g.bounded_y && g.Nx % 64 == 0 && g.Ny % CTY == 0) {
        int kc = 64;      // levels per workgroup: >= 8 wavefronts per SIMD (see march_chunk in bz_tendency_generic.hip)
        while (kc > 8 && (long long)(g.Nx / 64) * g.Ny * ((g.Nz + kc - 1) / kc) < 8192) kc >>= 1;
        dim3 block(64, CTY), grid(g.Nx / 64, g.Ny / CTY, (g.Nz + kc - 1) / kc);
        hipLaunchKernelGGL(k_scalar_tendency_rho3d_x, grid, block, 0, ctx->stream, g, Gc, Grho, rho, u, v, w, c, ru, rv, rw, zero_if_dry, kc);
        BZ_LAUNCH_CHECK();
        return BZ_OK;
    }
    const int kc = pick_kchunk_c(g, g.Nz);
    dim3 block(64, CTY), grid((g.Nx + 63) / 64, (g.Ny + CTY - 1) / CTY, (g.Nz + kc - 1) / kc);
    hipLaunchKernelGGL(k_scalar_tendency_rho3d, grid, block, 0, ctx->stream, g, Gc, Grho, rho, u, v, w, c, ru, rv, rw, kc, zero_if_dry);
    BZ_LAUNCH_CHECK();
    return BZ_OK;
}

// update_state! with the linearisation refresh of the next stage optionally folded in
static int bzi_compressible_update_state(bz_ctx *ctx, const bz_compressible_state *s, const bz_compressible_prognostic *G,
                                         const bz_acoustic_substepper *sub, bool compute_tendencies, bool with_linearization)
{
    const DevGrid &g = ctx->dg;
    {
        ProfileScope ps(ctx, with_linearization ? "update_state+linearization" : "update_state");
        DiagFields F = diag_fields(ctx, s, sub);
        if (with_linearization) ctx->thL_alt = false;      // theta_L of the next stage goes to the caller's array
        dim3 grid((g.Nx + 255) / 256, g.Ny, g.Nz), block(256);
        const double na = ctx->se.newton_abstol;
        const int nm = ctx->se.newton_maxiter;
#define BZ_DIAG(LIN, MP) hipLaunchKernelGGL((k_cmp_diagnose<true, LIN, MP>), grid, block, 0, ctx->stream, g, F, na, nm)
        if (with_linearization) {
            if (g.microphysics == 2) BZ_DIAG(true, 2);
            else if (g.microphysics == 1) BZ_DIAG(true, 1);
            else BZ_DIAG(true, 0);
        } else {
            if (g.microphysics == 2) BZ_DIAG(false, 2);
            else if (g.microphysics == 1) BZ_DIAG(false, 1);
            else BZ_DIAG(false, 0);
        }
#undef BZ_DIAG
        BZ_LAUNCH_CHECK();
    }
    if (compute_tendencies) return bz_compute_moisture_tendency(ctx, s, G, sub);
    return BZ_OK;
}

extern "C" int bz_compressible_update_state(bz_ctx *ctx, const bz_compressible_state *s, const bz_compressible_prognostic *G,
                                            const bz_acoustic_substepper *sub, int compute_tendencies)
{
    BZ_REQUIRE_COMPRESSIBLE();
    BZ_REJECT_WALLS("bz_compressible_kessler_update");
    BZ_REJECT_WALLS("bz_compressible_update_state");
    if (!valid_state(s)) return BZ_ERR_INVALID;
    if (compute_tendencies && (!valid_prog(G) || !valid_sub(sub))) return BZ_ERR_INVALID;
    if (!ctx->fused_ok) { ctx->last_error = "compressible path needs Nx >= 2Hx and Ny >= 2Hy"; return BZ_ERR_UNSUPPORTED; }
    if (ctx->d_qstate) BZ_HIP(hipMemsetAsync(ctx->d_qstate, 0, sizeof(int), ctx->stream));      // moisture scan: unknown again (set! ends here)
    bzi_moisture_unknown(ctx);
    return bzi_compressible_update_state(ctx, s, G, sub, compute_tendencies != 0, false);
}

extern "C" int bz_refresh_linearization(bz_ctx *ctx, const bz_compressible_state *s, const bz_acoustic_substepper *sub)
{
    BZ_REQUIRE_COMPRESSIBLE();
    if (!valid_state(s) || !valid_sub(sub)) return BZ_ERR_INVALID;
    const DevGrid &g = ctx->dg;
    ProfileScope ps(ctx, "refresh_linearization");
    ctx->thL_alt = false;
    const int hrows = g.wrap_y ? 0 : g.bounded_y ? 1 : ((ctx->se.direct_divergence_damping && ctx->se.damping_coefficient >= 0.0) ? 2 : 1);
    const int hcols = g.bounded_x ? 1 : 0;
    dim3 grid((g.Nx + 2 * hcols + 255) / 256, g.Ny + 2 * hrows, g.Nz), block(256);
    hipLaunchKernelGGL(k_cmp_linearization, grid, block, 0, ctx->stream, g, sub->exner, sub->potential_temperature,
                       sub->gamma_R_mixture, ctx->d_Clin, s->p, s->rho_d, s->rho_theta, s->q, ctx->substep_f32 ? 1 : 0, hrows, hcols);
    BZ_LAUNCH_CHECK();
    return BZ_OK;
}

extern "C" int bz_seed_time_averaged_velocities(bz_ctx *ctx, const bz_compressible_state *s, const bz_acoustic_substepper *sub)
{
    BZ_REQUIRE_COMPRESSIBLE();
    if (!valid_state(s) || !valid_sub(sub)) return BZ_ERR_INVALID;
    const DevGrid &g = ctx->dg;
    const size_t nc = (size_t)g.Sxy * (size_t)(g.Nz + 2 * g.Hz) * sizeof(double);
    const size_t nf = (size_t)g.Sxy * (size_t)(g.Nz + 1 + 2 * g.Hz) * sizeof(double);
    BZ_HIP(hipMemcpyAsync(sub->time_averaged_u, s->u, nc, hipMemcpyDeviceToDevice, ctx->stream));
    BZ_HIP(hipMemcpyAsync(sub->time_averaged_v, s->v, nc, hipMemcpyDeviceToDevice, ctx->stream));
    BZ_HIP(hipMemcpyAsync(sub->time_averaged_w, s->w, nf, hipMemcpyDeviceToDevice, ctx->stream));
    return BZ_OK;
}

extern "C" int bz_compute_slow_tendencies(bz_ctx *ctx, const bz_compressible_state *s, const bz_compressible_prognostic *G)
{
    BZ_REQUIRE_COMPRESSIBLE();
    BZ_REJECT_WALLS("bz_compute_slow_tendencies");
    if (!valid_state(s) || !valid_prog(G)) return BZ_ERR_INVALID;
    bz_state a;
    std::memset(&a, 0, sizeof(a));
    a.rho_u = s->rho_u; a.rho_v = s->rho_v; a.rho_w = s->rho_w;
    a.u = s->u; a.v = s->v; a.w = s->w; a.T = s->p; a.q = s->rho;
    a.rho_theta = s->rho_theta; a.rho_q = s->rho_q; a.theta = s->theta;
    bz_prognostic Ga;
    Ga.rho_u = G->rho_u; Ga.rho_v = G->rho_v; Ga.rho_w = G->rho_w; Ga.rho_theta = G->rho_theta; Ga.rho_q = G->rho_q;
    int rc;
    if (ctx->weno_R != 3) {      // WENO(order = 7 / 9): generic kernels (bz_tendency_generic.hip)
        if ((rc = bzi_momentum_advection_generic(ctx, &a, &Ga))) return rc;
    } else if (ctx->dg.flat_y) {
        if ((rc = bzi_momentum_advection_gen1(ctx, &a, &Ga))) return rc;
    } else {
        rc = bzi_u_tendency_lds(ctx, &a, &Ga);
        if (rc) return rc;
        rc = bzi_v_tendency_lds(ctx, &a, &Ga);
        if (rc) return rc;
        rc = bzi_w_tendency_ring(ctx, &a, &Ga, nullptr, nullptr, 1);
        if (rc) return rc;
    }
    if ((rc = launch_scalar_rho3d(ctx, "density+potential_temperature_tendency", G->rho_theta, G->rho_d, s->rho_d, s->u, s->v,
                                  s->w, s->theta, s->rho_u, s->rho_v, s->rho_w))) return rc;
    // - f x (rho U) of an FPlane and the density-keyed sponges are slow terms too (dynamics_kernel_functions.jl:79,99 through the same
    // x / y_momentum_tendency; examples/tropical_cyclone_with_rainband.jl:434-514)
    if (ctx->has_forcings && (rc = bzi_apply_forcings(ctx, &a, G->rho_u, G->rho_v, G->rho_theta, G->rho_q, 1.0))) return rc;
    return bzi_apply_relaxation(ctx, &a, &Ga, s->rho_d);
}

// compute_acoustic_substeps / stage_substep_count_and_size(::ProportionalSubsteps) (acoustic_substepping.jl:451-495)
static int acoustic_substeps_for(const bz_ctx *ctx, double dt)
{
    const double Rd = ctx->constants.dry_air_gas_constant, cpd = ctx->constants.dry_air_heat_capacity;
    const double gam = cpd / (cpd - Rd);
    const double cs = std::sqrt(gam * Rd * 300.0);
    const double dmin = ctx->dg.flat_y ? ctx->dg.dx : std::fmin(ctx->dg.dx, ctx->dg.dy);      // Flat directions do not bound the acoustic CFL (acoustic_substepping.jl:458-465)
    const double n = std::ceil(std::fabs(dt) * cs / (ctx->se.acoustic_cfl * dmin));
    return (int)std::fmax(1.0, n);
}

extern "C" int bz_stage_substeps(bz_ctx *ctx, double dt, double beta, int32_t *n_substeps, double *dtau)
{
    BZ_REQUIRE_COMPRESSIBLE();
    const double dt_stage = beta * dt;
    int n;
    double dtau_v;
    const int dist = ctx->se.substep_distribution;
    if (dist == 2 && beta < (1.0 / 3.0 + 1.0 / 2.0) / 2.0) {      // MonolithicFirstStage: stage 1 collapses to one substep of dt / 3 (:503-508)
        n = 1;
        dtau_v = dt / 3.0;
    } else if (dist == 1 || dist == 2) {      // ConstantSubstepSize (:497-501): one size dt / N for all stages, N a multiple of 6 so that beta N is integral
        const int n_raw = ctx->se.substeps > 0 ? ctx->se.substeps : acoustic_substeps_for(ctx, dt);
        const int N = std::max(6, 6 * ((n_raw + 5) / 6));
        n = std::max(1, (int)std::lround(beta * (double)N));
        dtau_v = dt / (double)N;
    } else {                                  // ProportionalSubsteps (:491-495)
        if (ctx->se.substeps > 0) n = (int)std::fmax(1.0, std::ceil(beta * (double)ctx->se.substeps));
        else n = acoustic_substeps_for(ctx, dt_stage);
        dtau_v = dt_stage / (double)n;
    }
    if (n_substeps) *n_substeps = n;
    if (dtau) *dtau = dtau_v;
    return BZ_OK;
}

static AcFields ac_fields(bz_ctx *ctx, const bz_compressible_state *s, const bz_compressible_prognostic *U0,
                          const bz_compressible_prognostic *G, const bz_acoustic_substepper *a)
{
    AcFields F;
    F.rho_d = s->rho_d; F.rth = s->rho_theta; F.ru = s->rho_u; F.rv = s->rho_v; F.rw = s->rho_w; F.rq = s->rho_q;
    F.rho = s->rho; F.p = s->p;
    F.U0_rho_d = U0->rho_d; F.U0_rth = U0->rho_theta; F.U0_ru = U0->rho_u; F.U0_rv = U0->rho_v; F.U0_rw = U0->rho_w; F.U0_rq = U0->rho_q;
    F.G_rho_d = G->rho_d; F.G_rth = G->rho_theta; F.G_ru = G->rho_u; F.G_rv = G->rho_v; F.G_rw = G->rho_w; F.G_rq = G->rho_q;
    F.thL = ctx->thL_alt ? ctx->d_thL2 : a->potential_temperature; F.Clin = ctx->d_Clin;
    F.rp = a->density_perturbation; F.rthp = a->density_potential_temperature_perturbation;
    F.rup = a->momentum_perturbation_u; F.rvp = a->momentum_perturbation_v; F.rwp = a->momentum_perturbation_w;
    F.rs = a->density_predictor; F.rths = a->density_potential_temperature_predictor;
    F.rth_old = a->previous_density_potential_temperature_perturbation;
    F.au = a->time_averaged_u; F.av = a->time_averaged_v; F.aw = a->time_averaged_w;
    F.Gs = a->slow_vertical_momentum_tendency; F.phi = a->vertical_solver_source_term;
    F.tfac = ctx->d_tfac_ac;
    F.sponge = ctx->d_sponge;
    F.rup_in = F.rup; F.rvp_in = F.rvp; F.rthp_out = F.rthp;
    F.Gp_ru = ctx->d_Gp_ru; F.Gp_rv = ctx->d_Gp_rv;
    const bz_kessler_model_fields &K = ctx->kessler;
    F.rqcl = K.cloud_liquid_density; F.rqr = K.rain_density;
    F.U0_rqcl = K.U0_cloud_liquid_density; F.U0_rqr = K.U0_rain_density;
    F.G_rqcl = K.G_cloud_liquid_density; F.G_rqr = K.G_rain_density;
    return F;
}

// apply_divergence_damping!(::DirectDivergenceDamping) (acoustic_substepping.jl:1158-1188): delta = V^-1 (dx(thetaF^x) + dy(thetaF^y))
// into the density predictor (free between recovery and the next predictor build), then the theta_L-scaled gradient of delta onto the
// horizontal momentum perturbations.  Periodic neighbours by wrap indexing: no halo fill of delta, (rho u)', (rho v)' or theta_L.
// y-slabs (wrap_y == 0): the neighbours are halo rows — theta_L is linearised on one halo row each side, (rho u)' and (rho v)' are
// exchanged by the driver between the column solve and this pair (bz_acoustic_direct_damping) — and delta is evaluated from row -1
// (jofs = -1, Ny + 1 rows) so that the gradient at row 0 needs no exchange of its own.
__global__ __launch_bounds__(256) void k_ac_direct_delta(DevGrid g, double *__restrict__ delta, const double *__restrict__ thL,
                                                         const double *__restrict__ up, const double *__restrict__ vp, int jofs)
{
    const int i = blockIdx.x * 256 + threadIdx.x, j = (int)blockIdx.y + jofs, k = blockIdx.z;
    if (i >= g.Nx) return;
    const long long n = g.idx(i, j, k);
    const long long ip = (i + 1 < g.Nx || g.bounded_x) ? 1 : 1 - g.Nx, im = (i > 0 || g.bounded_x) ? -1 : g.Nx - 1;
    const long long jp = (j + 1 < g.Ny || !g.wrap_y) ? (long long)g.Sx : (long long)g.Sx * (1 - g.Ny);
    const long long jm = (j > 0 || !g.wrap_y) ? -(long long)g.Sx : (long long)g.Sx * (g.Ny - 1);
    const double Ax = g.Ax[k], Ay = g.Ay[k];
    // Bounded directions: the east / north wall face is an exact zero (AcParams::wall_w); the west / south one holds the value the substep stored
    const double up1 = (g.bounded_x && i == g.Nx - 1) ? 0.0 : up[n + ip], vp1 = (g.bounded_y && j == g.Ny - 1) ? 0.0 : vp[n + jp];
    const double fx = Ax * ((thL[n + ip] + thL[n]) / 2.0) * up1 - Ax * ((thL[n] + thL[n + im]) / 2.0) * up[n];
    const double fy = Ay * ((thL[n + jp] + thL[n]) / 2.0) * vp1 - Ay * ((thL[n] + thL[n + jm]) / 2.0) * vp[n];
    delta[n] = (fx + fy) * g.Vinv_c[k];
}
__global__ __launch_bounds__(256) void k_ac_direct_apply(DevGrid g, const double *__restrict__ delta, const double *__restrict__ thL,
                                                         double *__restrict__ up, double *__restrict__ vp, double alpha, int wall_w, int wall_s)
{
    const int i = blockIdx.x * 256 + threadIdx.x, j = blockIdx.y, k = blockIdx.z;
    if (i >= g.Nx) return;
    const long long n = g.idx(i, j, k);
    const long long im = (i > 0 || g.bounded_x) ? -1 : g.Nx - 1;
    const long long jm = (j > 0 || !g.wrap_y) ? -(long long)g.Sx : (long long)g.Sx * (g.Ny - 1);
    double u1 = up[n] + alpha * (g.dx * g.dx) * ((delta[n] - delta[n + im]) * g.rdx) / ((thL[n] + thL[n + im]) / 2.0);
    double v1 = vp[n] + alpha * (g.dy * g.dy) * ((delta[n] - delta[n + jm]) * g.rdy) / ((thL[n] + thL[n + jm]) / 2.0);
    if (g.bounded_x && i == 0 && wall_w) u1 = 0.0;      // enforce_wall_impenetrability! after the damping (acoustic_substepping.jl:1547-1550)
    if (g.bounded_y && j == 0 && wall_s) v1 = 0.0;
    up[n] = u1;
    vp[n] = v1;
}

// ---------------------------------------------------------------------------------------------------------------------
// Lateral boundaries of the acoustic loop (round 6): topologies with a Bounded x and / or y.
// The substepper's own fields carry the default boundary conditions of their location (acoustic_substepping.jl:207-231; the test
// test/acoustic_substepping_open_boundaries.jl:70-73 pins that the momentum perturbations do not inherit the model's): a field that is a
// centre along a Bounded direction gets a zero-gradient halo from fill_halo_regions!, the wall faces of a face field are not touched by it.
// k_ac_fill_walls is that fill for the one halo row / column the substep kernels read (the kernels reach neighbours through halo cells on
// Bounded directions: wrap_of); the model's own fields (rho_d, rho theta, p, the slow tendencies) keep the halos the caller's boundary
// conditions gave them.
// ---------------------------------------------------------------------------------------------------------------------
template <class T>
__global__ __launch_bounds__(256) void k_ac_fill_walls(DevGrid g, T *__restrict__ f)
{
    const int t = blockIdx.x * 256 + threadIdx.x, k = blockIdx.y;
    const int ny_rows = g.bounded_y ? 2 * g.Nx : 0, nx_cols = g.bounded_x ? 2 * g.Ny : 0;
    if (t < ny_rows) {
        const int side = t / g.Nx, i = t - side * g.Nx;
        f[g.idx(i, side ? g.Ny : -1, k)] = f[g.idx(i, side ? g.Ny - 1 : 0, k)];
    } else if (t - ny_rows < nx_cols) {
        const int u = t - ny_rows, side = u / g.Ny, j = u - side * g.Ny;
        f[g.idx(side ? g.Nx : -1, j, k)] = f[g.idx(side ? g.Nx - 1 : 0, j, k)];
    }
}

// _relax_open_boundary_x! / _relax_open_boundary_y! (acoustic_substepping.jl:1323-1337): the outermost cell of rho' and (rho theta)' is pulled
// towards the prescribed wall value v; the model's Value boundary condition left rho^L[halo] = 2 v - rho^L[cell], so the target perturbation
// v - rho^L[cell] is (rho^L[halo] - rho^L[cell]) / 2.  One launch per open side: dir 0 = a y-z plane (cell column cb, halo column ch), 1 = an x-z plane.
template <class ST>
__global__ __launch_bounds__(256) void k_ac_relax_open_boundary(DevGrid g, ST *__restrict__ rp, ST *__restrict__ rthp, const double *__restrict__ rho_d,
                                                                const double *__restrict__ rth, int dir, int cb, int ch, double alpha)
{
    const int t = blockIdx.x * 256 + threadIdx.x, k = blockIdx.y;
    if (t >= (dir ? g.Nx : g.Ny)) return;
    const long long nb = dir ? g.idx(t, cb, k) : g.idx(cb, t, k), nh = dir ? g.idx(t, ch, k) : g.idx(ch, t, k);
    const double r = rp[nb], q = rthp[nb];
    rp[nb] = r + alpha * ((rho_d[nh] - rho_d[nb]) / 2.0 - r);
    rthp[nb] = q + alpha * ((rth[nh] - rth[nb]) / 2.0 - q);
}

// ---- one WS-RK3 stage of the acoustic loop in three pieces (the y-slab driver exchanges halos between them) ---------
struct AcStage {
    int ntau = 0, cur = 0, done = 0;
    bool damping = false, fused = true, direct = false;
    bool fwd2 = false, pfold = false;      // k_ac_forward2 runs the forward sweeps of this stage; with the p^L gradient folded into Gp_ru / Gp_rv
    int init_mode = 0;                     // 1 / 2: the first sweeps of the stage form the initial perturbations (U0 - U / zeros) instead of reading stored ones
    AcParams P;
};
// k_ac_forward2 addresses every array by a 32-bit byte offset from its base and fetches the outer x neighbours of a row's two edge lanes
// with one load: arrays below 4 GB, two z halo levels (its ring words are requested two levels ahead), no row whose first lane is its last
static int forward2_cfg(const bz_ctx *ctx)
{
    const int cfg = ctx->tune.ac_cfg;
    return (cfg != 0 && cfg != 1 && cfg != 2 && cfg != 4 && cfg != 5 && cfg != 6 && cfg != 8 && cfg != 12 && cfg != 13 && cfg != 22 && cfg != 28) ? 29 : cfg;
}
static bool ac_forward2_ok(const bz_ctx *ctx)
{
    const DevGrid &g = ctx->dg;
    const unsigned long long bytes = (unsigned long long)g.Sxy * (unsigned long long)(g.Nz + 2 * g.Hz + 1) * sizeof(double);
    return ctx->tune.ac_forward2 && bytes < (1ull << 32) && g.Hz >= 1 && g.Nx >= 2 && (g.Nx % 64) != 1 && !g.bounded_x && !g.bounded_y;
}
// ---- lateral walls / open boundaries (Bounded x and / or y) ----
static bool ac_walls(const bz_ctx *ctx) { return ctx->dg.bounded_x || ctx->dg.bounded_y; }
// fill_halo_regions! of a substepper field that is a centre along the Bounded directions: one zero-gradient halo row / column
template <class T>
static void fill_walls(bz_ctx *ctx, T *f)
{
    const DevGrid &g = ctx->dg;
    const int n = (g.bounded_y ? 2 * g.Nx : 0) + (g.bounded_x ? 2 * g.Ny : 0);
    if (!n) return;
    hipLaunchKernelGGL((k_ac_fill_walls<T>), dim3((n + 255) / 256, g.Nz), dim3(256), 0, ctx->stream, g, f);
}
static void fill_walls_st(bz_ctx *ctx, double *f)      // a working field in the substep storage type
{
    if (ctx->substep_f32) fill_walls(ctx, (float *)f);
    else fill_walls(ctx, f);
}
// apply_open_boundary_relaxation! (acoustic_substepping.jl:1339-1361) on rho' and the current (rho theta)'
static void relax_open_boundaries(bz_ctx *ctx, const AcFields &F, double *rthp_cur)
{
    const DevGrid &g = ctx->dg;
    const double a = ctx->ac_open_relax;
    auto launch = [&](int dir, int cb, int ch) {
        const int n = dir ? g.Nx : g.Ny;
        dim3 grid((n + 255) / 256, g.Nz), block(256);
        if (ctx->substep_f32)
            hipLaunchKernelGGL((k_ac_relax_open_boundary<float>), grid, block, 0, ctx->stream, g, (float *)F.rp, (float *)rthp_cur, F.rho_d, F.rth, dir, cb, ch, a);
        else
            hipLaunchKernelGGL((k_ac_relax_open_boundary<double>), grid, block, 0, ctx->stream, g, (double *)F.rp, rthp_cur, F.rho_d, F.rth, dir, cb, ch, a);
    };
    if (g.bounded_x && ctx->ac_open[0]) launch(0, 0, -1);
    if (g.bounded_x && ctx->ac_open[1]) launch(0, g.Nx - 1, g.Nx);
    if (g.bounded_y && ctx->ac_open[2]) launch(1, 0, -1);
    if (g.bounded_y && ctx->ac_open[3]) launch(1, g.Ny - 1, g.Ny);
}
static AcStage &stage_of(bz_ctx *ctx)
{
    static_assert(sizeof(AcStage) <= sizeof(ctx->ac_stage_storage), "AcStage does not fit its storage in bz_ctx");
    return *reinterpret_cast<AcStage *>(ctx->ac_stage_storage);
}

static void stage_buffers(bz_ctx *ctx, const AcFields &F, double *th_buf[2], double *u_buf[2], double *v_buf[2])
{
    // ping-pong buffers of the fused substep: the start buffers are chosen by the parity of N_tau so that the final
    // (rho theta)', (rho u)', (rho v)' land in the substepper's own fields (and the previous (rho theta)' in
    // previous_density_potential_temperature_perturbation, as in the reference).
    th_buf[0] = F.rthp; th_buf[1] = F.rth_old;
    u_buf[0] = F.rup; u_buf[1] = ctx->up2_user ? ctx->up2_user : ctx->d_up2;
    v_buf[0] = F.rvp; v_buf[1] = ctx->vp2_user ? ctx->vp2_user : ctx->d_vp2;
}

// apply_divergence_damping!(::DirectDivergenceDamping) on the current perturbation buffers
static int direct_damping(bz_ctx *ctx, const AcFields &F, const bz_acoustic_substepper *sub)
{
    const DevGrid &g = ctx->dg;
    const AcStage &S = stage_of(ctx);
    ProfileScope ps(ctx, "acoustic_direct_damping");
    double *th_buf[2], *u_buf[2], *v_buf[2];
    stage_buffers(ctx, F, th_buf, u_buf, v_buf);
    double *up = S.fused ? u_buf[S.cur] : (double *)F.rup, *vp = S.fused ? v_buf[S.cur] : (double *)F.rvp;
    const int extra = ctx->slab_mode ? 1 : 0;      // slab: delta from row -1
    dim3 rows((g.Nx + 255) / 256, g.Ny, g.Nz), rows_d((g.Nx + 255) / 256, g.Ny + extra, g.Nz), b256(256);
    hipLaunchKernelGGL(k_ac_direct_delta, rows_d, b256, 0, ctx->stream, g, sub->density_predictor, sub->potential_temperature, up, vp, -extra);
    if (ac_walls(ctx)) fill_walls(ctx, sub->density_predictor);      // fill_halo_regions!(delta) (acoustic_substepping.jl:1168)
    hipLaunchKernelGGL(k_ac_direct_apply, rows, b256, 0, ctx->stream, g, sub->density_predictor, sub->potential_temperature, up, vp,
                       ctx->se.damping_coefficient, S.P.wall_w, S.P.wall_s);
    BZ_LAUNCH_CHECK();
    return BZ_OK;
}

// assemble_slow_vertical_momentum_tendency! + initialize_stage_perturbations!
static int bzi_acoustic_stage_begin(bz_ctx *ctx, const bz_compressible_state *s, const bz_compressible_prognostic *U0,
                                    const bz_compressible_prognostic *G, const bz_acoustic_substepper *sub, double dt, double beta,
                                    bool store0 = false)
{
    const DevGrid &g = ctx->dg;
    AcStage &S = stage_of(ctx);
    int32_t ntau = 1;
    double dtau = 0.0;
    bz_stage_substeps(ctx, dt, beta, &ntau, &dtau);
    const double om = ctx->se.forward_weight;
    AcParams &P = S.P;
    P.dtau = dtau; P.dtn = om * dtau; P.dto = (1.0 - om) * dtau;
    P.d_new = 0.0; P.d_old = 0.0;
    S.direct = ctx->se.damping_coefficient >= 0.0 && ctx->se.direct_divergence_damping != 0;
    S.damping = ctx->se.damping_coefficient >= 0.0 && !S.direct;      // the thermal form, folded into the substep kernels
    if (S.damping && ctx->se.damp_vertical) {
        const double base = ctx->se.damping_coefficient * (ctx->dz_min * ctx->dz_min);
        P.d_new = om * base;
        P.d_old = (1.0 - om) * base;
    }
    P.f_theta = ctx->se.thermodynamic_tendency_factor;
    P.f_w = ctx->se.vertical_momentum_tendency_factor;
    const double lmin = g.flat_y ? g.dx : std::fmin(g.dx, g.dy);                             // acoustic_substepping.jl:1102-1110
    const double lfix = ctx->se.damping_length_scale;                                        // ThermalDivergenceDamping(length_scale): :1085-1092
    P.kdamp = !S.damping ? 0.0 : (lfix > 0.0) ? (ctx->se.damping_coefficient * (lfix * lfix)) / dtau : ctx->se.damping_coefficient * (lmin * lmin) / dtau;
    P.inv_N = 1.0 / (double)ntau;
    P.gate = 1.0;
    P.xcd = 0;
    P.skip_avg_if_dry = nullptr;
    P.dry_q = nullptr;
    P.wall_w = ctx->ac_open[0] ? 0 : 1;
    P.wall_s = ctx->ac_open[2] ? 0 : 1;
    P.acc_mode = 0;
    P.out_of_place = 0;
    S.ntau = ntau;
    S.done = 0;
    S.fused = ctx->ac_fused;
    S.cur = S.fused ? (ntau & 1) : 0;       // index of the buffer holding the current perturbations
    AcFields F = ac_fields(ctx, s, U0, G, sub);
    double *th_buf[2], *u_buf[2], *v_buf[2];
    stage_buffers(ctx, F, th_buf, u_buf, v_buf);
    ProfileScope ps(ctx, "acoustic_stage_init");
    AcFields Fi = F;
    Fi.rthp_out = th_buf[S.cur]; Fi.rup = u_buf[S.cur]; Fi.rvp = v_buf[S.cur];
    dim3 rows((g.Nx + 255) / 256, g.Ny, g.Nz), b256(256);
    // first stage of a whole step (the caller passes store0): the state is U0 — store_initial_state! rides along
    S.fwd2 = S.fused && ac_forward2_ok(ctx);
    // stages 1 and 2 of a whole step on a single device (the caller says so through ctx->ac_skip_avg): see AcParams::skip_avg_if_dry
    if (ctx->ac_skip_avg && S.fwd2 && !ctx->slab_mode) P.skip_avg_if_dry = bzi_moisture_state(ctx);
    if (ctx->ac_whole_step && S.fwd2 && !ctx->slab_mode) P.dry_q = bzi_moisture_state(ctx);
    // the fold costs a stage four words per cell (R G_ru, G_rv; W Gp_ru, Gp_rv) and saves every substep one (p^L): stages of >= 5 substeps
    // (the 512 x 512 x 256 benchmark: 6, 9, 18; the supercell shape of configs[4]: 2, 3, 5 — its first two stages keep p^L in the substep)
    S.pfold = S.fwd2 && ctx->d_Gp_ru && (ntau >= 5 || ctx->tune.ac_pfold > 1);      // (y-slabs: the kernel folds the row above the slab too)
    // the stage's first sweeps form its initial perturbations themselves (default variant of k_ac_forward2 on a single device)
    S.init_mode = (S.fwd2 && ctx->tune.ac_init_fold && forward2_cfg(ctx) == 29) ? (store0 ? 2 : 1) : 0;      // (y-slabs too: U0 and U carry exchanged halo rows)
    // buffer rotation (compressible_step_body): the caller passed the state arrays themselves as U0 — nothing to copy; the perturbations
    // U0 - U are (+0) by subtraction where a kernel forms them, and known zeros to the folded first sweeps
    const bool copy0 = store0 && !ctx->ac_rotate;
    if (S.pfold && ctx->slab_mode) rows.y = g.Ny + 1;
    if (S.init_mode) {
        if (S.pfold && copy0) AC_LAUNCH0(k_ac_stage_init, false COMMA true COMMA true COMMA true COMMA, rows, b256, Fi);
        else if (S.pfold) AC_LAUNCH0(k_ac_stage_init, false COMMA false COMMA true COMMA true COMMA, rows, b256, Fi);
        else if (copy0) AC_LAUNCH0(k_ac_stage_init, false COMMA true COMMA false COMMA true COMMA, rows, b256, Fi);
        else AC_LAUNCH0(k_ac_stage_init, false COMMA false COMMA false COMMA true COMMA, rows, b256, Fi);
    }
    else if (S.pfold && copy0) AC_LAUNCH0(k_ac_stage_init, false COMMA true COMMA true COMMA false COMMA, rows, b256, Fi);
    else if (S.pfold) AC_LAUNCH0(k_ac_stage_init, false COMMA false COMMA true COMMA false COMMA, rows, b256, Fi);
    else if (S.fused && copy0) AC_LAUNCH0(k_ac_stage_init, false COMMA true COMMA false COMMA false COMMA, rows, b256, Fi);
    else if (S.fused) AC_LAUNCH0(k_ac_stage_init, false COMMA false COMMA false COMMA false COMMA, rows, b256, Fi);
    else AC_LAUNCH0(k_ac_stage_init, true COMMA false COMMA false COMMA false COMMA, rows, b256, Fi);
    BZ_LAUNCH_CHECK();
    return BZ_OK;
}

// the forward sweep of a fused substep through k_ac_forward2: first substep of the stage / damping of the previous substep / folded p^L
// gradient, at the register budget for MW waves per SIMD (BZ_AC_MW)
template <bool PF, int CFG>
static void launch_forward2_cfg(bz_ctx *ctx, const AcFields &Fs, const AcParams &P, dim3 cols, dim3 bcol, bool first, bool damp, int init = 0)
{
    const DevGrid &g = ctx->dg;
    if constexpr (CFG == 29) {
        if (first && init) {
            if (ctx->substep_f32) {
                if (init == 2) hipLaunchKernelGGL((k_ac_forward2<true, false, PF, CFG, float, 2>), cols, bcol, 0, ctx->stream, g, ac_cast<float>(Fs), P);
                else hipLaunchKernelGGL((k_ac_forward2<true, false, PF, CFG, float, 1>), cols, bcol, 0, ctx->stream, g, ac_cast<float>(Fs), P);
            } else {
                if (init == 2) hipLaunchKernelGGL((k_ac_forward2<true, false, PF, CFG, double, 2>), cols, bcol, 0, ctx->stream, g, Fs, P);
                else hipLaunchKernelGGL((k_ac_forward2<true, false, PF, CFG, double, 1>), cols, bcol, 0, ctx->stream, g, Fs, P);
            }
            return;
        }
    }
    if (first) AC_LAUNCH(k_ac_forward2, true COMMA false COMMA PF COMMA CFG COMMA, cols, bcol, Fs, P);
    else if (damp) AC_LAUNCH(k_ac_forward2, false COMMA true COMMA PF COMMA CFG COMMA, cols, bcol, Fs, P);
    else AC_LAUNCH(k_ac_forward2, false COMMA false COMMA PF COMMA CFG COMMA, cols, bcol, Fs, P);
}
template <bool PF>
static void launch_forward2_pf(bz_ctx *ctx, const AcFields &Fs, const AcParams &P, dim3 cols, dim3 bcol, bool first, bool damp, int cfg, int init)
{
    if (cfg == 29) { launch_forward2_cfg<PF, 29>(ctx, Fs, P, cols, bcol, first, damp, init); return; }
    switch (cfg) {
    case 0: launch_forward2_cfg<PF, 0>(ctx, Fs, P, cols, bcol, first, damp); break;
    case 1: launch_forward2_cfg<PF, 1>(ctx, Fs, P, cols, bcol, first, damp); break;
    case 2: launch_forward2_cfg<PF, 2>(ctx, Fs, P, cols, bcol, first, damp); break;
    case 4: launch_forward2_cfg<PF, 4>(ctx, Fs, P, cols, bcol, first, damp); break;
    case 5: launch_forward2_cfg<PF, 5>(ctx, Fs, P, cols, bcol, first, damp); break;
    case 8: launch_forward2_cfg<PF, 8>(ctx, Fs, P, cols, bcol, first, damp); break;
    case 12: launch_forward2_cfg<PF, 12>(ctx, Fs, P, cols, bcol, first, damp); break;
    case 13: launch_forward2_cfg<PF, 13>(ctx, Fs, P, cols, bcol, first, damp); break;
    case 22: launch_forward2_cfg<PF, 22>(ctx, Fs, P, cols, bcol, first, damp); break;
    case 28: launch_forward2_cfg<PF, 28>(ctx, Fs, P, cols, bcol, first, damp); break;
    case 6: launch_forward2_cfg<PF, 6>(ctx, Fs, P, cols, bcol, first, damp); break;
    default: launch_forward2_cfg<PF, 29>(ctx, Fs, P, cols, bcol, first, damp); break;
    }
}
static void launch_forward2(bz_ctx *ctx, const AcFields &Fs, const AcParams &P, dim3 cols, dim3 bcol, bool first, bool damp, bool pfold, int cfg, int init)
{
    if (pfold) launch_forward2_pf<true>(ctx, Fs, P, cols, bcol, first, damp, cfg, init);
    else launch_forward2_pf<false>(ctx, Fs, P, cols, bcol, first, damp, cfg, init);
}

// substep `sstep` (1-based) of the stage opened by bzi_acoustic_stage_begin
static int bzi_acoustic_substep(bz_ctx *ctx, const bz_compressible_state *s, const bz_compressible_prognostic *U0,
                                const bz_compressible_prognostic *G, const bz_acoustic_substepper *sub, int sstep)
{
    const DevGrid &g = ctx->dg;
    AcStage &S = stage_of(ctx);
    if (sstep != S.done + 1 || sstep > S.ntau) {
        ctx->last_error = "bz_acoustic_substep: substeps must be issued in order 1..N_tau after bz_acoustic_stage_begin";
        return BZ_ERR_INVALID;
    }
    AcParams P = S.P;
    const int ntau = S.ntau;
    const bool gate = ctx->se.apply_first_substep_pressure_gradient || (sstep != 1) || (ntau == 1);
    P.gate = gate ? 1.0 : 0.0;
    const bool damp = S.damping && sstep > 1;
    AcFields F = ac_fields(ctx, s, U0, G, sub);
    dim3 rows((g.Nx + 255) / 256, g.Ny, g.Nz), b256(256);
    dim3 cols((g.Nx + ACX - 1) / ACX, (g.Ny + ACY - 1) / ACY), bcol(ACX, ACY);
    P.xcd = (ctx->tune.ac_xcd && cols.y % 8 == 0) ? 1 : 0;
    dim3 colsb((g.Nx + ABX - 1) / ABX, (g.Ny + ABY - 1) / ABY), bcolb(ABX, ABY);
    const bool walls = ac_walls(ctx);
    if (S.fused) {
        double *th_buf[2], *u_buf[2], *v_buf[2];
        stage_buffers(ctx, F, th_buf, u_buf, v_buf);
        const int cur = S.cur;
        // Bounded x / y: fill_halo_regions! of the current (rho theta)' (acoustic_substepping.jl:798,1538); the previous one, read by the
        // damping, is the buffer that was current — and filled — one substep ago
        if (walls) fill_walls_st(ctx, th_buf[cur]);
        AcFields Fs = F;
        Fs.rthp = th_buf[cur]; Fs.rth_old = th_buf[cur ^ 1]; Fs.rthp_out = th_buf[cur ^ 1];
        Fs.rup_in = u_buf[cur]; Fs.rup = u_buf[cur ^ 1];
        Fs.rvp_in = v_buf[cur]; Fs.rvp = v_buf[cur ^ 1];
        {
            ProfileScope ps(ctx, "acoustic_horizontal+column_forward");
            if (S.fwd2) {
                const int cfg = forward2_cfg(ctx);
                const int bt = (cfg & 1) ? 512 : 256;
                const int fx = ctx->tune.ac_bx == 512 && bt == 512 ? 512 : ctx->tune.ac_bx == 256 ? 256 : ctx->tune.ac_bx == 128 ? 128 : 64, fy = bt / fx;
                dim3 cols2((g.Nx + fx - 1) / fx, (g.Ny + fy - 1) / fy), bcol2(fx, fy);
                P.xcd = (ctx->tune.ac_xcd && cols2.y % 8 == 0) ? 1 : 0;
                // <u>, <v> in pairs counted from the stage's last substep (AcParams::acc_mode): substep 1 starts the accumulators, an odd
                // substep out (substep 2) is added alone
                if (ctx->tune.ac_pair_avg && !S.direct && !ctx->substep_f32 && sstep >= 2)      // (y-slabs too: a row accumulates its own faces)
                    P.acc_mode = ((ntau - sstep) & 1) ? 1 : (sstep >= 3 ? 2 : 0);
                launch_forward2(ctx, Fs, P, cols2, bcol2, sstep == 1, damp, S.pfold, cfg, S.init_mode);
            }
            else if (sstep == 1)
                AC_LAUNCH(k_ac_column_forward, true COMMA true COMMA false COMMA, cols, bcol, Fs, P);
            else if (damp)
                AC_LAUNCH(k_ac_column_forward, false COMMA true COMMA true COMMA, cols, bcol, Fs, P);
            else
                AC_LAUNCH(k_ac_column_forward, false COMMA true COMMA false COMMA, cols, bcol, Fs, P);
        }
        {
            ProfileScope ps(ctx, "acoustic_column_backward");
            if (sstep == 1 && S.init_mode) {
                if (ctx->substep_f32) {
                    if (S.init_mode == 2) hipLaunchKernelGGL((k_ac_column_backward<true, float, 2>), colsb, bcolb, 0, ctx->stream, g, ac_cast<float>(Fs), P);
                    else hipLaunchKernelGGL((k_ac_column_backward<true, float, 1>), colsb, bcolb, 0, ctx->stream, g, ac_cast<float>(Fs), P);
                } else {
                    if (S.init_mode == 2) hipLaunchKernelGGL((k_ac_column_backward<true, double, 2>), colsb, bcolb, 0, ctx->stream, g, Fs, P);
                    else hipLaunchKernelGGL((k_ac_column_backward<true, double, 1>), colsb, bcolb, 0, ctx->stream, g, Fs, P);
                }
            }
            else if (sstep == 1) AC_LAUNCH(k_ac_column_backward, true COMMA, colsb, bcolb, Fs, P);
            else AC_LAUNCH(k_ac_column_backward, false COMMA, colsb, bcolb, Fs, P);
        }
        if (walls) relax_open_boundaries(ctx, F, th_buf[cur ^ 1]);
        S.cur ^= 1;
    } else {
        if (walls) { fill_walls_st(ctx, (double *)F.rthp); fill_walls_st(ctx, (double *)F.rth_old); }
        {
            ProfileScope ps(ctx, "acoustic_horizontal");
            if (damp)
                AC_LAUNCH(k_ac_horizontal, true COMMA true COMMA, rows, b256, F, P);
            else
                AC_LAUNCH(k_ac_horizontal, false COMMA true COMMA, rows, b256, F, P);
        }
        {
            ProfileScope ps(ctx, "acoustic_column_forward");
            if (sstep == 1)
                AC_LAUNCH(k_ac_column_forward, true COMMA false COMMA false COMMA, cols, bcol, F, P);
            else
                AC_LAUNCH(k_ac_column_forward, false COMMA false COMMA false COMMA, cols, bcol, F, P);
        }
        {
            ProfileScope ps(ctx, "acoustic_column_backward");
            AC_LAUNCH(k_ac_column_backward, false COMMA, colsb, bcolb, F, P);
        }
        if (walls) relax_open_boundaries(ctx, F, (double *)F.rthp);
    }
    S.done = sstep;
    // DirectDivergenceDamping closes every substep (also the last one) on the current perturbation buffers; on a y-slab the driver
    // exchanges (rho u)', (rho v)' first and then calls bz_acoustic_direct_damping
    if (S.direct && !ctx->slab_mode) return direct_damping(ctx, F, sub);
    BZ_LAUNCH_CHECK();
    return BZ_OK;
}

extern "C" int bz_acoustic_direct_damping(bz_ctx *ctx, const bz_compressible_state *s, const bz_compressible_prognostic *U0,
                                          const bz_compressible_prognostic *G, const bz_acoustic_substepper *sub)
{
    BZ_REQUIRE_COMPRESSIBLE();
    if (!valid_state(s) || !valid_prog(U0) || !valid_prog(G) || !valid_sub(sub)) return BZ_ERR_INVALID;
    AcStage &S = stage_of(ctx);
    if (!S.direct || !ctx->slab_mode) return BZ_OK;      // single-GPU contexts damp inside bz_acoustic_substep
    if (S.done < 1) { ctx->last_error = "bz_acoustic_direct_damping: no substep has run in this stage"; return BZ_ERR_INVALID; }
    AcFields F = ac_fields(ctx, s, U0, G, sub);
    return direct_damping(ctx, F, sub);
}

// last substep's damping + time-averaged velocities + recovery of the full state [+ WS-RK3 moisture update]
// [+ the halo fills / compute_velocities! that close the reference's loop]
static int bzi_acoustic_stage_end(bz_ctx *ctx, const bz_compressible_state *s, const bz_compressible_prognostic *U0,
                                  const bz_compressible_prognostic *G, const bz_acoustic_substepper *sub, double dt,
                                  double beta, bool moist, bool velocities)
{
    const DevGrid &g = ctx->dg;
    AcStage &S = stage_of(ctx);
    if (S.done != S.ntau) {
        ctx->last_error = "bz_acoustic_stage_end: the stage still has substeps to run";
        return BZ_ERR_INVALID;
    }
    AcFields F = ac_fields(ctx, s, U0, G, sub);
    dim3 rows((g.Nx + 255) / 256, g.Ny, g.Nz), b256(256);
    if (ac_walls(ctx)) { fill_walls_st(ctx, (double *)F.rthp); if (!S.fused) fill_walls_st(ctx, (double *)F.rth_old); }
    {
        ProfileScope ps(ctx, "acoustic_finalize");
        if (S.damping)
            AC_LAUNCH(k_ac_finalize, true COMMA, rows, b256, F, S.P);
        else
            AC_LAUNCH(k_ac_finalize, false COMMA, rows, b256, F, S.P);
    }
    {
        ProfileScope ps(ctx, "acoustic_recover");
        const long long per_level = (long long)g.Ny * g.Sx;
        dim3 grid((unsigned)((per_level + 255) / 256), g.Nz);
        if (moist && g.microphysics == 2)
            AC_LAUNCH(k_ac_recover, 2 COMMA, grid, b256, F, beta * dt);
        else if (moist)
            AC_LAUNCH(k_ac_recover, 1 COMMA, grid, b256, F, beta * dt);
        else
            AC_LAUNCH(k_ac_recover, 0 COMMA, grid, b256, F, beta * dt);
    }
    if (velocities) {
        ProfileScope ps(ctx, "acoustic_velocities");
        DiagFields D = diag_fields(ctx, s, sub);
        hipLaunchKernelGGL((k_cmp_diagnose<false, false>), rows, b256, 0, ctx->stream, g, D, 0.0, 0);
    }
    S.ntau = 0;
    BZ_LAUNCH_CHECK();
    return BZ_OK;
}

// The stage epilogue of the whole-step seam as k_ac_stage_end + k_ac_recover_density: finalize + recover (with the WS-RK3 moisture
// update) + update_state! [+ the next stage's linearisation].  Single-device contexts, thermal or no divergence damping.
static bool stage_end_fusable(const bz_ctx *ctx)
{
    return !ctx->slab_mode && !ac_walls(ctx) && !ctx->tune.no_ac_end_fuse && !(ctx->se.damping_coefficient >= 0.0 && ctx->se.direct_divergence_damping != 0);
}
static int bzi_acoustic_stage_end_fused(bz_ctx *ctx, const bz_compressible_state *s, const bz_compressible_prognostic *U0,
                                        const bz_compressible_prognostic *G, const bz_acoustic_substepper *sub, double dt, double beta,
                                        bool with_linearization, const bz_compressible_state *s_out = nullptr)
{
    // s_out: the recovered prognostic fields go to this state's arrays instead of in place (buffer rotation of compressible_step_body)
    const bool oop = s_out && s_out->rho_d != s->rho_d;
    const DevGrid &g = ctx->dg;
    AcStage &S = stage_of(ctx);
    if (S.done != S.ntau) {
        ctx->last_error = "bz_acoustic_stage_end: the stage still has substeps to run";
        return BZ_ERR_INVALID;
    }
    AcFields F = ac_fields(ctx, s, U0, G, sub);      // F.thL: the linearisation this stage ran on
    DiagFields D = diag_fields(ctx, oop ? s_out : s, sub);
    AcParams Pe = S.P;
    Pe.out_of_place = oop ? 1 : 0;
    double *thL_out = ctx->thL_alt ? sub->potential_temperature : ctx->d_thL2;
    dim3 rows((g.Nx + 255) / 256, g.Ny, g.Nz), b256(256);
    const double na = ctx->se.newton_abstol, dts = beta * dt;
    const int nm = ctx->se.newton_maxiter;
    {
        ProfileScope ps(ctx, with_linearization ? "acoustic_stage_end+update_state+linearization" : "acoustic_stage_end+update_state");
#define BZ_END(DAMP, LIN, MP)                                                                                                              \
    do {                                                                                                                                   \
        if (ctx->substep_f32)                                                                                                              \
            hipLaunchKernelGGL((k_ac_stage_end<DAMP, LIN, MP, float>), rows, b256, 0, ctx->stream, g, ac_cast<float>(F), D, Pe, dts, (float *)thL_out, na, nm); \
        else hipLaunchKernelGGL((k_ac_stage_end<DAMP, LIN, MP, double>), rows, b256, 0, ctx->stream, g, F, D, Pe, dts, thL_out, na, nm);  \
    } while (0)
#define BZ_END_MP(DAMP, LIN)                                \
    do {                                                    \
        if (g.microphysics == 2) BZ_END(DAMP, LIN, 2);      \
        else if (g.microphysics == 1) BZ_END(DAMP, LIN, 1); \
        else BZ_END(DAMP, LIN, 0);                          \
    } while (0)
        if (S.damping) { if (with_linearization) BZ_END_MP(true, true); else BZ_END_MP(true, false); }
        else { if (with_linearization) BZ_END_MP(false, true); else BZ_END_MP(false, false); }
#undef BZ_END_MP
#undef BZ_END
    }
    if (!oop) {      // (out of place the density went out with the other fields: nothing reads the output set)
        ProfileScope ps(ctx, "acoustic_recover_density");
        if (ctx->substep_f32) hipLaunchKernelGGL((k_ac_recover_density<float>), rows, b256, 0, ctx->stream, g, s->rho_d, (const float *)F.rp);
        else hipLaunchKernelGGL((k_ac_recover_density<double>), rows, b256, 0, ctx->stream, g, s->rho_d, (const double *)F.rp);
    }
    if (with_linearization) ctx->thL_alt = !ctx->thL_alt;
    S.ntau = 0;
    BZ_LAUNCH_CHECK();
    return BZ_OK;
}

// acoustic_rk3_substep_loop!; moist: fold the WS-RK3 moisture update into the recovery kernel; velocities: finish with the
// halo fills + compute_velocities! of the reference (skipped when a full update_state! follows immediately).
static int bzi_acoustic_substep_loop(bz_ctx *ctx, const bz_compressible_state *s, const bz_compressible_prognostic *U0,
                                     const bz_compressible_prognostic *G, const bz_acoustic_substepper *sub, double dt,
                                     double beta, bool moist, bool velocities)
{
    int rc = bzi_acoustic_stage_begin(ctx, s, U0, G, sub, dt, beta);
    if (rc) return rc;
    const int ntau = stage_of(ctx).ntau;
    for (int sstep = 1; sstep <= ntau; ++sstep)
        if ((rc = bzi_acoustic_substep(ctx, s, U0, G, sub, sstep))) return rc;
    return bzi_acoustic_stage_end(ctx, s, U0, G, sub, dt, beta, moist, velocities);
}

static int check_loop_args(bz_ctx *ctx, const bz_compressible_state *s, const bz_compressible_prognostic *U0,
                           const bz_compressible_prognostic *G, const bz_acoustic_substepper *sub)
{
    if (!valid_state(s) || !valid_prog(U0) || !valid_prog(G) || !valid_sub(sub)) return BZ_ERR_INVALID;
    if (!ctx->fused_ok) { ctx->last_error = "compressible path needs Nx >= 2Hx and Ny >= 2Hy"; return BZ_ERR_UNSUPPORTED; }
    return BZ_OK;
}

static int require_no_slab(bz_ctx *ctx, const char *what)
{
    if (ctx->slab_mode) {
        ctx->last_error = std::string(what) + ": a y-slab context needs the distributed driver (halo exchanges between the pieces)";
        return BZ_ERR_UNSUPPORTED;
    }
    return BZ_OK;
}

extern "C" int bz_acoustic_stage_begin(bz_ctx *ctx, const bz_compressible_state *s, const bz_compressible_prognostic *U0,
                                       const bz_compressible_prognostic *G, const bz_acoustic_substepper *sub, double dt,
                                       double beta, int32_t *n_substeps, int32_t *current_buffer)
{
    BZ_REQUIRE_COMPRESSIBLE();
    int rc = check_loop_args(ctx, s, U0, G, sub);
    if (rc) return rc;
    // a caller that drives the slab exchanges itself packs rows with the grid's row and plane strides (bz_pack_rows): the Float32 working
    // fields of substep_floattype = Float32 have half-width rows, which only the library-owned communicator's packer knows
    if (ctx->slab_mode && ctx->substep_f32 && !ctx->comm) {
        ctx->last_error = "bz_acoustic_stage_begin: substep_float_bytes = 4 on a y-slab needs the library-owned communicator (bz_comm_init_*): the per-substep "
                          "halo rows of the Float32 working fields are packed by it";
        return BZ_ERR_UNSUPPORTED;
    }
    rc = bzi_acoustic_stage_begin(ctx, s, U0, G, sub, dt, beta);
    if (n_substeps) *n_substeps = stage_of(ctx).ntau;
    if (current_buffer) *current_buffer = stage_of(ctx).cur;
    return rc;
}

extern "C" int bz_acoustic_substep(bz_ctx *ctx, const bz_compressible_state *s, const bz_compressible_prognostic *U0,
                                   const bz_compressible_prognostic *G, const bz_acoustic_substepper *sub, int32_t substep,
                                   int32_t *current_buffer)
{
    BZ_REQUIRE_COMPRESSIBLE();
    int rc = check_loop_args(ctx, s, U0, G, sub);
    if (rc) return rc;
    rc = bzi_acoustic_substep(ctx, s, U0, G, sub, substep);
    if (current_buffer) *current_buffer = stage_of(ctx).cur;
    return rc;
}

extern "C" int bz_acoustic_stage_end(bz_ctx *ctx, const bz_compressible_state *s, const bz_compressible_prognostic *U0,
                                     const bz_compressible_prognostic *G, const bz_acoustic_substepper *sub, double dt,
                                     double beta, int update_moisture)
{
    BZ_REQUIRE_COMPRESSIBLE();
    int rc = check_loop_args(ctx, s, U0, G, sub);
    if (rc) return rc;
    return bzi_acoustic_stage_end(ctx, s, U0, G, sub, dt, beta, update_moisture != 0, false);
}

// Which lateral sides of a Bounded x / y carry an active open boundary condition on the wall-normal momentum (is_active_open_bc,
// acoustic_substepping.jl:1318): their outermost cells of rho', (rho theta)' are relaxed every substep with factor
// open_boundary_relaxation in (0, 1] (SplitExplicitTimeDiscretization(open_boundary_relaxation = 0.5)), and their west / south wall face
// is not zeroed; every other side of a Bounded direction is an impenetrable wall (the default).
extern "C" int bz_set_acoustic_lateral_boundaries(bz_ctx *ctx, int west_open, int east_open, int south_open, int north_open,
                                                  double open_boundary_relaxation)
{
    if (ctx) ++ctx->config_epoch;
    BZ_REQUIRE_COMPRESSIBLE();
    if (!(open_boundary_relaxation > 0.0) || open_boundary_relaxation > 1.0) {
        ctx->last_error = "bz_set_acoustic_lateral_boundaries: open_boundary_relaxation must be in (0, 1]";
        return BZ_ERR_INVALID;
    }
    const DevGrid &g = ctx->dg;
    if (((west_open || east_open) && !g.bounded_x) || ((south_open || north_open) && !g.bounded_y)) {
        ctx->last_error = "bz_set_acoustic_lateral_boundaries: an open side needs a Bounded topology in its direction";
        return BZ_ERR_INVALID;
    }
    ctx->ac_open[0] = west_open != 0; ctx->ac_open[1] = east_open != 0; ctx->ac_open[2] = south_open != 0; ctx->ac_open[3] = north_open != 0;
    ctx->ac_open_relax = open_boundary_relaxation;
    return BZ_OK;
}

extern "C" int bz_set_acoustic_scratch(bz_ctx *ctx, double *momentum_u_second_buffer, double *momentum_v_second_buffer)
{
    if (ctx) ++ctx->config_epoch;      // captured steps (bz_graph.hip) belong to one configuration
    BZ_REQUIRE_COMPRESSIBLE();
    if ((momentum_u_second_buffer == nullptr) != (momentum_v_second_buffer == nullptr)) return BZ_ERR_INVALID;
    ctx->up2_user = momentum_u_second_buffer;
    ctx->vp2_user = momentum_v_second_buffer;
    return BZ_OK;
}

extern "C" int bz_compute_moisture_tendency(bz_ctx *ctx, const bz_compressible_state *s, const bz_compressible_prognostic *G,
                                            const bz_acoustic_substepper *sub)
{
    BZ_REQUIRE_COMPRESSIBLE();
    BZ_REJECT_WALLS("bz_compute_moisture_tendency");
    if (!valid_state(s) || !valid_prog(G) || !valid_sub(sub)) return BZ_ERR_INVALID;
    // (WENO order 5 kernels; the generic order 7 / 9 path evaluates the field whatever it holds)
    int rc = launch_scalar_rho3d(ctx, "moisture_tendency", G->rho_q, nullptr, s->rho, sub->time_averaged_u, sub->time_averaged_v,
                                 sub->time_averaged_w, s->q, nullptr, nullptr, nullptr, ctx->weno_R == 3 ? bzi_moisture_state(ctx) : nullptr);
    if (rc || ctx->dg.microphysics != 2) return rc;
    const bz_kessler_model_fields &K = ctx->kessler;      // the Kessler species ride the same transport velocities
    rc = launch_scalar_rho3d(ctx, "kessler_species_tendencies", K.G_cloud_liquid_density, nullptr, s->rho, sub->time_averaged_u,
                             sub->time_averaged_v, sub->time_averaged_w, K.cloud_liquid_mass_fraction, nullptr, nullptr, nullptr);
    if (rc) return rc;
    return launch_scalar_rho3d(ctx, "kessler_species_tendencies", K.G_rain_density, nullptr, s->rho, sub->time_averaged_u,
                               sub->time_averaged_v, sub->time_averaged_w, K.rain_mass_fraction, nullptr, nullptr, nullptr);
}

extern "C" int bz_acoustic_substep_loop(bz_ctx *ctx, const bz_compressible_state *s, const bz_compressible_prognostic *U0,
                                        const bz_compressible_prognostic *G, const bz_acoustic_substepper *sub, double dt,
                                        double beta)
{
    BZ_REQUIRE_COMPRESSIBLE();
    int rc = check_loop_args(ctx, s, U0, G, sub);
    if (rc) return rc;
    if ((rc = require_no_slab(ctx, "bz_acoustic_substep_loop"))) return rc;
    // Bounded x / y: the loop ends with _recover_full_state!; the halo fills with the model's boundary conditions and compute_velocities! that
    // close the reference's function (acoustic_substepping.jl:1584-1587) are the caller's
    return bzi_acoustic_substep_loop(ctx, s, U0, G, sub, dt, beta, false, !ac_walls(ctx));
}

extern "C" int bz_acoustic_rk3_substep(bz_ctx *ctx, const bz_compressible_state *s, const bz_compressible_prognostic *U0,
                                       const bz_compressible_prognostic *G, const bz_acoustic_substepper *sub, double dt,
                                       double beta)
{
    BZ_REQUIRE_COMPRESSIBLE();
    BZ_REJECT_WALLS("bz_acoustic_rk3_substep");
    int rc = check_loop_args(ctx, s, U0, G, sub);
    if (rc) return rc;
    if ((rc = require_no_slab(ctx, "bz_acoustic_rk3_substep"))) return rc;
    rc = bz_refresh_linearization(ctx, s, sub);
    if (rc) return rc;
    rc = bz_compute_slow_tendencies(ctx, s, G);
    if (rc) return rc;
    return bzi_acoustic_substep_loop(ctx, s, U0, G, sub, dt, beta, true, true);
}

// buffer rotation of the whole-step seam: the z-halo levels of the six prognostic fields (the step's kernels write the first one of the centre
// fields and none of rho w's; the deeper ones never), state arrays -> U0 arrays; blockIdx.y: level pair (below / above), blockIdx.z: field
// (5 = rho w, whose upper halo sits one level higher)
struct RotPtrs { double *dst[6]; const double *src[6]; };
__global__ __launch_bounds__(256) void k_copy_deep_zhalo(DevGrid g, RotPtrs R)
{
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
    if (t >= g.Sxy) return;
    const int h = (int)(blockIdx.y >> 1), f = blockIdx.z;
    const bool up = blockIdx.y & 1;
    const long long lev = up ? (long long)(g.Hz + g.Nz + h + (f == 5 ? 1 : 0)) : (long long)(g.Hz - 1 - h);
    R.dst[f][lev * g.Sxy + t] = R.src[f][lev * g.Sxy + t];
}

// store_initial_state! (prognostic fields incl. the Kessler species into timestepper.U0)
int bzi_compressible_store_initial_state(bz_ctx *ctx, const bz_compressible_state *s, const bz_compressible_prognostic *U0)
{
    const DevGrid &g = ctx->dg;
    ProfileScope ps(ctx, "store_initial_state");
    const size_t nc = (size_t)g.Sxy * (size_t)(g.Nz + 2 * g.Hz) * sizeof(double);
    const size_t nf = (size_t)g.Sxy * (size_t)(g.Nz + 1 + 2 * g.Hz) * sizeof(double);
    BZ_HIP(hipMemcpyAsync(U0->rho_d, s->rho_d, nc, hipMemcpyDeviceToDevice, ctx->stream));
    BZ_HIP(hipMemcpyAsync(U0->rho_u, s->rho_u, nc, hipMemcpyDeviceToDevice, ctx->stream));
    BZ_HIP(hipMemcpyAsync(U0->rho_v, s->rho_v, nc, hipMemcpyDeviceToDevice, ctx->stream));
    BZ_HIP(hipMemcpyAsync(U0->rho_w, s->rho_w, nf, hipMemcpyDeviceToDevice, ctx->stream));
    BZ_HIP(hipMemcpyAsync(U0->rho_theta, s->rho_theta, nc, hipMemcpyDeviceToDevice, ctx->stream));
    BZ_HIP(hipMemcpyAsync(U0->rho_q, s->rho_q, nc, hipMemcpyDeviceToDevice, ctx->stream));
    if (g.microphysics == 2) {
        const bz_kessler_model_fields &K = ctx->kessler;
        BZ_HIP(hipMemcpyAsync(K.U0_cloud_liquid_density, K.cloud_liquid_density, nc, hipMemcpyDeviceToDevice, ctx->stream));
        BZ_HIP(hipMemcpyAsync(K.U0_rain_density, K.rain_density, nc, hipMemcpyDeviceToDevice, ctx->stream));
    }
    return BZ_OK;
}

static int compressible_step_body(bz_ctx *ctx, const bz_compressible_state *s, const bz_compressible_prognostic *U0,
                                  const bz_compressible_prognostic *G, const bz_acoustic_substepper *sub, double dt);

extern "C" int bz_time_step_compressible(bz_ctx *ctx, const bz_compressible_state *s, const bz_compressible_prognostic *U0,
                                         const bz_compressible_prognostic *G, const bz_acoustic_substepper *sub, double dt)
{
    BZ_REQUIRE_COMPRESSIBLE();
    BZ_REJECT_WALLS("bz_time_step_compressible");
    int rc = check_loop_args(ctx, s, U0, G, sub);
    if (rc) return rc;
    if ((rc = bzi_scan_moisture_field(ctx, s->rho_q))) return rc;      // dry models: the moisture tendency kernels write exact zeros without reading
    if (ctx->slab_mode && ctx->comm) return bzi_dist_time_step_compressible(ctx, s, U0, G, sub, dt);     // bz_comm.hip owns the exchanges
    if ((rc = require_no_slab(ctx, "bz_time_step_compressible"))) return rc;
    // launch-bound grids replay the recorded step (bz_graph.hip); a failed recording has executed nothing and falls through
    const uint64_t key = bzi_graph_key(ctx, 2, dt, s, sizeof(*s), U0, sizeof(*U0), G, sizeof(*G), sub, sizeof(*sub));
    bool capture = false;
    if (bzi_graph_begin(ctx, key, &capture) == 1) return BZ_OK;
    if (capture) {
        rc = compressible_step_body(ctx, s, U0, G, sub, dt);
        if ((rc = bzi_graph_end(ctx, key, rc)) != -1) return rc;
    }
    return compressible_step_body(ctx, s, U0, G, sub, dt);
}

static int compressible_step_body(bz_ctx *ctx, const bz_compressible_state *s, const bz_compressible_prognostic *U0,
                                  const bz_compressible_prognostic *G, const bz_acoustic_substepper *sub, double dt)
{
    int rc;
    const DevGrid &g = ctx->dg;
    BZ_REJECT_WALLS("bz_time_step_compressible");
    // round 4: on single-device contexts the stage epilogue is one pass (k_ac_stage_end) and store_initial_state! rides on the first
    // stage's initialisation kernel (the state IS U0 there); BZ_NO_AC_END_FUSE=1 restores the separate passes
    const bool fuse_end = stage_end_fusable(ctx);
    const bool store0 = fuse_end && ctx->ac_fused;
    // Round 6, buffer rotation (the anelastic lean seam's scheme, bz_step.hip: bzi_lean_stage): nothing is copied into U0.  The state arrays
    // stay intact as "U0" until the last writer of the step: stage 1 reads them as U^L and writes its recovered state into the U0 ARRAYS
    // (out of place: k_ac_stage_end writes rho_d with the other fields, no separate density pass), stages 2 and 3 run on those with
    // U0 := the state arrays, and stage 3's epilogue writes the final state back into the state arrays (a thread reads U0 only at its own
    // cell, before it writes it).  12 words of store_initial_state! and two density passes less per step; the U0 arrays are the time
    // stepper's scratch (as in the reference, where nothing reads U0 outside time_step!).  Kessler species keep their copies.
    const bool rotate = store0 && ctx->tune.ac_rotate;
    bz_compressible_state sB = *s;                 // the state with its six prognostic fields living in the U0 arrays
    bz_compressible_prognostic uS = *U0;           // "U0" = the state arrays
    if (rotate) {
        sB.rho_d = U0->rho_d; sB.rho_theta = U0->rho_theta; sB.rho_u = U0->rho_u; sB.rho_v = U0->rho_v; sB.rho_w = U0->rho_w; sB.rho_q = U0->rho_q;
        uS.rho_d = s->rho_d; uS.rho_theta = s->rho_theta; uS.rho_u = s->rho_u; uS.rho_v = s->rho_v; uS.rho_w = s->rho_w; uS.rho_q = s->rho_q;
        // z-halo levels the step's kernels do not write: the U0 arrays get the state's, once per step
        {
            ProfileScope ps(ctx, "store_initial_state");
            RotPtrs R;
            R.dst[0] = U0->rho_d; R.dst[1] = U0->rho_theta; R.dst[2] = U0->rho_u; R.dst[3] = U0->rho_v; R.dst[4] = U0->rho_q; R.dst[5] = U0->rho_w;
            R.src[0] = s->rho_d; R.src[1] = s->rho_theta; R.src[2] = s->rho_u; R.src[3] = s->rho_v; R.src[4] = s->rho_q; R.src[5] = s->rho_w;
            hipLaunchKernelGGL(k_copy_deep_zhalo, dim3((unsigned)((g.Sxy + 255) / 256), 2 * g.Hz, 6), dim3(256), 0, ctx->stream, g, R);
        }
    }
    if (store0) {
        if (g.microphysics == 2) {      // the species' U0 copies stay copies (k_ac_stage_init does not know them)
            const bz_kessler_model_fields &K = ctx->kessler;
            const size_t nc = (size_t)g.Sxy * (size_t)(g.Nz + 2 * g.Hz) * sizeof(double);
            ProfileScope ps(ctx, "store_initial_state");
            BZ_HIP(hipMemcpyAsync(K.U0_cloud_liquid_density, K.cloud_liquid_density, nc, hipMemcpyDeviceToDevice, ctx->stream));
            BZ_HIP(hipMemcpyAsync(K.U0_rain_density, K.rain_density, nc, hipMemcpyDeviceToDevice, ctx->stream));
        }
    } else if ((rc = bzi_compressible_store_initial_state(ctx, s, U0))) return rc;
    // freeze_linearization_state! (acoustic_substepping.jl:288-292): the linearisation of stage 1 (refreshed again by
    // prepare_acoustic_cache! from the same state).  Its second half, seed_time_averaged_velocities!, is deliberately not
    // issued here: nothing between this point and stage 1's substep loop reads the time-averaged velocities (the slow
    // rho theta tendency of stage 1 uses model.velocities, the stage-1 moisture tendency was built by the previous
    // update_state!), and the first substep of the loop assigns the three accumulators — the seed is
    // unobservable inside a whole step (tests/test_gpu_compressible.py::test_whole_step_matches_operator_sequence runs the
    // per-operator sequence WITH the seed against this seam).  Per-operator drivers call bz_seed_time_averaged_velocities.
    rc = bz_refresh_linearization(ctx, s, sub);
    if (rc) return rc;
    const double betas[3] = {1.0 / 3.0, 1.0 / 2.0, 1.0};
    const bz_compressible_state *s_step = s;
    const bz_compressible_prognostic *U0_step = U0;
    for (int st = 0; st < 3; ++st) {
        // rotation: stage 1 runs on the state arrays (U0 = the same arrays), stages 2 and 3 on the U0 arrays (U0 = the state arrays)
        const bz_compressible_state *s = (rotate && st > 0) ? &sB : s_step;
        const bz_compressible_prognostic *U0 = rotate ? &uS : U0_step;
        const bz_compressible_state *s_next = !rotate ? s : (st == 2 ? s_step : &sB);
        rc = bz_compute_slow_tendencies(ctx, s, G);
        if (rc) return rc;
        if (!fuse_end) {
            rc = bzi_acoustic_substep_loop(ctx, s, U0, G, sub, dt, betas[st], true, false);
            if (rc) return rc;
            // update_state! (+ prepare_acoustic_cache! of the next stage: same inputs, folded into the diagnosis kernel)
            rc = bzi_compressible_update_state(ctx, s, G, sub, true, st < 2);
            if (rc) return rc;
            continue;
        }
        ctx->ac_skip_avg = st < 2;      // the averages of stages 1 and 2 feed only the (skipped) moisture tendency of a dry model
        ctx->ac_whole_step = true;
        ctx->ac_rotate = rotate;
        rc = bzi_acoustic_stage_begin(ctx, s, U0, G, sub, dt, betas[st], store0 && st == 0);
        ctx->ac_rotate = false;
        ctx->ac_skip_avg = false;
        ctx->ac_whole_step = false;
        if (rc) return rc;
        const int ntau = stage_of(ctx).ntau;
        for (int sstep = 1; sstep <= ntau; ++sstep)
            if ((rc = bzi_acoustic_substep(ctx, s, U0, G, sub, sstep))) return rc;
        if ((rc = bzi_acoustic_stage_end_fused(ctx, s, U0, G, sub, dt, betas[st], st < 2, s_next))) return rc;
        if ((rc = bz_compute_moisture_tendency(ctx, s_next, G, sub))) return rc;
    }
    s = s_step;
    if (g.microphysics == 2) return bz_compressible_kessler_update(ctx, s, G, sub, dt);     // microphysics_model_update! (:316)
    return BZ_OK;
}

// microphysics_model_update!(::DCMIP2016KesslerMicrophysics, model) for CompressibleDynamics: density = dynamics_density
// (rho_d), pressure = dynamics.pressure (dcmip2016_kessler.jl:460-485), then update_state!(model)
extern "C" int bz_compressible_kessler_update(bz_ctx *ctx, const bz_compressible_state *s, const bz_compressible_prognostic *G,
                                              const bz_acoustic_substepper *sub, double dt)
{
    BZ_REQUIRE_COMPRESSIBLE();
    if (!valid_state(s) || !valid_prog(G) || !valid_sub(sub)) return BZ_ERR_INVALID;
    if (ctx->dg.microphysics != 2) { ctx->last_error = "bz_compressible_kessler_update: no Kessler microphysics attached"; return BZ_ERR_INVALID; }
    const bz_kessler_model_fields &K = ctx->kessler;
    bz_kessler_fields F;
    F.density = s->rho_d; F.pressure = s->p;
    F.potential_temperature = s->theta; F.potential_temperature_density = s->rho_theta;
    F.moisture_density = s->rho_q; F.cloud_liquid_density = K.cloud_liquid_density; F.rain_density = K.rain_density;
    F.vapor_mass_fraction = K.vapor_mass_fraction; F.cloud_liquid_mass_fraction = K.cloud_liquid_mass_fraction;
    F.rain_mass_fraction = K.rain_mass_fraction; F.rain_terminal_velocity = K.rain_terminal_velocity;
    F.precipitation_rate = K.precipitation_rate;
    int rc = bz_kessler_microphysics_update(ctx, &ctx->kessler_params, &F, dt, ctx->kessler_pst);
    if (rc) return rc;
    // the columns are rank-local; on a y-slab the update_state! that follows needs the neighbour exchanges of the driver
    if (ctx->slab_mode) return BZ_OK;
    return bzi_compressible_update_state(ctx, s, G, sub, true, false);
}
