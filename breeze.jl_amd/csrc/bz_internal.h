// bz_internal.h — shared host/device declarations of libbreeze_hip (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <hipfft/hipfft.h>

#include <cstdint>
#include <cstdio>
#include <string>
#include <vector>

#include "../../include/breeze_hip.h"

// A column table (Nz + 2 Hz + 1 entries, written once at bz_create).  On the device `table[k]` reads through the CONSTANT address
// space: with a wave-uniform level index that is an s_load from the scalar cache, batched by the compiler at the top of a level,
// instead of a vector load with its own s_waitcnt (the generic-pointer form could not be scalarised because the tables might alias
// the kernel's stores: the ISA of the tendency kernels had 4-7 such loads per level, each followed by a full wait).
struct ColPtr {
    const double *p = nullptr;
    ColPtr() = default;
    __host__ __device__ ColPtr(const double *q) : p(q) {}
    __host__ __device__ operator const double *() const { return p; }
#ifdef __HIPCC__
    typedef const double __attribute__((address_space(4))) *cptr;
    __device__ __forceinline__ double operator[](int k) const { return ((cptr)p)[k]; }
    __device__ __forceinline__ double operator[](long long k) const { return ((cptr)p)[k]; }
#endif
};

// Device view of the grid + reference columns.  Passed by value (kernarg) to every kernel.
// Column pointers are pre-offset so that index k (0-based interior) is valid for
// k = -Hz .. Nz+Hz-1 (centres) / Nz+Hz (faces).
struct DevGrid {
    int Nx, Ny, Nz;
    int Hx, Hy, Hz;
    int Sx, Sy;            // parent row length / rows per plane
    long long Sxy;         // plane stride
    double dx, dy, rdx, rdy, Az;
    ColPtr dzc, dzf, rdzf;               // thickness at centres; centre spacing at faces; 1/dzf
    ColPtr rdzc;                         // 1/dzc
    ColPtr zc;                           // cell-centre heights
    ColPtr Ax, Ay;                       // dy*dzc[k], dx*dzc[k]
    ColPtr Vinv_c, Vinv_f;               // 1/(dx*dy*dzc[k]), 1/(dx*dy*dzf[k])
    ColPtr rho, rho_f;                   // rho_r at centres;  0.5*(rho[k-1]+rho[k]) at faces
    ColPtr rrho, rrho_f;                 // their correctly rounded reciprocals (host 1.0/x), for bz_cdiv
    ColPtr p_r, T_r;
    ColPtr pi_dry, lnpi;                 // dry Exner factor (p_r[k]/p_st)^(Rd/cpd) (device pow, bz_tendency5.hip: k_pi_dry) and ln(p_r[k]/p_st): bz_exner_factor
    double kap_num;                      // Rv cpd - Rd cpv
    double g, Rd, Rv, cpd, cpv, pst;
    int formulation;       // 0: liquid-ice potential temperature (theta), 1: static energy (e) in the `theta` slots
    // microphysics = SaturationAdjustment(equilibrium = WarmPhaseEquilibrium()) (bz_set_saturation_adjustment)
    int microphysics;      // 0: nothing (q = q^v), 1: warm-phase saturation adjustment (q = q^e; q^v, q^l diagnosed),
                           // 2: DCMIP2016 Kessler (q = q^v; rho q^cl, rho q^r prognostic)
    int sa_maxiter;
    double sa_Ll, sa_cl, sa_dc, sa_L0, sa_Ttr, sa_ptr, sa_abstol;   // dc = cpv - cl, L0 = Ll - dc * T_energy
    double *qv_field, *ql_field;                                     // model.microphysical_fields.q^v, q^l (parents)
    // microphysics == 2: DCMIP2016KesslerMicrophysics — prognostic rho q^cl, rho q^r, diagnostic q^cl, q^r (qv_field = mu.q^v)
    double *rqcl_field, *rqr_field, *qcl_field, *qr_field;
    int wrap_y;            // 1: y halos are this rank's own periodic images; 0: y-slab, halos filled by the neighbour ranks
    int flat_y;            // 1: topology (Periodic, Flat, Bounded): Ny = 1, Hy = 0, no y neighbours (per-operator kernels only)
    int bounded_x;         // 1: topology (Bounded, Flat, Bounded): walls in x of a 2-D x-z model (the reference's cloudy_thermal_bubble.jl): per-lane
                           // WENO / Centered buffers in the per-operator kernels, u / rho u with wall faces i = 0, Nx, cosine transform along x
    int bounded_y;         // 1: topology (Periodic, Bounded, Bounded): walls in y (wrap_y = 0: y neighbours are halo rows — a no-flux row for
                           // fields that are centres in y, impenetrable wall faces j = 0, Ny for rho v and v; WENO / Centered buffers by row;
                           // cosine transform along y in the pressure solve; per-operator kernels only)

    __host__ __device__ inline long long idx(int i, int j, int k) const {
        return (long long)(i + Hx) + (long long)Sx * ((long long)(j + Hy)) + Sxy * (long long)(k + Hz);
    }
};

// Optional SSP-RK3 epilogue of a tendency kernel (ssp_runge_kutta_3.jl:167-173): instead of storing the tendency G,
// store  u_new = (1-alpha) u0 + alpha (u_old + dt G).  mode 0: store G;  1: first stage (alpha = 1; also writes
// u0 = u_old, i.e. store_initial_state!);  2: later stages (reads u0).
struct RKEpilogue {
    int mode = 0;
    double dt = 0.0, alpha = 0.0, oma = 0.0;
    const double *u0 = nullptr;     // mode 2
    double *u0_out = nullptr;       // mode 1
    const double *u0b = nullptr;    // second field of a fused pair
    double *u0b_out = nullptr;
};
#ifdef __HIPCC__
// a / b for a column constant b whose correctly rounded reciprocal rb = RN(1/b) is tabulated: q0 = a rb, one fused
// residual, one fused correction (Markstein's division: the result is the correctly rounded quotient, i.e. bit-identical
// to the IEEE `a / b` the diagnosis kernels use — checked on 4e8 random pairs, tools/check_cdiv.c), 3 instructions instead
// of the ~12 of a full FP64 division.  Lets the lean tendency kernels derive u = rho_u / rho_r(k) etc. on the fly.
__device__ __forceinline__ double bz_cdiv(double a, double b, double rb)
{
    const double q0 = a * rb;
    return fma(fma(-q0, b, a), rb, q0);
}
__device__ __forceinline__ double bz_rk_apply(int mode, double dt, double alpha, double oma, const double *u0,
                                              double *u0_out, double G, double uold, long long n)
{
    if (mode == 0) return G;
    double u0v;
    if (mode == 1) { u0_out[n] = uold; u0v = uold; }
    else u0v = u0[n];
    return oma * u0v + alpha * (uold + dt * G);
}
// the same update with u0 already in a register (loaded at the top of the level by the caller so that its latency overlaps the
// flux arithmetic): u0v is ignored in the first stage
__device__ __forceinline__ double bz_rk_apply_pre(int mode, double dt, double alpha, double oma, double u0v, double *u0_out, double G,
                                                  double uold, long long n)
{
    if (mode == 1) { u0_out[n] = uold; u0v = uold; }
    else if (mode == 3) u0v = uold;      // first stage of the lean seam: the stage-start array itself stays intact as U0 (bz_step.hip), nothing is stored
    return oma * u0v + alpha * (uold + dt * G);
}
#endif

#ifdef __HIPCC__
// Exner factor Pi^(Rm / cpm) of a moist cell in the anelastic model, Pi = p_r[k] / p_st (T = Pi^(Rm/cpm) theta,
// /root/reference/src/Thermodynamics/dynamic_states.jl:31-58).  The base is a COLUMN constant and only the exponent varies with the
// cell, so pow() — ~150 FP64 instructions of extended-precision log + exp, which round 4 parked behind a call in the z-momentum kernel
// (2.0 -> 2.8 ms per launch at 512^3 as soon as a wavefront held vapour) — is one exp() of a small argument:
//     Pi^(Rm/cpm) = Pi^kappa_d exp((Rm/cpm - kappa_d) ln Pi),   Rm/cpm - kappa_d = q (Rv cpd - Rd cpv) / (cpm cpd)   (exact algebra, no cancellation)
// with Pi^kappa_d and ln Pi tabulated per level.  q = 0 gives exp(0) = 1: the dry table entry, bit for bit, so dry and moist cells of one
// field are continuous and the wave-uniform dry branch of the lean kernels is an optimisation, not a different formula.  |argument| <
// 0.07 q |ln Pi|: a few 1e-3, so the result is within ~1.5 ulp of pow() (the oracle's form) — tests/test_gpu_parity.py holds T to 1e-14.
// Every kernel of the theta formulation that derives T without microphysics uses this one function (k_thermo, k_project_diagnose<0>,
// the lean z-momentum kernel), so stored and derived temperatures carry the same bits.
// exp(a) of a small argument: the degree-9 Taylor polynomial in Horner form on |a| <= 1/16 (truncation a^10 / 10! < 3e-19: nine FMAs,
// no extra registers — inlined, the library exp() pushed the z-momentum kernel past its 128 registers, and behind a call it cost spills
// around the call).  For atmospheric q and pressures |a| ~ 0.07 q |ln Pi| is a few 1e-3.  Larger arguments (q -> 1 at stratospheric
// pressures: unphysical, but the function must not be wrong there) are halved lane by lane until they fit and the result is squared
// back: exp(a) = exp(a / 2^s)^(2^s), each squaring doubling the relative error (s <= 5 for |a| <= 2).
// a Float64 literal as a scalar-register pair materialised where it is used: without it the compiler hoists the nine coefficients below
// out of the level loop of the z-momentum kernel into eighteen VGPRs, which that kernel (128 of 128) pays for with scratch spills
__device__ __forceinline__ double bz_sconst(double c)
{
    asm volatile("" : "+s"(c));
    return c;
}
__device__ __forceinline__ double bz_exp_small(double a)
{
    int s = 0;      // per lane (round 6, ADVICE r05: a wave-uniform count made a cell's result depend on the cells that share its wavefront)
    while (fabs(a) > 0.0625 && s < 64) { a *= 0.5; ++s; }
    double p = bz_sconst(1.0 / 362880.0);
    p = fma(p, a, bz_sconst(1.0 / 40320.0));
    p = fma(p, a, bz_sconst(1.0 / 5040.0));
    p = fma(p, a, bz_sconst(1.0 / 720.0));
    p = fma(p, a, bz_sconst(1.0 / 120.0));
    p = fma(p, a, bz_sconst(1.0 / 24.0));
    p = fma(p, a, bz_sconst(1.0 / 6.0));
    p = fma(p, a, 0.5);
    p = fma(p, a, 1.0);
    p = fma(p, a, 1.0);
    for (; s > 0; --s) p *= p;
    return p;
}
__device__ __forceinline__ double bz_exner_factor(const DevGrid &g, int k, double q, double cpm)
{
    return g.pi_dry[k] * bz_exp_small(((q * g.kap_num) / (cpm * g.cpd)) * g.lnpi[k]);
}
// ---- moist thermodynamics shared by the diagnosis and buoyancy kernels -----------------------------------------------
// anelastic buoyancy -g rho' with rho' = rho_r (R_m,r T_r / (R_m T) - 1)  (anelastic_buoyancy.jl:36-72); the moisture
// fractions come from grid_moisture_fractions: (q, 0) without microphysics, (q^v, q^l) fields with saturation adjustment.
__device__ __forceinline__ double bz_buoyancy(const DevGrid &g, const double *__restrict__ T, const double *__restrict__ q,
                                              long long n, int k)
{
    double qv, ql = 0.0;
    if (g.microphysics == 1) { qv = g.qv_field[n]; ql = g.ql_field[n]; }
    else if (g.microphysics == 2) { qv = q[n]; ql = g.qcl_field[n] + g.qr_field[n]; }
    else qv = q[n];
    const double Rm = (1.0 - (qv + ql)) * g.Rd + qv * g.Rv;
    const double rhop = g.rho[k] * (g.Rd * g.T_r[k] / (Rm * T[n]) - 1.0);
    return -g.g * rhop;
}
// Warm-phase saturation adjustment of a liquid-ice potential temperature state (saturation_adjustment.jl:168-235 with
// clausius_clapeyron.jl:59-68, vapor_saturation.jl:250-256, Solvers.jl:243-262): returns T, sets qv, ql.
// Kessler moisture state: q = (q^v, q^cl + q^r), T = Pi(q) theta + L q^l / c_pm  (dcmip2016_kessler.jl:222-227,298-303;
// dynamic_states.jl:46-58)
__device__ __forceinline__ double bz_kessler_T(const DevGrid &g, double th, double qv, double ql, double pr)
{
    const double qd = 1.0 - (qv + ql);
    const double cpm = qd * g.cpd + qv * g.cpv + ql * g.sa_cl;
    return pow(pr / g.pst, (qd * g.Rd + qv * g.Rv) / cpm) * th + (g.sa_Ll * ql) / cpm;
}
__device__ __forceinline__ double bz_sa_psat(const DevGrid &g, double T)
{
    return g.sa_ptr * pow(T / g.sa_Ttr, g.sa_dc / g.Rv) * exp((1.0 / g.sa_Ttr - 1.0 / T) * g.sa_L0 / g.Rv);
}
__device__ __forceinline__ double bz_sa_T(const DevGrid &g, double th, double qv, double ql, double pr)
{
    const double qd = 1.0 - (qv + ql);
    const double cpm = qd * g.cpd + qv * g.cpv + ql * g.sa_cl;
    return pow(pr / g.pst, (qd * g.Rd + qv * g.Rv) / cpm) * th + (g.sa_Ll * ql) / cpm;
}
__device__ __forceinline__ void bz_sa_adjust(const DevGrid &g, double T, double qt, double pr, double &qv, double &ql)
{
    const double ps = bz_sa_psat(g, T);
    const double qs = (g.Rd / g.Rv) * (1.0 - qt) * ps / (pr - ps);
    ql = fmax(0.0, qt - qs);
    qv = qt - ql;
}
__device__ __forceinline__ double bz_sa_residual(const DevGrid &g, double T, double th, double qt, double pr)
{
    double qv, ql;
    bz_sa_adjust(g, T, qt, pr, qv, ql);
    return T - bz_sa_T(g, th, qv, ql, pr);
}
__device__ __forceinline__ double bz_sa_diagnose(const DevGrid &g, double th, double qt, double pr, double &qv, double &ql)
{
    qv = qt; ql = 0.0;
    const double T1 = bz_sa_T(g, th, qt, 0.0, pr);
    if (th == 0.0) return T1;
    const double rho1 = pr / (((1.0 - qt) * g.Rd + qt * g.Rv) * T1);
    if (qt <= bz_sa_psat(g, T1) / (rho1 * g.Rv * T1)) return T1;
    double qv1, ql1;
    bz_sa_adjust(g, T1, qt, pr, qv1, ql1);
    const double dT = (g.sa_Ll * ql1) / ((1.0 - (qv1 + ql1)) * g.cpd + qv1 * g.cpv + ql1 * g.sa_cl);
    double x1 = T1, x2 = T1 + fmax(0.01, dT / 2.0);
    double r1 = bz_sa_residual(g, x1, th, qt, pr), r2 = bz_sa_residual(g, x2, th, qt, pr);
    for (int it = 0; it < g.sa_maxiter && fabs(r2) > g.sa_abstol; ++it) {
        double s = (x2 - x1) / (r2 - r1);
        const bool valid = isfinite(s);
        s = valid ? s : 0.0;
        x1 = x2; r1 = r2;
        x2 -= r2 * s;
        r2 = valid ? bz_sa_residual(g, x2, th, qt, pr) : 0.0;
    }
    bz_sa_adjust(g, x2, qt, pr, qv, ql);
    return bz_sa_T(g, th, qv, ql, pr);
}
// ---- density-based liquid-ice potential temperature state of CompressibleDynamics (LiquidIceDensityState) ----------------------------
// temperature(::LiquidIceDensityState): Newton on T = (rho R_m T / p_st)^kappa theta + L  (dynamic_states.jl:197-232)
__device__ __forceinline__ double bz_ds_temperature(const DevGrid &g, double th, double qv, double ql, double rho, double abstol,
                                                    int maxiter)
{
    const double qd = 1.0 - (qv + ql);
    const double Rm = qd * g.Rd + qv * g.Rv;
    const double cpm = qd * g.cpd + qv * g.cpv + ql * g.sa_cl;
    const double kap = Rm / cpm, gam = cpm / (cpm - Rm);
    const double L = (g.sa_Ll * ql) / cpm;
    double T = pow(th, gam) * pow(rho * Rm / g.pst, gam - 1.0) + L;
    double dT = T;
    for (int it = 0; it < maxiter && fabs(dT) > abstol; ++it) {
        const double Phi = pow(rho * Rm * T / g.pst, kap) * th;
        dT = -(T - Phi - L) / (1.0 - kap * Phi / T);
        T += dT;
    }
    return T;
}
// saturated_density_residual (saturation_adjustment.jl:236-253): theta^li(T) - theta0 of the state saturated at its own density
__device__ __forceinline__ double bz_ds_residual(const DevGrid &g, double T, double th0, double rho, double qt, double &qv, double &ql)
{
    const double qs = bz_sa_psat(g, T) / (rho * g.Rv * T);
    ql = fmax(0.0, qt - qs);
    qv = qt - ql;
    const double qd = 1.0 - (qv + ql);
    const double Rm = qd * g.Rd + qv * g.Rv;
    const double cpm = qd * g.cpd + qv * g.cpv + ql * g.sa_cl;
    const double L = (g.sa_Ll * ql) / cpm;
    const double p = rho * Rm * T;
    return (T - L) * pow(g.pst / p, Rm / cpm) - th0;
}
// adjust_thermodynamic_state(::LiquidIceDensityState, ::SaturationAdjustment) (saturation_adjustment.jl:264-301) -> T; sets qv, ql
__device__ __forceinline__ double bz_ds_adjust(const DevGrid &g, double th, double qt, double rho, double nabstol, int nmaxiter,
                                               double &qv, double &ql)
{
    qv = qt; ql = 0.0;
    if (th == 0.0) return 0.0;
    const double T1 = bz_ds_temperature(g, th, qt, 0.0, rho, nabstol, nmaxiter);
    if (qt <= bz_sa_psat(g, T1) / (rho * g.Rv * T1)) return T1;
    double qv1, ql1;
    bz_ds_residual(g, T1, th, rho, qt, qv1, ql1);
    const double dT = (g.sa_Ll * ql1) / ((1.0 - (qv1 + ql1)) * g.cpd + qv1 * g.cpv + ql1 * g.sa_cl);
    double x1 = T1, x2 = T1 + fmax(0.01, dT / 2.0);
    double a, b;
    double r1 = bz_ds_residual(g, x1, th, rho, qt, a, b), r2 = bz_ds_residual(g, x2, th, rho, qt, a, b);
    for (int it = 0; it < g.sa_maxiter && fabs(r2) > g.sa_abstol; ++it) {
        double s = (x2 - x1) / (r2 - r1);
        const bool valid = isfinite(s);
        s = valid ? s : 0.0;
        x1 = x2; r1 = r2;
        x2 -= r2 * s;
        r2 = valid ? bz_ds_residual(g, x2, th, rho, qt, a, b) : 0.0;
    }
    bz_ds_residual(g, x2, th, rho, qt, qv, ql);
    return bz_ds_temperature(g, th, qv, ql, rho, nabstol, nmaxiter);
}
#endif

struct BzComm;
struct ProfileSlot {
    const char *name;
    double total_ms = 0.0;
    int64_t launches = 0;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> pending;
};

// Every BZ_* environment switch of the library, read ONCE when a context is created (bzi_read_tuning, bz_context.hip): nothing on a
// launch path calls getenv.  They select measured alternatives for A/B runs; defaults are the shipped choices.
struct bz_tuning {
    bool no_fused = false;            // BZ_NO_FUSED: per-operator sequence instead of the fused whole-step tiers
    bool no_fuse_rk = false;          // BZ_NO_FUSE_RK
    bool no_tend_lds = false;         // BZ_NO_TEND_LDS: generation-1 (untiled) tendency kernels
    int tend_gen = 0;                 // BZ_TEND_GEN (0: default)
    bool no_lean = false;             // BZ_NO_LEAN: fused-RK tier instead of the lean (prognostic-only) seam
    bool no_xcd = false;              // BZ_NO_XCD: hardware block order in the lean kernels
    bool no_closure_march = false;    // BZ_NO_CLOSURE_MARCH: the cell-per-thread closure kernels everywhere
    bool no_k6_stored = false;        // BZ_NO_K6_STORED: the stored-velocity tiers keep the fourth-generation tile kernels (k_{u,v,w}_tend_lds)
    bool no_dry_shortcut = false;     // BZ_NO_DRY_SHORTCUT: the lean kernels always carry rho q (no moisture scan)
    bool side_scalar = false;         // BZ_SIDE_SCALAR: scalar kernel beside the pressure solve on one GPU too
    int side_cus = 0;                 // BZ_SIDE_CUS=N: the side stream is created on N of the GPU's CUs (CU mask)
    int side_cu_layout = 0;           // BZ_SIDE_CU_LAYOUT (experiments): 0 the first N mask bits, 1 every (total / N)-th bit
    bool no_fuse_forcing = false;     // BZ_NO_FUSE_FORCING
    bool no_fold_forcing = false;     // BZ_NO_FOLD_FORCING: the fused-RK tier keeps the momentum terms of the forcing stack in the forcing pass
    int ac_xcd = 1;                   // BZ_AC_XCD=0: the forward acoustic sweep in launch order (XCD = tile column) instead of XCD = band of tile rows
    int ac_forward2 = 1;              // BZ_AC_FWD2=0: the round-5 forward acoustic sweep (k_ac_column_forward) instead of k_ac_forward2
    int ac_rotate = 1;                // BZ_AC_ROTATE=0: store_initial_state! copies the state into U0 instead of the buffer rotation of the whole-step seam
    int ac_pair_avg = 1;              // BZ_AC_PAIR_AVG=0: <u>, <v> accumulated in every substep instead of two substeps at a time (AcParams::acc_mode)
    int ac_init_fold = 1;             // BZ_AC_INIT_FOLD=0: k_ac_stage_init stores the stage's initial perturbations and the first sweeps read them back
    int ac_pfold = 1;                 // BZ_AC_PFOLD=0: the horizontal gradient of p^L stays in every substep instead of folded into the stage's slow tendencies
    int ac_bx = 128;                  // BZ_AC_BX: columns of a k_ac_forward2 block along x (64, 128, 256, 512; rows = threads / columns)
    int ac_cfg = 29;                  // BZ_AC_CFG: k_ac_forward2 variant (bit 0: 512-thread blocks, bit 1: register budget of three waves per SIMD, bit 2: barrier per level,
                                      // bit 3: next level's loads in flight, bit 4: DPP lane shifts); built: 0 1 2 4 5 6 8 12 13 22 28 29
    bool no_fuse_level_sums = false;  // BZ_NO_FUSE_LEVEL_SUMS: the subsidence averages always come from their own pass over u, v, theta, q
    bool no_tridiag_coop = false;     // BZ_NO_TRIDIAG_COOP: sequential Thomas kernel
    bool no_xfft = false;             // BZ_NO_XFFT: library 2-D plans instead of the hand-written x transforms
    int poisson_kxmajor = 1;          // BZ_POISSON_KXMAJOR=0: level-major half spectrum, three whole-spectrum passes between the x transforms
    int poisson_kx_chunk_mb = 0;      // BZ_POISSON_KX_CHUNK_MB: spectrum bytes per chunk of the kx-major pipeline (0: 256 MB)
    int poisson_kx_chunk_kb = 0;      // BZ_POISSON_KX_CHUNK_KB: the same in KB (tests: the chunked pipeline on small grids)
    int poisson_kx_pad = 1;           // BZ_POISSON_KX_PAD: phantom lines per wavenumber of the kx-major spectrum (0 .. 4)
    int poisson_chunk = 0;            // BZ_POISSON_CHUNK: level-chunked Poisson pipeline (needs BZ_NO_XFFT)
    int xf_kchunk_f = 0, xf_kchunk_i = 0;      // BZ_XF_KCHUNK_F / _I: levels per block of the x transforms (0: automatic)
    int scalar_lds = 1;               // BZ_SCALAR_LDS=0: the 3-D-density scalar tendency keeps the kernel that reads its stencils through the L1 (k_scalar_tendency_rho3d_x)
    bool no_rho3d_exchange = false;   // BZ_NO_RHO3D_EXCHANGE: the compressible scalar tendency keeps the kernel that evaluates five fluxes per cell
    bool no_generic_march = false;    // BZ_NO_GENERIC_MARCH: WENO 7 / 9 keep the two-pass (flux arrays + divergence) kernels everywhere
    bool generic_onepass = false;     // BZ_GENERIC_ONEPASS: WENO 7 / 9 with every flux evaluated by both of its cells
    bool no_ac_fuse = false;          // BZ_NO_AC_FUSE: three kernels per acoustic substep
    bool no_ac_end_fuse = false;      // BZ_NO_AC_END_FUSE: stage epilogue as finalize + recover + update_state (three passes) and store_initial_state as copies
    bool comm_no_overlap = false;     // BZ_COMM_NO_OVERLAP
    bool comm_self_messages = false;  // BZ_COMM_SELF_MESSAGES: world 1 sends every message to itself
    bool comm_no_side_scalar = false; // BZ_COMM_NO_SIDE_SCALAR
    int graph = -1;                   // BZ_GRAPH (-1: unset)
    bool graph_debug = false;         // BZ_GRAPH_DEBUG
};
void bzi_read_tuning(bz_tuning &t);
// (Periodic, Bounded, Bounded) contexts run the dry anelastic WENO-5 model with column forcings / bottom fluxes; everything else says so
#define BZ_REJECT_BOUNDED_Y(ctx, enabling, what)                                                                             \
    do {                                                                                                                     \
        if ((ctx) && (ctx)->dg.bounded_y && (enabling)) {                                                                                \
            (ctx)->last_error = what ": not implemented on a Bounded y (topology (Periodic, Bounded, Bounded))";            \
            return BZ_ERR_UNSUPPORTED;                                                                                       \
        }                                                                                                                    \
    } while (0)
#define BZ_HALO_XFACE 8      // the same for x faces (rho u, u) on a Bounded x
#define BZ_HALO_YFACE 4      // halo kind bit: the field sits on y faces (rho v, v): wall faces instead of a no-flux row on a Bounded y
struct bz_ctx;
bool bzi_lean_forcings_ok(const bz_ctx *ctx);

struct bz_ctx {
    bz_tuning tune;
    bz_grid grid;
    bz_constants constants;
    DevGrid dg;
    hipStream_t stream = nullptr;
    bool walls_lean_ok = false;       // (Periodic, Bounded, Bounded): the grid is large enough for the lean seam's tiles (Nx >= 2 Hx, Ny >= 2 Hy)
    int num_cus = 256;                // compute units of the device (persistent-workgroup launches size their grids with it)
    std::string last_error;

    // column tables (one device allocation)
    double *d_columns = nullptr;
    // Poisson
    int NXH = 0;                      // Nx/2+1
    hipfftHandle plan_fwd = 0, plan_inv = 0;
    hipfftDoubleComplex *d_dctx = nullptr;      // Bounded x: half spectrum of the permuted rows ((Nx/2 + 1) x Nz), between the row transform and the cosine combination
    bool plans_ok = false;
    // chunked Poisson pipeline (bz_poisson.hip): 2-D plans over `pchunk` levels, so that source term -> x transform -> y transform
    // (and y -> x -> projection on the way back) of one level range run back to back while the range sits in the 256 MiB Infinity Cache
    hipfftHandle plan_fwd_c = 0, plan_inv_c = 0;
    int pchunk = 0;
    int kr0 = 0, krn = 0;            // level range of the next source / projection launch (krn = 0: all levels)
    // hand-written x transforms with a transposed (ky-fastest) half spectrum (bz_xfft_kernels.h)
    bool xf = false;
    bool xf_slab = false;            // y-slab context whose rows the same kernels transform (bz_comm.hip: dist_poisson)
    void *d_wtab = nullptr;          // exp(-2 pi i t / Nx), t < 3 Nx / 4
    hipfftHandle plan_y = 0;         // contiguous batched 1-D transform along y of the transposed spectrum
    // round 6: kx-major half spectrum hatT[(kx Nz + k) Ny + ky] and the middle of the solve in chunks of kx_cw wavenumbers (bzi_xf_middle)
    bool kxmajor = false;
    int kx_cw = 0, kx_nch = 0, kx_pad = 0;      // wavenumbers per chunk, chunks, phantom lines per wavenumber
    hipfftHandle plan_yc = 0, plan_yc_last = 0;      // y transforms of one chunk / of the (shorter) last chunk
    int profile_mute = 0;            // > 0: ProfileScope objects record nothing (an enclosing scope covers the launches)
    // y-slab mode: 1-D batched plans of the distributed transform (bz_slab_transform, created on first use)
    hipfftHandle slab_plan_x_fwd = 0, slab_plan_x_inv = 0, slab_plan_y = 0;
    bool slab_plans_ok = false;
    int slab_inv_ld = 0;              // leading dimension (complex elements per row) the inverse x plan was built for
    double *d_rhs = nullptr;          // Nx*Ny*Nz real (source term, then inverse-transform output)
    hipfftDoubleComplex *d_hat = nullptr;   // NXH*Ny*Nz
    double *d_ibeta = nullptr;        // NXH*Ny*Nz : 1/beta_k
    double *d_tfac = nullptr;         // NXH*Ny*Nz : t_k = c_{k-1}/beta_{k-1}
    double *d_lower = nullptr;        // Nz
    double *d_scalar = nullptr;       // small scratch (mean, reductions)
    int tend_gen = 2;                 // tendency kernel generation (BZ_TEND_GEN: 1 = gen-1 everywhere, 2 = gen-1 momentum + fused scalar pair, 3 = gen-3 u,v + pair, 4 = gen-3 everywhere)
    // y-slab decomposition (bz_create_slab): this rank owns Ny rows of Ny*y_nranks and the kx block
    // [kx0, kx0+nkx) of the zero-padded half spectrum; the horizontal transforms are done by the caller.
    int y_nranks = 1, y_rank = 0, nkx = 0, kx0 = 0, Ny_global = 0;
    bool slab_mode = false;           // created by bz_create_slab (also with one rank): caller fills y halos and does the FFTs
    bool tend_lds = true;             // LDS y-tile variants of the u and w tendency kernels (BZ_NO_TEND_LDS=1 disables)
    bool fuse_rk = true;              // whole-step seam: RK update folded into the tendency kernels (BZ_NO_FUSE_RK=1 disables)
    bool G_is_predictor = false;      // after a fused step the G arrays hold predictor momentum, not tendencies
    bool fused_ok = true;             // Nx >= 2Hx && Ny >= 2Hy: fused halo-image stores are valid
    // host view of the scan's verdict (round 5): 0 unknown (after bz_update_state: the next scan is read back synchronously, once), 1 dry at
    // the last scan (a dry model is scanned at every call; its verdict travels back asynchronously), 2 moist (sticky, like the device word:
    // no more scans, and only the general kernels are launched).  Lets the dry and the general tendency bodies be separate KERNELS.
    int q_host = 0;
    int *h_qstate = nullptr;          // pinned
    hipEvent_t ev_q = nullptr;
    bool q_pending = false;
    int *d_qstate = nullptr;          // moisture scan of the lean seam (bz_step.hip: bzi_scan_moisture): 2 = rho q has a non-zero element (sticky until
                                      // update_state!), anything else after a scan = identically zero; the lean kernels then skip every access to it
    bool diagnostics_stale = false;   // u, v, w, theta, q, T, phi of `s` are older than the prognostic state (bz_time_steps_anelastic without the last diagnosis)
    bool lean_step_last = false;      // the last step body took the lean tier
    bool lean = true;                 // whole-step seam on prognostic-only kernels (bz_tendency5_kernels.h; BZ_NO_LEAN=1 disables)
    bool lean_xcd = true;             // XCD-contiguous block order of the lean kernels (BZ_NO_XCD=1 disables)
    hipStream_t side_stream = nullptr;   // the scalar-pair kernel of a stage runs here, beside the pressure solve on the main stream
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    bool side_scalar = false;         // BZ_SIDE_SCALAR=1: single-GPU seam too (the distributed step does it whenever W > 1)
    void *d_lev_rows = nullptr;       // LevRow5[Nz + 2 Hz]: the lean kernels' column constants, one 64-byte row per level (bz_tendency5.hip)
    double *d_pi_dry = nullptr;       // (p_r[k]/p_st)^(Rd/cpd), k = -Hz .. Nz+Hz-1: Exner factor of a dry cell, built with the device pow()
    // CompressibleDynamics + SplitExplicitTimeDiscretization (bz_create_compressible)
    bool compressible = false;
    bool has_reference = false;       // ExnerReferenceState columns present (else p_r = rho_r = 0)
    bz_split_explicit se;
    int weno_R = 3;                   // stencil half-width of the advection scheme: WENO(order = 2 R - 1)
    int scalar_R = 3;                 // ... of the scalars' scheme when momentum_advection and scalar_advection differ (bz_set_scalar_advection_order;
                                      // examples/tropical_cyclone_world.jl:167-169: momentum WENO(order = 9), scalars WENO(order = 5)); else = weno_R
    double *d_gflux = nullptr;        // flux scratch of the generic (order 7 / 9) kernels' two-pass evaluation: 3 parent-shaped arrays
    double dz_min = 0.0;              // minimum_zspacing(grid)
    double *d_sponge = nullptr;       // UpperSponge rate * ramp per face (compressible contexts)
    double *d_Clin = nullptr;         // centre array: gamma R_m * Pi of the current linearisation
    double *d_tfac_ac = nullptr;      // centre array: Thomas factors t_k of the acoustic column system (per stage)
    double *d_up2 = nullptr, *d_vp2 = nullptr;   // second buffers of the (rho u)', (rho v)' ping-pong (fused substep)
    double *d_Gp_ru = nullptr, *d_Gp_rv = nullptr;   // slow horizontal momentum tendencies minus the gradient of the stage's p^L (k_ac_stage_init<PF>, round 6)
    double *d_thL2 = nullptr;         // second buffer of theta_L: the fused stage epilogue writes the next stage's linearisation while it still reads this one's
    bool thL_alt = false;             // the current theta_L lives in d_thL2 (whole-step seam only; every per-operator linearisation resets it)
    bool substep_f32 = false;         // substep_floattype = Float32 inside the Float64 library: the substepper's working fields are float arrays
    bool ac_whole_step = false;       // set around bzi_acoustic_stage_begin by the whole-step seam (AcParams::dry_q)
    bool ac_rotate = false;           // set around bzi_acoustic_stage_begin by the whole-step seam while its buffer rotation is on (U0 = the state arrays in stage 1)
    bool ac_skip_avg = false;         // set around bzi_acoustic_stage_begin by the whole-step seam for its stages 1 and 2 (AcParams::skip_avg_if_dry)
    bool ac_fused = true;             // horizontal step folded into the forward column sweep (BZ_NO_AC_FUSE=1 disables)
    int ac_open[4] = {0, 0, 0, 0};    // west, east, south, north side of a Bounded x / y carries an active open boundary condition (bz_set_acoustic_lateral_boundaries)
    double ac_open_relax = 0.5;       // SplitExplicitTimeDiscretization(open_boundary_relaxation)
    // DCMIP2016KesslerMicrophysics attached to the model (bz_set_kessler_microphysics)
    bz_kessler_microphysics kessler_params;
    bz_kessler_model_fields kessler;
    double kessler_pst = 1e5;
    // forcing / Coriolis / bottom-flux stack (bz_set_forcings, bz_forcing.hip)
    bool has_relaxation = false;      // bz_set_relaxation: [rate target](u v w theta q), Nz + 1 entries each
    double *d_relax = nullptr;
    int relax_mask = 0, relax_specific = 0;
    const double *field_forcing = nullptr;      // bz_set_field_forcing: 3-D forcing of the thermodynamic variable (caller-owned)
    int field_forcing_specific = 0;
    bool has_forcings = false;
    double *d_forcing = nullptr;      // static profiles, subsidence velocity, level averages, subsidence profiles, partial sums
    int forcing_static_mask = 0, forcing_subsidence_mask = 0;
    // horizontal sums of u, v, theta, q emitted by the projection + diagnosis kernel of the previous stage (bz_fused.hip: PDFields::lsum):
    // true only between that launch and the next stage's bzi_compute_forcings INSIDE one bz_time_step(s)_anelastic call (the host may
    // write the fields between calls, so nothing is trusted across them)
    double *d_lsum_rows = nullptr;
    long long lsum_P = 0;
    bool lsum_fresh = false;
    bool lsum_step_last = false;      // what the last recorded / launched step body left in lsum_fresh (graph replays restore it)
    bool fold_momentum_forcing = false;      // set by the fused-RK tier around its tendency launches: bzi_k6_stored folds the momentum terms of the stack
    double forcing_f = 0.0, forcing_flux_theta = 0.0, forcing_flux_q = 0.0, forcing_drag = 0.0, forcing_drag_eps = 0.0, forcing_flux_energy = 0.0;
    // BulkDrag / BulkSensibleHeatFlux / BulkVaporFlux bottom conditions (bz_set_bulk_surface_fluxes, bz_forcing.hip)
    bool has_bulk = false;
    bz_bulk_surface_fluxes bulk;
    // user tracers (bz_set_tracers, bz_tracers.hip)
    int n_tracers = 0;
    bz_tracer_fields tracers[BZ_MAX_TRACERS];
    // closure = SmagorinskyLilly() (bz_set_closure, bz_closure.hip)
    bool has_closure = false;
    bz_smagorinsky_lilly closure;
    double *closure_nu = nullptr;     // model.closure_fields.nu_e (caller-owned centre field)
    double *d_closure_ipi = nullptr;  // (pst/p_r[k])^(Rd/cpd), k = -1 .. Nz
    double *up2_user = nullptr, *vp2_user = nullptr;   // caller-owned replacements of d_up2 / d_vp2 (bz_set_acoustic_scratch)
    alignas(8) unsigned char ac_stage_storage[160] = {0};   // AcStage of the stage in flight (bz_compressible.hip)
    // advection = (; rho_q = WENO(order = 5, bounds = (lo, hi))) (bz_set_bounds_preserving_advection, bz_bounded.hip)
    int bounded_mask = 0;             // 1 moisture, 2 microphysical species, 4 tracers
    double bounded_lo = 0.0, bounded_hi = 1.0;
    struct BzComm *comm = nullptr;    // y-slab communicator (bz_comm_init_rccl / bz_comm_init_local, bz_comm.hip)
    // profiling
    bool profiling = false;
    std::vector<ProfileSlot> slots;
    // hipGraph replay of whole steps on launch-bound grids (bz_graph.hip)
    int graph_mode = 0;               // 0: off, 1: capture a step the second time it is asked for with the same arguments, then replay
    bool graph_capturing = false;
    uint64_t config_epoch = 0;        // bumped by every bz_set_* call: a captured step is valid for one configuration
    struct GraphSlot {
        uint64_t key = 0;
        int seen = 0;
        hipGraph_t graph = nullptr;
        hipGraphExec_t exec = nullptr;
        bool g_is_predictor = false, lean_step_last = false, lsum_step_last = false;      // host-side bookkeeping the recorded body leaves (restored on replay)
    } graph_slots[2];
    int graph_next = 0;
    hipStream_t graph_stream = nullptr, graph_user_stream = nullptr;      // recording stream (the legacy default stream cannot be captured)
    long long graph_replays = 0, graph_captures = 0;
};

// ---- bz_graph.hip ----
uint64_t bzi_graph_key(const bz_ctx *ctx, int kind, double dt, const void *a, size_t na, const void *b, size_t nb, const void *c, size_t nc,
                       const void *d, size_t nd);
// begin: 1 = the step was replayed from its graph (done), 0 = run the body now; *capture = the stream is recording, bracket the body
// with bzi_graph_end.  end: BZ_OK = recorded and launched, -1 = recording failed and nothing has executed (run the body the ordinary
// way), otherwise the body's own error code
int bzi_graph_begin(bz_ctx *ctx, uint64_t key, bool *capture);
int bzi_graph_end(bz_ctx *ctx, uint64_t key, int body_rc);
void bzi_graph_destroy(bz_ctx *ctx);
void bzi_graph_configure(bz_ctx *ctx);
int bzi_apply_stream(bz_ctx *ctx, hipStream_t stream);

#define BZ_HIP(expr)                                                                        \
    do {                                                                                    \
        hipError_t _e = (expr);                                                             \
        if (_e != hipSuccess) {                                                             \
            ctx->last_error = std::string(#expr) + ": " + hipGetErrorString(_e);            \
            return -(int)_e;                                                                \
        }                                                                                   \
    } while (0)

#define BZ_FFT(expr)                                                                        \
    do {                                                                                    \
        hipfftResult _r = (expr);                                                           \
        if (_r != HIPFFT_SUCCESS) {                                                         \
            ctx->last_error = std::string(#expr) + ": hipfft status " + std::to_string((int)_r); \
            return -1000 - (int)_r;                                                         \
        }                                                                                   \
    } while (0)

#define BZ_LAUNCH_CHECK()                                                                   \
    do {                                                                                    \
        hipError_t _e = hipGetLastError();                                                  \
        if (_e != hipSuccess) {                                                             \
            ctx->last_error = std::string("kernel launch: ") + hipGetErrorString(_e);       \
            return -(int)_e;                                                                \
        }                                                                                   \
    } while (0)

// Scoped profiling region: records start/stop events on the ctx stream when enabled.
// Bit pattern of a non-negative real for atomicMax reductions (monotone in the value).  tools/gen_f32_sources.py swaps these three
// definitions for their 32-bit counterparts in the Float32 build.
typedef unsigned long long bz_bits_t;
__device__ __forceinline__ bz_bits_t bz_real_bits(double d) { return (bz_bits_t)__double_as_longlong(d); }
__device__ __forceinline__ double bz_real_inf() { return __longlong_as_double(0x7ff0000000000000LL); }

struct ProfileScope {
    bz_ctx *ctx;
    int slot = -1;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    ProfileScope(bz_ctx *c, const char *name);
    ~ProfileScope();
};

// internal entry points shared between translation units
int bzi_fill_halo(bz_ctx *ctx, double *f, int kind);
int bzi_apply_forcings(bz_ctx *ctx, const bz_state *s, double *Gu, double *Gv, double *Gth, double *Gq, double scale, bool momentum_done = false);
int bzi_compute_forcings(bz_ctx *ctx, const bz_state *s);      // bz_compute_forcings; takes the level sums of bz_ctx::d_lsum_rows while lsum_fresh
double *bzi_level_sum_rows(bz_ctx *ctx, long long *P);          // the array the projection + diagnosis kernel emits them into (nullptr: not applicable)
bool bzi_level_sums_ride(const bz_ctx *ctx);
int bzi_flux_bc(bz_ctx *ctx, const bz_state *s, double *Gu, double *Gv, double *Gth, double *Gq, double scale);
void bzi_forcing_teardown(bz_ctx *ctx);
int bzi_tracer_specific(bz_ctx *ctx);
int bzi_tracer_rk3(bz_ctx *ctx, double dt, double alpha, bool first);
int bzi_tracer_store_initial_state(bz_ctx *ctx);
int bzi_tracer_tendencies(bz_ctx *ctx, const bz_state *s);
int bzi_momentum_advection_gen1(bz_ctx *ctx, const bz_state *s, const bz_prognostic *G);
int bzi_momentum_advection_generic(bz_ctx *ctx, const bz_state *s, const bz_prognostic *G);
int bzi_generic_tendencies_fused_rk(bz_ctx *ctx, const bz_state *s, const bz_prognostic *U0, const bz_prognostic *G, double dt, double alpha,
                                    bool first);
int bzi_scalar_tendency_generic(bz_ctx *ctx, double *Gc, const double *u, const double *v, const double *w, const double *c);
int bzi_scalar_rho3d_generic(bz_ctx *ctx, double *Gc, double *Grho, const double *rho, const double *u, const double *v, const double *w,
                             const double *c, const double *ru, const double *rv, const double *rw);
void bzi_closure_teardown(bz_ctx *ctx);
int bzi_apply_closure(bz_ctx *ctx, const bz_state *s, double *Gu, double *Gv, double *Gw, double *Gth, double *Gq, double scale);
int bzi_kessler_tendencies(bz_ctx *ctx, const bz_state *s);
int bzi_kessler_rk3(bz_ctx *ctx, double dt, double alpha, bool first);
int bzi_kessler_update(bz_ctx *ctx, const bz_state *s, const bz_prognostic *G, double dt);
int bzi_fill_halos_multi(bz_ctx *ctx, double *const *fields, const int *kinds, int n);
int bzi_poisson_setup(bz_ctx *ctx, const double *h_rho_halo /* Nz+2Hz */);
int bzi_bounded_tendencies(bz_ctx *ctx, const bz_state *s, const bz_prognostic *G);
int bzi_compute_tendencies_generic(bz_ctx *ctx, const bz_state *s, const bz_prognostic *G);
int bzi_momentum_tendencies_generic(bz_ctx *ctx, const bz_state *s, const bz_prognostic *G);
int bzi_apply_relaxation(bz_ctx *ctx, const bz_state *s, const bz_prognostic *G, const double *rho3d = nullptr);      // rho3d: coupling density of a compressible context
int bzi_lean_setup(bz_ctx *ctx);
int bzi_dist_time_step(bz_ctx *ctx, const bz_state *s, const bz_prognostic *U0, const bz_prognostic *G, double dt, bool diagnose = true);
int bzi_pack_rows_geom(bz_ctx *ctx, double *const *fields, const int32_t *levels, int32_t n, int32_t row0, int32_t nrows, double *buffer,
                       int32_t unpack, int sx, long long sxy);
int bzi_comm_join_pending(bz_ctx *ctx);
// entry points that read the stored diagnostics (u, v, w, theta, q, T): after bz_time_steps_anelastic(..., diagnose_last = 0) they are older
// than the prognostic state.  With the state at hand they are rebuilt first (bz_update_state, on slab contexts with its halo exchange);
// without it (s == nullptr) the call fails with BZ_ERR_INVALID and a message instead of answering from stale fields
int bzi_refresh_diagnostics(bz_ctx *ctx, const bz_state *s, const char *who);
void bzi_lean_step_done(bz_ctx *ctx, bool diagnosed);
int bzi_scan_moisture(bz_ctx *ctx, const bz_state *s);
int bzi_scan_moisture_field(bz_ctx *ctx, const double *rho_q);
const int *bzi_moisture_state(const bz_ctx *ctx);
void bzi_moisture_unknown(bz_ctx *ctx);      // bz_update_state / bz_compressible_update_state: the field may have been set!
struct LeanStage;
void bzi_lean_stage(const bz_state *s, const bz_prognostic *U0, const bz_prognostic *G, int stage, LeanStage *L);
int bzi_dist_time_step_compressible(bz_ctx *ctx, const bz_compressible_state *s, const bz_compressible_prognostic *U0,
                                    const bz_compressible_prognostic *G, const bz_acoustic_substepper *sub, double dt);
int bzi_compressible_store_initial_state(bz_ctx *ctx, const bz_compressible_state *s, const bz_compressible_prognostic *U0);
void bzi_comm_teardown(bz_ctx *ctx);
int bzi_comm_allreduce_sum(bz_ctx *ctx, double *buf, int n);
void bzi_lean_teardown(bz_ctx *ctx);
void bzi_poisson_teardown(bz_ctx *ctx);
int bzi_poisson_solve(bz_ctx *ctx, const bz_state *s, double dt);
int bzi_poisson_spectral(bz_ctx *ctx);
int bzi_poisson_from_momentum(bz_ctx *ctx, const bz_state *s, double dt, const bz_prognostic *predictor);
int bzi_fft_chunk(bz_ctx *ctx, int k0, bool forward);
int bzi_xf_forward(bz_ctx *ctx, const bz_state *s, double dt, const bz_prognostic *predictor, const double *rhs = nullptr,
                   double *hat = nullptr, int blocks = 1, int klo = 0, int khi = 0);      // khi = 0: all levels
int bzi_xf_inverse(bz_ctx *ctx, const double *hat = nullptr, double *phi = nullptr, int blocks = 1, int klo = 0, int khi = 0);
int bzi_xf_y(bz_ctx *ctx, bool forward);
int bzi_tridiag_launch(bz_ctx *ctx, double *hat, double scale, int Ny, int mean_column, int kx_lo = 0, int kx_hi = 0);
int bzi_xf_middle(bz_ctx *ctx);      // y transform, vertical solves, inverse y transform on ctx->d_hat (chunked by wavenumber where ctx->kxmajor)
// fused streaming kernels (bz_fused.hip)
int bzi_rk3_fused(bz_ctx *ctx, const bz_state *s, const bz_prognostic *U0, const bz_prognostic *G, double dt,
                  double alpha, bool first);
int bzi_poisson_source_fused(bz_ctx *ctx, const bz_state *s, double dt, double *rhs = nullptr,
                             const bz_prognostic *predictor = nullptr);
int bzi_project_diagnose(bz_ctx *ctx, const bz_state *s, double dt, const double *phi_c = nullptr,
                         const double *phi_below = nullptr, const bz_prognostic *predictor = nullptr, bool store_phi = true,
                         const double *rtheta_in = nullptr, const double *rq_in = nullptr, bool level_sums = false);
int bzi_project_lean(bz_ctx *ctx, const bz_state *s, double dt, const double *phi_c, const double *phi_below,
                     const bz_prognostic *predictor, double *sa, double *sb);
// lean whole-step tendencies (bz_tendency5.hip): prognostic-only inputs, rho theta / rho q advance from (pa, pb) into (oa, ob)
// Buffer rotation of the lean seam (bz_step.hip: bzi_lean_stage): where stage `stage` reads and writes
struct LeanStage {
    bz_state sin;             // `s` with rho_u, rho_v, rho_w pointing at the stage-start momentum (stage 1: the state arrays, then the U0 slots)
    bz_state sout;            // `s` with rho_u, rho_v, rho_w pointing at where the projection of the stage writes (stages 1-2: U0 slots, stage 3: state)
    bz_prognostic u0;         // the step-start fields: the state arrays themselves, intact until the last projection / scalar update of the step
    const double *pa, *pb;    // stage-start rho theta, rho q
    double *oa, *ob;          // updated rho theta, rho q
};
int bzi_tendencies_lean(bz_ctx *ctx, const bz_state *s, const bz_prognostic *U0, const bz_prognostic *G, const double *pa,
                        const double *pb, double *oa, double *ob, double dt, double alpha, bool first, int rows = 0, int which = 3);
int bzi_tendencies_fused_rk(bz_ctx *ctx, const bz_state *s, const bz_prognostic *U0, const bz_prognostic *G, double dt,
                            double alpha, bool first);
int bzi_create(bz_ctx **out, const bz_grid *grid, const bz_constants *constants, const bz_reference_state *ref,
               int weno_order, int y_nranks, int y_rank, bool slab_mode, bool compressible = false);
void bzi_compressible_teardown(bz_ctx *ctx);
int bzi_compute_tendencies3(bz_ctx *ctx, const bz_state *s, const bz_prognostic *G, bool include_w);
bool bzi_k6_stored_ok(const bz_ctx *ctx);
int bzi_k6_stored(bz_ctx *ctx, int comp, const bz_state *s, const bz_prognostic *G, const bz_prognostic *U0, const RKEpilogue *Ein, int bm);
int bzi_scalar_pair_tendency(bz_ctx *ctx, const bz_state *s, const bz_prognostic *G, const bz_prognostic *U0 = nullptr,
                             const RKEpilogue *E = nullptr);
int bzi_u_tendency_lds(bz_ctx *ctx, const bz_state *s, const bz_prognostic *G, const bz_prognostic *U0 = nullptr,
                       const RKEpilogue *E = nullptr);
int bzi_v_tendency_lds(bz_ctx *ctx, const bz_state *s, const bz_prognostic *G, const bz_prognostic *U0 = nullptr,
                       const RKEpilogue *E = nullptr);
// buoyancy_mode 0: anelastic buoyancy (T, q, reference columns); 1: none (SlowTendencyMode); 2: compressible slow
// vertical momentum, s->T = pressure, s->q = total density, reference columns p_r, rho (acoustic_substepping.jl:727-752)
int bzi_w_tendency_ring(bz_ctx *ctx, const bz_state *s, const bz_prognostic *G, const bz_prognostic *U0 = nullptr,
                        const RKEpilogue *E = nullptr, int buoyancy_mode = 0);
