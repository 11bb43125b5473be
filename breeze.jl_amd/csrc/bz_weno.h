// bz_weno.h — device-side reconstruction primitives (WENO-Z order 5 with the order-3 / order-1
// buffer schemes used next to Bounded walls, Centered(order 4/2) advecting-flux interpolation).
// Semantics: Oceananigans.Advection as called from /root/reference/src/Advection.jl:20-35 and
// /root/reference/src/AtmosphereModels/dynamics_kernel_functions.jl:54-62.
#pragma once
#include <hip/hip_runtime.h>

#define BZ_WENO_EPS 1e-8

#ifndef BZ_WENO_ONE_DIVISION
#define BZ_WENO_ONE_DIVISION 1
#endif

// cells (a,b,c,d,e), upwind cell c, value at the face between c and d.
// Reference order of operations (three quotients tau/(beta_s+eps), then normalisation).
__device__ __forceinline__ double bz_weno5_ref(double a, double b, double c, double d, double e)
{
    double b0 = c * (10.0 * c - 31.0 * d + 11.0 * e) + d * (25.0 * d - 19.0 * e) + e * (4.0 * e);
    double b1 = b * (4.0 * b - 13.0 * c + 5.0 * d) + c * (13.0 * c - 13.0 * d) + d * (4.0 * d);
    double b2 = a * (4.0 * a - 19.0 * b + 11.0 * c) + b * (25.0 * b - 31.0 * c) + c * (10.0 * c);
    double tau = fabs(b0 - b2);
    double r0 = tau / (b0 + BZ_WENO_EPS);
    double r1 = tau / (b1 + BZ_WENO_EPS);
    double r2 = tau / (b2 + BZ_WENO_EPS);
    double a0 = (3.0 / 10.0) * (1.0 + r0 * r0);
    double a1 = (3.0 / 5.0) * (1.0 + r1 * r1);
    double a2 = (1.0 / 10.0) * (1.0 + r2 * r2);
    double p0 = (1.0 / 3.0) * c + (5.0 / 6.0) * d - (1.0 / 6.0) * e;
    double p1 = -(1.0 / 6.0) * b + (5.0 / 6.0) * c + (1.0 / 3.0) * d;
    double p2 = (1.0 / 3.0) * a - (7.0 / 6.0) * b + (11.0 / 6.0) * c;
    return (a0 * p0 + a1 * p1 + a2 * p2) / (a0 + a1 + a2);
}

// Fast form used by the shipped kernels.  Same weights and candidate polynomials, rearranged for
// FP64 issue slots (46 FP64 instructions + one v_rcp_f64 against ~85 + 4 divisions of the reference order; round 2's form of the
// same idea needed 57 — measured on MI355X every FP64 instruction of a wave64 costs ~2.1 ns of SIMD time, tools/valu_rates.hip):
//  * everything is expressed through the first differences D1..D4 of the five cells: the smoothness
//    indicators (3 x Jiang-Shu, as Oceananigans tabulates them) are
//        beta_2 = 13/4 (D2-D1)^2 + 3/4 (3 D2 - D1)^2,  beta_1 = 13/4 (D3-D2)^2 + 3/4 (D2+D3)^2,
//        beta_0 = 13/4 (D4-D3)^2 + 3/4 (D4 - 3 D3)^2,
//    and the candidates are c + delta_s with delta_2 = 5/6 D2 - 1/3 D1, delta_1 = 1/3 D3 + 1/6 D2,
//    delta_0 = 2/3 D3 - 1/6 D4 (less cancellation than the cell-value form);
//  * the weights depend on the indicators only through tau / (beta_s + eps), which is unchanged when beta_s, tau and eps are all
//    divided by 3/4: d_s = (beta_s + eps) / (3/4) = L_s^2 + (13/3 S_s^2 + 4/3 eps) is two FMAs after S_s^2, and tau / (3/4) = d_0 - d_2;
//  * alpha_s = C_s (1 + tau^2/d_s^2) is multiplied through by d_0^2 d_1^2 d_2^2 (positive, so the normalised weights are
//    unchanged) and by 10: with s_j = d_j^2, Q = s_0 s_1 s_2 and P_s = Q / s_s the scaled weights are w_s = 10 C_s (Q + tau^2 P_s):
//    one division instead of four, and the constants 10 C_s = (3, 6, 1) are folded into the candidate increments
//    (3 delta_0 = 2 D3 - D4/2, 6 delta_1 = 2 D3 + D2).  FP64 range: d_s in [1e-8, ~1e6] => products in [1e-48, 1e37];
//  * that division is a v_rcp_f64 seed plus two Newton steps (denominator positive and normal, so the
//    div_scale / div_fixup special-casing is unnecessary): relative error ~1e-16.  The reference itself
//    evaluates tau/(beta+eps) with a reduced-precision reciprocal and one Newton step (newton_div).
// Differs from bz_weno5_ref by a few 1e-16 of the reconstructed value.
__device__ __forceinline__ double bz_weno5_fast(double a, double b, double c, double d, double e)
{
    const double D1 = b - a, D2 = c - b, D3 = d - c, D4 = e - d;
    const double S1 = D2 - D1, S2 = D3 - D2, S3 = D4 - D3;
    const double L2 = fma(3.0, D2, -D1), L1 = D2 + D3, L0 = fma(-3.0, D3, D4);
    const double E43 = (4.0 / 3.0) * BZ_WENO_EPS, C133 = 13.0 / 3.0;
    const double d2 = fma(L2, L2, fma(C133, S1 * S1, E43));
    const double d1 = fma(L1, L1, fma(C133, S2 * S2, E43));
    const double d0 = fma(L0, L0, fma(C133, S3 * S3, E43));
    const double tau = d0 - d2;
    const double t2 = tau * tau;
    const double s0 = d0 * d0, s1 = d1 * d1, s2 = d2 * d2;
    const double P0 = s1 * s2, P1 = s0 * s2, P2 = s0 * s1;
    const double Q = s0 * P0;
    const double a0 = fma(t2, P0, Q), a1 = fma(t2, P1, Q), a2 = fma(t2, P2, Q);      // w_0 / 3, w_1 / 6, w_2
    const double e0 = fma(2.0, D3, -0.5 * D4);                                       // 3 delta_0
    const double e1 = fma(2.0, D3, D2);                                              // 6 delta_1
    const double e2 = fma(5.0 / 6.0, D2, -(1.0 / 3.0) * D1);
    const double num = fma(a2, e2, fma(a1, e1, a0 * e0));
    const double den = fma(3.0, a0, fma(6.0, a1, a2));
    double r = __builtin_amdgcn_rcp(den);
    r = fma(fma(-den, r, 1.0), r, r);
    r = fma(fma(-den, r, 1.0), r, r);
    return fma(num, r, c);
}

// Difference form of the indicators and candidates (as bz_weno5_fast) with the reference's three quotients for the weights: no
// product of squared indicators (range-safe in Float32, where the one-division form underflows at 1e-48) and no expanded
// polynomial of the cell values (in Float32 the expanded beta of a field with a large mean — theta ~ 300 K — loses every digit:
// measured 3e-4 of the tendency scale against 2e-5 for this form).  The Float32 build uses it (BZ_WENO_ONE_DIVISION=2).
// The quotients are reciprocal-multiplies (v_rcp: 1 ulp in Float32, where this form is used; the reference's own newton_div is a
// reduced-precision reciprocal + one Newton step) — an IEEE division costs ~10 instructions, four of them per reconstruction.
// Indicators scaled by 4/3 as in bz_weno5_fast; the optimal weights (3, 6, 1) / 10 are folded into the candidate increments.
__device__ __forceinline__ double bz_weno5_diff(double a, double b, double c, double d, double e)
{
    const double D1 = b - a, D2 = c - b, D3 = d - c, D4 = e - d;
    const double S1 = D2 - D1, S2 = D3 - D2, S3 = D4 - D3;
    const double L2 = fma(3.0, D2, -D1), L1 = D2 + D3, L0 = fma(-3.0, D3, D4);
    const double E43 = (4.0 / 3.0) * BZ_WENO_EPS, C133 = 13.0 / 3.0;
    const double d2 = fma(L2, L2, fma(C133, S1 * S1, E43));
    const double d1 = fma(L1, L1, fma(C133, S2 * S2, E43));
    const double d0 = fma(L0, L0, fma(C133, S3 * S3, E43));
    const double tau = fabs(d0 - d2);
    const double r0 = tau * __builtin_amdgcn_rcp(d0), r1 = tau * __builtin_amdgcn_rcp(d1), r2 = tau * __builtin_amdgcn_rcp(d2);
    const double a0 = fma(r0, r0, 1.0), a1 = fma(r1, r1, 1.0), a2 = fma(r2, r2, 1.0);
    const double e0 = fma(2.0, D3, -0.5 * D4);                  // 3 delta_0
    const double e1 = fma(2.0, D3, D2);                         // 6 delta_1
    const double e2 = fma(5.0 / 6.0, D2, -(1.0 / 3.0) * D1);
    const double num = fma(a2, e2, fma(a1, e1, a0 * e0));
    const double den = fma(3.0, a0, fma(6.0, a1, a2));
    return fma(num, __builtin_amdgcn_rcp(den), c);
}

// ---- the open parity hypothesis ------------------------------------------------------------
// Recent Oceananigans carries a second float type in its WENO scheme, WENO{N, FT, FT2} with FT2 = Float32 by default, "for the weight
// computation".  What exactly is evaluated in FT2 cannot be read here (Oceananigans is not vendored), so two variants exist beside the
// default (everything in the grid's FT), selected at build time (lib/libbreeze_hip_ft2_<level>.so, make ft2) and by og_set_weno_ft2()
// in the oracle:
//   BZ_WENO_FT2 = 1  the quotients tau / (beta_s + eps) through newton_div(FT2, a, b): reciprocal of b in Float32, one Newton step in FT
//                    (x = a * inv; x += (a - x * b) * inv) — beta, tau, alpha and the normalisation stay in FT.  Differs from the
//                    default by ~1e-14 of the reconstructed value: a golden file could not tell them apart at the 1e-12 tolerance.
//   BZ_WENO_FT2 = 2  beta, tau, alpha and the normalised weights in Float32 (inputs rounded to Float32 for the indicators), the
//                    candidate polynomials and their weighted sum in FT.  Differs from the default by ~1e-7 of the cell-to-cell
//                    variation: the case in which a perfect Float64 port would sit at 1e-7, not 1e-12, from CPU().
// Reference operation order (as bz_weno5_ref) in both.  tools/dump_goldens.jl records typeof(advection) and
// tests/test_reference_goldens.py reports which variant a golden file matches.
#ifndef BZ_WENO_FT2
#define BZ_WENO_FT2 0
#endif
__device__ __forceinline__ double bz_newton_div(double a, double b)
{
    const double inv = (double)(1.0f / (float)b);
    double x = a * inv;
    x = x + (a - x * b) * inv;
    return x;
}
__device__ __forceinline__ double bz_weno5_ft2(double a, double b, double c, double d, double e)
{
    const double p0 = (1.0 / 3.0) * c + (5.0 / 6.0) * d - (1.0 / 6.0) * e;
    const double p1 = -(1.0 / 6.0) * b + (5.0 / 6.0) * c + (1.0 / 3.0) * d;
    const double p2 = (1.0 / 3.0) * a - (7.0 / 6.0) * b + (11.0 / 6.0) * c;
#if BZ_WENO_FT2 == 2
    const float af = (float)a, bf = (float)b, cf = (float)c, df = (float)d, ef = (float)e;
    const float b0 = cf * (10.0f * cf - 31.0f * df + 11.0f * ef) + df * (25.0f * df - 19.0f * ef) + ef * (4.0f * ef);
    const float b1 = bf * (4.0f * bf - 13.0f * cf + 5.0f * df) + cf * (13.0f * cf - 13.0f * df) + df * (4.0f * df);
    const float b2 = af * (4.0f * af - 19.0f * bf + 11.0f * cf) + bf * (25.0f * bf - 31.0f * cf) + cf * (10.0f * cf);
    const float tau = fabsf(b0 - b2);
    const float r0 = tau / (b0 + 1e-8f), r1 = tau / (b1 + 1e-8f), r2 = tau / (b2 + 1e-8f);
    const float a0 = (3.0f / 10.0f) * (1.0f + r0 * r0), a1 = (3.0f / 5.0f) * (1.0f + r1 * r1), a2 = (1.0f / 10.0f) * (1.0f + r2 * r2);
    const float sum = a0 + a1 + a2;
    const double w0 = (double)(a0 / sum), w1 = (double)(a1 / sum), w2 = (double)(a2 / sum);
    return w0 * p0 + w1 * p1 + w2 * p2;
#else
    const double b0 = c * (10.0 * c - 31.0 * d + 11.0 * e) + d * (25.0 * d - 19.0 * e) + e * (4.0 * e);
    const double b1 = b * (4.0 * b - 13.0 * c + 5.0 * d) + c * (13.0 * c - 13.0 * d) + d * (4.0 * d);
    const double b2 = a * (4.0 * a - 19.0 * b + 11.0 * c) + b * (25.0 * b - 31.0 * c) + c * (10.0 * c);
    const double tau = fabs(b0 - b2);
    const double r0 = bz_newton_div(tau, b0 + BZ_WENO_EPS), r1 = bz_newton_div(tau, b1 + BZ_WENO_EPS), r2 = bz_newton_div(tau, b2 + BZ_WENO_EPS);
    const double a0 = (3.0 / 10.0) * (1.0 + r0 * r0), a1 = (3.0 / 5.0) * (1.0 + r1 * r1), a2 = (1.0 / 10.0) * (1.0 + r2 * r2);
    return (a0 * p0 + a1 * p1 + a2 * p2) / (a0 + a1 + a2);
#endif
}

__device__ __forceinline__ double bz_weno5(double a, double b, double c, double d, double e)
{
#if BZ_WENO_FT2
    return bz_weno5_ft2(a, b, c, d, e);
#endif
#ifdef BZ_WENO_STUB      // timing experiments only: keeps every input live, no WENO arithmetic
    return 0.2 * (a + b + c + d + e);
#elif BZ_WENO_ONE_DIVISION == 2
    return bz_weno5_diff(a, b, c, d, e);
#elif BZ_WENO_ONE_DIVISION
    return bz_weno5_fast(a, b, c, d, e);
#else
    return bz_weno5_ref(a, b, c, d, e);
#endif
}

// 1 / x for a positive normal x: v_rcp seed (~2^-23 relative in Float64 as the ISA documents it, 1 ulp in Float32) + NR Newton
// steps in Float64 (1: ~1e-14, 2: full precision); Float32 needs none
template <int NR>
__device__ __forceinline__ double bz_recip(double x)
{
    double r = __builtin_amdgcn_rcp(x);
    if (sizeof(double) == 8) {
#pragma unroll
        for (int it = 0; it < NR; ++it) r = fma(fma(-x, r, 1.0), r, r);
    }
    return r;
}

// cells (a,b,c), upwind cell b, value at the face between b and c
__device__ __forceinline__ double bz_weno3(double a, double b, double c)
{
    double p0 = 0.5 * b + 0.5 * c;
    double p1 = -0.5 * a + 1.5 * b;
#if BZ_WENO_FT2 == 2
    {
        const float af = (float)a, bf = (float)b, cf = (float)c;
        const float b0 = (cf - bf) * (cf - bf), b1 = (bf - af) * (bf - af);
        const float tau = fabsf(b0 - b1);
        const float r0 = tau / (b0 + 1e-8f), r1 = tau / (b1 + 1e-8f);
        const float a0 = (2.0f / 3.0f) * (1.0f + r0 * r0), a1 = (1.0f / 3.0f) * (1.0f + r1 * r1);
        const float sum = a0 + a1;
        return (double)(a0 / sum) * p0 + (double)(a1 / sum) * p1;
    }
#endif
    double b0 = (c - b) * (c - b);
    double b1 = (b - a) * (b - a);
    double tau = fabs(b0 - b1);
#if BZ_WENO_FT2 == 1
    double r0 = bz_newton_div(tau, b0 + BZ_WENO_EPS);
    double r1 = bz_newton_div(tau, b1 + BZ_WENO_EPS);
#else
    double r0 = tau / (b0 + BZ_WENO_EPS);
    double r1 = tau / (b1 + BZ_WENO_EPS);
#endif
    double a0 = (2.0 / 3.0) * (1.0 + r0 * r0);
    double a1 = (1.0 / 3.0) * (1.0 + r1 * r1);
    return (a0 * p0 + a1 * p1) / (a0 + a1);
}

// Lane-mask select m ? a : b as v_cndmask_b32_e64 with the mask in an SGPR pair.  Why not the ternary operator: hipcc emits a
// share of its selects as the VOP2 form that reads VCC, and on MI355X that encoding issues at ~9.5 ns of SIMD time per wave64
// instruction against ~1.9 ns for the VOP3 form with an explicit SGPR-pair mask (tools/valu_rates.hip, profiles/r03_valu_rates.txt);
// an upwind WENO-5 selects five doubles = ten of them.
template <class R>
__device__ __forceinline__ R bz_sel(unsigned long long m, R a, R b)
{
    if constexpr (sizeof(R) == 8) {
        const int alo = __double2loint(a), ahi = __double2hiint(a), blo = __double2loint(b), bhi = __double2hiint(b);
        int lo, hi;
        asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(lo) : "v"(blo), "v"(alo), "s"(m));
        asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(hi) : "v"(bhi), "v"(ahi), "s"(m));
        return __hiloint2double(hi, lo);
    } else {
        const int ai = __float_as_int(a), bi = __float_as_int(b);
        int r;
        asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(r) : "v"(bi), "v"(ai), "s"(m));
        return __int_as_float(r);
    }
}
__device__ __forceinline__ unsigned long long bz_lanes(bool p) { return __builtin_amdgcn_ballot_w64(p); }
// Wave-uniform upwind shortcut (bz_up5): on in the Float64 build.  Measured at 512^3: Float64 41.35 -> 40.95 ms/step; in the Float32
// build the three inlined copies of the reconstruction push the 80-VGPR kernels into spills (25.7 -> 29.3 ms/step), so it stays off.
#ifndef BZ_UPWIND_UNIFORM
#define BZ_UPWIND_UNIFORM (sizeof(double) == 8)
#endif

#ifdef BZ_CENTERED2
// libbreeze_hip_centered2.so: advection = Centered(order = 2), the AtmosphereModel constructor's default
// (/root/reference/src/AtmosphereModels/atmosphere_model.jl advection keyword; Oceananigans Centered: the advected quantity and
// the advecting mass flux are both 2-point symmetric interpolations, no upwinding, no order reduction at walls).  The kernels
// are unchanged; every reconstruction collapses to the mean of the two cells adjacent to the target.
__device__ __forceinline__ double bz_up5(double, double, double m1, double p0, double, double, bool) { return 0.5 * (m1 + p0); }
__device__ __forceinline__ double bz_up3(double, double m1, double p0, double, bool) { return 0.5 * (m1 + p0); }
__device__ __forceinline__ double bz_upB(double, double, double m1, double p0, double, double, bool, int) { return 0.5 * (m1 + p0); }
#else
// Six values straddling the target (which lies between m1 and p0).  left = advecting flux > 0.
__device__ __forceinline__ double bz_up5(double m3, double m2, double m1, double p0, double p1,
                                         double p2, bool left)
{
    const unsigned long long m = bz_lanes(left);
    if constexpr (BZ_UPWIND_UNIFORM) {
    // every active lane of the wavefront upwinds to the same side (a wavefront is 64 consecutive x cells of one row and level: under a
    // mean wind, or in any smooth stretch of the flow, that is the common case): the stencil is known without a select — ten
    // v_cndmask per reconstruction otherwise.  The branch is scalar (ballot mask against the EXEC mask), the arithmetic identical.
    if (m == bz_lanes(true)) return bz_weno5(m3, m2, m1, p0, p1);
    if (m == 0ull) return bz_weno5(p2, p1, p0, m1, m2);
    }
    double a = bz_sel(m, m3, p2);
    double b = bz_sel(m, m2, p1);
    double c = bz_sel(m, m1, p0);
    double d = bz_sel(m, p0, m1);
    double e = bz_sel(m, p1, m2);
    return bz_weno5(a, b, c, d, e);
}
__device__ __forceinline__ double bz_up3(double m2, double m1, double p0, double p1, bool left)
{
    const unsigned long long m = bz_lanes(left);
    double a = bz_sel(m, m2, p1);
    double b = bz_sel(m, m1, p0);
    double c = bz_sel(m, p0, m1);
    return bz_weno3(a, b, c);
}
// buffer-aware (B wave-uniform: 3, 2 or 1)
__device__ __forceinline__ double bz_upB(double m3, double m2, double m1, double p0, double p1,
                                         double p2, bool left, int B)
{
    if (B == 3) return bz_up5(m3, m2, m1, p0, p1, p2, left);
    if (B == 2) return bz_up3(m2, m1, p0, p1, left);
    return bz_sel(bz_lanes(left), m1, p0);
}
#endif

// Largest buffer usable at index idx of a Bounded direction with N cells
// (face target: B <= idx <= N-B; centre target: B-1 <= idx <= N-B).
__device__ __forceinline__ int bz_buffer_face(int idx, int N)
{
    if (idx >= 3 && idx <= N - 3) return 3;
    if (idx >= 2 && idx <= N - 2) return 2;
    return 1;
}
__device__ __forceinline__ int bz_buffer_center(int idx, int N)
{
    if (idx >= 2 && idx <= N - 3) return 3;
    if (idx >= 1 && idx <= N - 2) return 2;
    return 1;
}

// Centered(order 4) of four values straddling the target (between qm1 and q0); order 2 fallback.
__device__ __forceinline__ double bz_symm4(double qm2, double qm1, double q0, double qp1)
{
#ifdef BZ_CENTERED2
    return 0.5 * (qm1 + q0);
#else
    return (7.0 / 12.0) * (qm1 + q0) - (1.0 / 12.0) * (qm2 + qp1);
#endif
}
__device__ __forceinline__ double bz_symm2(double qm1, double q0) { return 0.5 * (qm1 + q0); }
