// bz_tendency3_kernels.h — WENO-5 flux-form tendencies, third generation.
//
// Measurements that shaped it (MI355X, 512x512x256, tools/tendbench + rocprofv3 PMC):
//  * gen-1 (bz_tendency.hip: every thread evaluates both x and both y faces, 5 reconstructions per
//    cell and field) runs at ~80 % VALU utilisation: it is FP64-issue bound (one WENO-5 with the
//    single-division weights = ~85 FP64 instructions = ~390 cycles of a SIMD at 4 waves/SIMD).
//  * gen-2 (bz_tendency2: one flux per face through an LDS exchange and a barrier per level) cut the
//    reconstructions to 3.27 per cell but dropped to 43 % VALU utilisation (SQ_WAIT_ANY 59 %): the
//    lock-step load -> compute -> barrier cycle exposes memory latency that the free-running gen-1
//    waves hide.
// So this generation keeps gen-1's free-running waves (no LDS, no barriers) and removes redundancy
// with registers and cross-lane moves only:
//   x  each lane evaluates the flux of its own face and reads the neighbour's with a wave shuffle; the
//      one flux per wave-row outside the wave is batched over 64 levels (lane l <-> level kb + l) and
//      broadcast with a readlane: 1 + 1/64 reconstructions per cell;
//   y  each thread owns R consecutive rows and evaluates the R + 1 y-fluxes bounding them from one
//      (R + 6)-value column it loads once: 1 + 1/R reconstructions per cell;
//   z  upward march, 6-value register ring per row, lower-face flux carried: 1 per cell.
// R = 2: 3.52 reconstructions per cell instead of 5, and 40 % fewer load instructions per cell.
//
// Reference semantics: src/Advection.jl:20-35, src/AtmosphereModels/dynamics_kernel_functions.jl:54-159,
// src/AnelasticEquations/anelastic_buoyancy.jl:36-72 (paths relative to /root/reference).
#pragma once
#include "bz_internal.h"
#include "bz_weno.h"

enum { T3_SCALAR = 0, T3_U = 1, T3_V = 2, T3_W = 3 };

struct Tend3Fields {
    const double *ru, *rv, *rw;     // advecting momentum (momentum kernels)
    const double *u, *v, *w;        // advecting velocities (scalar kernel)
    const double *c;                // advected quantity: theta | q | u | v | w
    const double *T, *q;            // buoyancy inputs (W kernel)
    double *G;
};

// advecting flux in x at (i,j,k) [index n], for the x-flux location of KIND
template <int KIND>
__device__ __forceinline__ double adv_x(const DevGrid &g, const Tend3Fields &F, long long n, int k)
{
    const long long sy = g.Sx, sz = g.Sxy;
    if constexpr (KIND == T3_SCALAR) {
        return F.u[n];
    } else if constexpr (KIND == T3_U) {        // to cell centre i
        const double A = g.Ax[k];
        const double *m = F.ru;
        return bz_symm4(A * m[n - 1], A * m[n], A * m[n + 1], A * m[n + 2]);
    } else if constexpr (KIND == T3_V) {        // to (x-face i, y-face j)
        const double A = g.Ax[k];
        const double *m = F.ru;
        return bz_symm4(A * m[n - 2 * sy], A * m[n - sy], A * m[n], A * m[n + sy]);
    } else {                                    // to (x-face i, z-face k)
        const double *m = F.ru, *A = g.Ax;
        return (bz_buffer_face(k, g.Nz) == 3)
                   ? bz_symm4(A[k - 2] * m[n - 2 * sz], A[k - 1] * m[n - sz], A[k] * m[n], A[k + 1] * m[n + sz])
                   : bz_symm2(A[k - 1] * m[n - sz], A[k] * m[n]);
    }
}
template <int KIND>
__device__ __forceinline__ double adv_y(const DevGrid &g, const Tend3Fields &F, long long n, int k)
{
    const long long sy = g.Sx, sz = g.Sxy;
    if constexpr (KIND == T3_SCALAR) {
        return F.v[n];
    } else if constexpr (KIND == T3_U) {        // to (x-face i, y-face j)
        const double A = g.Ay[k];
        const double *m = F.rv;
        return bz_symm4(A * m[n - 2], A * m[n - 1], A * m[n], A * m[n + 1]);
    } else if constexpr (KIND == T3_V) {        // to cell centre j
        const double A = g.Ay[k];
        const double *m = F.rv;
        return bz_symm4(A * m[n - sy], A * m[n], A * m[n + sy], A * m[n + 2 * sy]);
    } else {                                    // to (y-face j, z-face k)
        const double *m = F.rv, *A = g.Ay;
        return (bz_buffer_face(k, g.Nz) == 3)
                   ? bz_symm4(A[k - 2] * m[n - 2 * sz], A[k - 1] * m[n - sz], A[k] * m[n], A[k + 1] * m[n + sz])
                   : bz_symm2(A[k - 1] * m[n - sz], A[k] * m[n]);
    }
}
// horizontal flux from the advecting flux `a` and the 6 advected values straddling the target
template <int KIND>
__device__ __forceinline__ double hflux(const DevGrid &g, double a, double area, int k, double m3, double m2,
                                        double m1, double p0, double p1, double p2)
{
    const double cR = bz_up5(m3, m2, m1, p0, p1, p2, a > 0.0);
    if constexpr (KIND == T3_SCALAR) return g.rho[k] * ((area * a) * cR);
    else return a * cR;
}
// full x flux at column i (cell or face index by KIND), row j, level k — used for the out-of-wave flux
template <int KIND>
__device__ __forceinline__ double flux_x_at(const DevGrid &g, const Tend3Fields &F, int i, int j, int k)
{
    const long long n = g.idx(i, j, k);
    const double a = adv_x<KIND>(g, F, n, k);
    const double *c = F.c;
    if constexpr (KIND == T3_U) return hflux<KIND>(g, a, 0.0, k, c[n - 2], c[n - 1], c[n], c[n + 1], c[n + 2], c[n + 3]);
    else return hflux<KIND>(g, a, g.Ax[k], k, c[n - 3], c[n - 2], c[n - 1], c[n], c[n + 1], c[n + 2]);
}
// vertical flux: SCALAR/U/V at z-face kt, W at cell centre kt; nt = idx(i,j,kt)
template <int KIND>
__device__ __forceinline__ double vflux(const DevGrid &g, const Tend3Fields &F, long long nt, int kt, double m3,
                                        double m2, double m1, double p0, double p1, double p2)
{
    const long long sy = g.Sx, sz = g.Sxy;
    const double Az = g.Az;
    if constexpr (KIND == T3_SCALAR) {
        const double wt = F.w[nt];
        const double cR = bz_upB(m3, m2, m1, p0, p1, p2, wt > 0.0, bz_buffer_face(kt, g.Nz));
        return g.rho_f[kt] * ((Az * wt) * cR);
    } else if constexpr (KIND == T3_U) {
        const double *m = F.rw;
        const double wt = bz_symm4(Az * m[nt - 2], Az * m[nt - 1], Az * m[nt], Az * m[nt + 1]);
        return wt * bz_upB(m3, m2, m1, p0, p1, p2, wt > 0.0, bz_buffer_face(kt, g.Nz));
    } else if constexpr (KIND == T3_V) {
        const double *m = F.rw;
        const double wt = bz_symm4(Az * m[nt - 2 * sy], Az * m[nt - sy], Az * m[nt], Az * m[nt + sy]);
        return wt * bz_upB(m3, m2, m1, p0, p1, p2, wt > 0.0, bz_buffer_face(kt, g.Nz));
    } else {
        const double *m = F.rw;
        const int B = bz_buffer_center(kt, g.Nz);
        const double wt = (B == 3) ? bz_symm4(Az * m[nt - sz], Az * m[nt], Az * m[nt + sz], Az * m[nt + 2 * sz])
                                   : bz_symm2(Az * m[nt], Az * m[nt + sz]);
        return wt * bz_upB(m3, m2, m1, p0, p1, p2, wt > 0.0, B);
    }
}
__device__ __forceinline__ double buoyancy3(const DevGrid &g, double T, double q, int k)
{   // anelastic_buoyancy.jl:36-72, dry reference state (R_m,r = Rd)
    const double Rm = (1.0 - q) * g.Rd + q * g.Rv;
    const double rhop = g.rho[k] * (g.Rd * g.T_r[k] / (Rm * T) - 1.0);
    return -g.g * rhop;
}

// block = 64 lanes (x) x TYW waves; each wave owns R consecutive rows of 64 columns.
template <int KIND, int R, int TYW>
__global__ __launch_bounds__(64 * TYW) void k_tend3(DevGrid g, Tend3Fields F, int kchunk)
{
    constexpr bool XC = (KIND == T3_U);     // x flux at cell centres: G(i) = F(i) - F(i-1)
    constexpr bool YC = (KIND == T3_V);     // y flux at cell centres: G(j) = F(j) - F(j-1)
    const int lane = threadIdx.x;
    const int i0 = blockIdx.x * 64;
    const int i = i0 + lane;
    const int ja = (blockIdx.y * TYW + threadIdx.y) * R;      // first row of this thread
    if (ja >= g.Ny) return;                                   // whole wave out of range (wave-uniform)
    const int ic = min(i, g.Nx - 1);
    const int nact = min(64, g.Nx - i0);
    const int ie = XC ? (i0 - 1) : (i0 + nact);               // x index of the out-of-wave flux
    const int le = XC ? 0 : (nact - 1);                       // lane that consumes it
    const int kbeg = (KIND == T3_W ? 1 : 0) + blockIdx.z * kchunk;
    const int kend = min(kbeg + kchunk, g.Nz);
    if (kbeg >= kend) return;
    const long long sy = g.Sx, sz = g.Sxy;
    const double *c = F.c;

    int jr[R];              // clamped row index per owned row (loads stay in bounds for a ragged last tile)
    bool st[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        jr[r] = min(ja + r, g.Ny - 1);
        st[r] = (i < g.Nx) && (ja + r < g.Ny);
    }

    // vertical rings and carried lower fluxes
    double ring[R][6], fz_lo[R], b_lo[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const long long n = g.idx(ic, jr[r], kbeg);
#pragma unroll
        for (int s = 0; s < 6; ++s) ring[r][s] = c[n + (s - 3) * sz];
        if constexpr (KIND == T3_W) {
            fz_lo[r] = vflux<KIND>(g, F, n - sz, kbeg - 1, ring[r][0], ring[r][1], ring[r][2], ring[r][3], ring[r][4], ring[r][5]);
            b_lo[r] = bz_buoyancy(g, F.T, F.q, n - sz, kbeg - 1);
        } else {
            fz_lo[r] = vflux<KIND>(g, F, n, kbeg, ring[r][0], ring[r][1], ring[r][2], ring[r][3], ring[r][4], ring[r][5]);
            b_lo[r] = 0.0;
        }
    }
    double edge[R];
#pragma unroll
    for (int r = 0; r < R; ++r) edge[r] = 0.0;

    for (int k = kbeg; k < kend; ++k) {
        // ---- out-of-wave x fluxes, batched over the next 64 levels (lane l <-> level k + l) ----
        if (((k - kbeg) & 63) == 0) {
            const int kk = min(k + lane, kend - 1);
#pragma unroll
            for (int r = 0; r < R; ++r) edge[r] = flux_x_at<KIND>(g, F, ie, jr[r], kk);
        }
        const int src = (k - kbeg) & 63;
        const double Ax = g.Ax[k], Ay = g.Ay[k];

        // ---- y: column of R + 6 advected values, R + 1 fluxes ----
        // window row index m <-> grid row ja - 3 + m; face-type flux f sits at y-face ja + f and uses
        // cy[f..f+5]; centre-type (V) flux f sits at centre ja - 1 + f and uses the same window.
        double cy[R + 6];
        const long long ncol = g.idx(ic, ja, k);          // rows ja-3 .. ja+R+2 are inside the halo (Hy >= 3)
#pragma unroll
        for (int m = 0; m < R + 6; ++m) cy[m] = c[ncol + (long long)(min(ja - 3 + m, g.Ny + 2) - ja) * sy];
        double fy[R + 1];
#pragma unroll
        for (int f = 0; f <= R; ++f) {
            const int jf = YC ? min(ja - 1 + f, g.Ny - 1) : min(ja + f, g.Ny);
            const long long nf = g.idx(ic, jf, k);
            const double a = adv_y<KIND>(g, F, nf, k);
            fy[f] = hflux<KIND>(g, a, Ay, k, cy[f], cy[f + 1], cy[f + 2], cy[f + 3], cy[f + 4], cy[f + 5]);
        }

#pragma unroll
        for (int r = 0; r < R; ++r) {
            const long long n = g.idx(ic, jr[r], k);
            // ---- x: own flux, neighbour by shuffle, out-of-wave flux by readlane ----
            const double a = adv_x<KIND>(g, F, n, k);
            const double c0 = cy[3 + r];
            double fx;
            if constexpr (XC) fx = hflux<KIND>(g, a, Ax, k, c[n - 2], c[n - 1], c0, c[n + 1], c[n + 2], c[n + 3]);
            else fx = hflux<KIND>(g, a, Ax, k, c[n - 3], c[n - 2], c[n - 1], c0, c[n + 1], c[n + 2]);
            double fx_nb = XC ? __shfl_up(fx, 1) : __shfl_down(fx, 1);
            const double e = __shfl(edge[r], src);
            if (lane == le) fx_nb = e;
            const double dx = XC ? (fx - fx_nb) : (fx_nb - fx);
            // ---- z: ring + carried flux ----
            const double t = c[n + 3 * sz];
            double fz_hi, b_hi = 0.0;
            if constexpr (KIND == T3_W) {
                fz_hi = vflux<KIND>(g, F, n, k, ring[r][1], ring[r][2], ring[r][3], ring[r][4], ring[r][5], t);
                b_hi = bz_buoyancy(g, F.T, F.q, n, k);
            } else {
                fz_hi = vflux<KIND>(g, F, n + sz, k + 1, ring[r][1], ring[r][2], ring[r][3], ring[r][4], ring[r][5], t);
            }
#pragma unroll
            for (int s = 0; s < 5; ++s) ring[r][s] = ring[r][s + 1];
            ring[r][5] = t;
            const double dy = fy[r + 1] - fy[r];
            const double dz = fz_hi - fz_lo[r];
            double G;
            if constexpr (KIND == T3_W) G = -(g.Vinv_f[k] * (dx + dy + dz)) + 0.5 * (b_lo[r] + b_hi);
            else G = -(g.Vinv_c[k] * (dx + dy + dz));
            if (st[r]) F.G[n] = G;
            fz_lo[r] = fz_hi;
            b_lo[r] = b_hi;
        }
    }
}

// ---------------------------------------------------------------------------------------------------
// Fused potential-temperature + moisture tendency: both scalars are advected by the same velocities, so
// one kernel shares the u, v, w loads, the upwind decisions and the flux prefactors (8 words of HBM
// traffic per cell for two fields instead of 2 x 6).  x fluxes are shared across lanes as in k_tend3
// (shuffle + batched out-of-wave flux); the two y faces of a cell are evaluated in place (R = 1 was
// the fastest scalar variant in tools/tendbench).
// ---------------------------------------------------------------------------------------------------
template <int TYW>
__global__ __launch_bounds__(64 * TYW) void k_scalar_pair(DevGrid g, const double *__restrict__ u,
                                                         const double *__restrict__ v,
                                                         const double *__restrict__ w,
                                                         const double *__restrict__ ca,
                                                         const double *__restrict__ cb, double *__restrict__ Ga,
                                                         double *__restrict__ Gb, int kchunk)
{
    const int lane = threadIdx.x;
    const int i0 = blockIdx.x * 64, i = i0 + lane;
    const int j = blockIdx.y * TYW + threadIdx.y;
    if (j >= g.Ny) return;                                    // wave-uniform
    const int ic = min(i, g.Nx - 1);
    const int nact = min(64, g.Nx - i0);
    const int ie = i0 + nact, le = nact - 1;
    const int kbeg = blockIdx.z * kchunk, kend = min(kbeg + kchunk, g.Nz);
    if (kbeg >= kend) return;
    const long long sy = g.Sx, sz = g.Sxy;
    const bool store = i < g.Nx;
    long long n = g.idx(ic, j, kbeg);

    double a[6], b[6];
#pragma unroll
    for (int s = 0; s < 6; ++s) { a[s] = ca[n + (s - 3) * sz]; b[s] = cb[n + (s - 3) * sz]; }
    double fza, fzb;
    {
        const double wt = w[n];
        const bool left = wt > 0.0;
        const int B = bz_buffer_face(kbeg, g.Nz);
        const double cf = g.Az * wt, rf = g.rho_f[kbeg];
        fza = rf * (cf * bz_upB(a[0], a[1], a[2], a[3], a[4], a[5], left, B));
        fzb = rf * (cf * bz_upB(b[0], b[1], b[2], b[3], b[4], b[5], left, B));
    }
    double ea = 0.0, eb = 0.0;

    for (int k = kbeg; k < kend; ++k, n += sz) {
        if (((k - kbeg) & 63) == 0) {       // out-of-wave x flux for the next 64 levels (lane l <-> level k + l)
            const int kk = min(k + lane, kend - 1);
            const long long ne = g.idx(ie, j, kk);
            const double ue = u[ne];
            const bool le_ = ue > 0.0;
            const double cf = g.Ax[kk] * ue, rho = g.rho[kk];
            ea = rho * (cf * bz_up5(ca[ne - 3], ca[ne - 2], ca[ne - 1], ca[ne], ca[ne + 1], ca[ne + 2], le_));
            eb = rho * (cf * bz_up5(cb[ne - 3], cb[ne - 2], cb[ne - 1], cb[ne], cb[ne + 1], cb[ne + 2], le_));
        }
        const int src = (k - kbeg) & 63;
        const double rho = g.rho[k];
        // ---- z ----
        const double ta = ca[n + 3 * sz], tb = cb[n + 3 * sz];
        double fza_hi, fzb_hi;
        {
            const double wt = w[n + sz];
            const bool left = wt > 0.0;
            const int B = bz_buffer_face(k + 1, g.Nz);
            const double cf = g.Az * wt, rf = g.rho_f[k + 1];
            fza_hi = rf * (cf * bz_upB(a[1], a[2], a[3], a[4], a[5], ta, left, B));
            fzb_hi = rf * (cf * bz_upB(b[1], b[2], b[3], b[4], b[5], tb, left, B));
        }
        // ---- x ----
        double dxa, dxb;
        {
            const double ut = u[n];
            const bool left = ut > 0.0;
            const double cf = g.Ax[k] * ut;
            const double fa = rho * (cf * bz_up5(ca[n - 3], ca[n - 2], ca[n - 1], a[3], ca[n + 1], ca[n + 2], left));
            const double fb = rho * (cf * bz_up5(cb[n - 3], cb[n - 2], cb[n - 1], b[3], cb[n + 1], cb[n + 2], left));
            double na = __shfl_down(fa, 1), nb = __shfl_down(fb, 1);
            const double xa = __shfl(ea, src), xb = __shfl(eb, src);
            if (lane == le) { na = xa; nb = xb; }
            dxa = na - fa;
            dxb = nb - fb;
        }
        // ---- y ----
        double dya, dyb;
        {
            const double v0 = v[n], v1 = v[n + sy];
            const bool l0 = v0 > 0.0, l1 = v1 > 0.0;
            const double c0 = g.Ay[k] * v0, c1 = g.Ay[k] * v1;
            {
                const double m3 = ca[n - 3 * sy], m2 = ca[n - 2 * sy], m1 = ca[n - sy], p1 = ca[n + sy], p2 = ca[n + 2 * sy], p3 = ca[n + 3 * sy];
                const double lo = rho * (c0 * bz_up5(m3, m2, m1, a[3], p1, p2, l0));
                const double hi = rho * (c1 * bz_up5(m2, m1, a[3], p1, p2, p3, l1));
                dya = hi - lo;
            }
            {
                const double m3 = cb[n - 3 * sy], m2 = cb[n - 2 * sy], m1 = cb[n - sy], p1 = cb[n + sy], p2 = cb[n + 2 * sy], p3 = cb[n + 3 * sy];
                const double lo = rho * (c0 * bz_up5(m3, m2, m1, b[3], p1, p2, l0));
                const double hi = rho * (c1 * bz_up5(m2, m1, b[3], p1, p2, p3, l1));
                dyb = hi - lo;
            }
        }
        const double Vi = g.Vinv_c[k];
        if (store) {
            Ga[n] = -(Vi * (dxa + dya + (fza_hi - fza)));
            Gb[n] = -(Vi * (dxb + dyb + (fzb_hi - fzb)));
        }
        fza = fza_hi; fzb = fzb_hi;
#pragma unroll
        for (int s = 0; s < 5; ++s) { a[s] = a[s + 1]; b[s] = b[s + 1]; }
        a[5] = ta; b[5] = tb;
    }
}

// ---------------------------------------------------------------------------------------------------
// z-momentum tendency with every vertical stencil in registers.
// rocprofv3 PMC on the gen-1 kernel: 21 GB fetched per launch at 512^3 for 6.4 GB of input arrays — the
// centred-in-z interpolation of the advecting fluxes (rho_u, rho_v at levels k-2..k+1, rho_w at k-1..k+2)
// re-read four planes of three arrays at every level.  Here each of those columns is a register ring fed by one
// new load per level; the x flux of the neighbouring column comes from a wave shuffle (+ batched out-of-wave
// flux, as in k_tend3), the two y faces of a cell use two rho_v rings.
// ---------------------------------------------------------------------------------------------------
template <int TYW>
__global__ __launch_bounds__(64 * TYW) void k_w_tend_ring(DevGrid g, Tend3Fields F, int kchunk, RKEpilogue E)
{
    const int lane = threadIdx.x;
    const int i0 = blockIdx.x * 64, i = i0 + lane;
    const int j = blockIdx.y * TYW + threadIdx.y;
    if (j >= g.Ny) return;                                    // wave-uniform
    const int ic = min(i, g.Nx - 1);
    const int nact = min(64, g.Nx - i0);
    const int ie = i0 + nact, le = nact - 1;
    const int kbeg = 1 + blockIdx.z * kchunk, kend = min(kbeg + kchunk, g.Nz);   // faces kbeg .. kend-1
    if (kbeg >= kend) return;
    const long long sy = g.Sx, sz = g.Sxy;
    const bool store = i < g.Nx;
    const double *w = F.c, *ru = F.ru, *rv = F.rv, *rw = F.rw;
    const double Az = g.Az;
    long long n = g.idx(ic, j, kbeg);

    double wr[6];            // w at faces k-3 .. k+2 (straddles centre k-1)
    double qu[4];            // Ax*rho_u at levels k-2 .. k+1, own column
    double qv0[4], qv1[4];   // Ay*rho_v at levels k-2 .. k+1, rows j and j+1
    double qw[4];            // Az*rho_w at faces k-2 .. k+1 (centre k-1 uses all four)
#pragma unroll
    for (int s = 0; s < 6; ++s) wr[s] = w[n + (s - 3) * sz];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        const int kk = kbeg - 2 + s;
        qu[s] = g.Ax[kk] * ru[n + (s - 2) * sz];
        qv0[s] = g.Ay[kk] * rv[n + (s - 2) * sz];
        qv1[s] = g.Ay[kk] * rv[n + sy + (s - 2) * sz];
        qw[s] = Az * rw[n + (s - 2) * sz];
    }
    // carried: flux through centre kbeg-1 and buoyancy of cell kbeg-1
    double fz_lo, b_lo;
    {
        const int B = bz_buffer_center(kbeg - 1, g.Nz);
        const double wt = (B == 3) ? bz_symm4(qw[0], qw[1], qw[2], qw[3]) : bz_symm2(qw[1], qw[2]);
        fz_lo = wt * bz_upB(wr[0], wr[1], wr[2], wr[3], wr[4], wr[5], wt > 0.0, B);
        b_lo = bz_buoyancy(g, F.T, F.q, n - sz, kbeg - 1);
    }
    double edge = 0.0;

    for (int k = kbeg; k < kend; ++k, n += sz) {
        if (((k - kbeg) & 63) == 0) {
            const int kk = min(k + lane, kend - 1);
            edge = flux_x_at<T3_W>(g, F, ie, j, kk);
        }
        const int src = (k - kbeg) & 63;
        const int Bf = bz_buffer_face(k, g.Nz);
        // ---- z: centre k needs Az*rho_w at faces k-1..k+2 and w at faces k-2..k+3 ----
        const double wnew = w[n + 3 * sz];
        const double qwnew = Az * rw[n + 2 * sz];
        double fz_hi;
        {
            const int B = bz_buffer_center(k, g.Nz);
            const double wt = (B == 3) ? bz_symm4(qw[1], qw[2], qw[3], qwnew) : bz_symm2(qw[2], qw[3]);
            fz_hi = wt * bz_upB(wr[1], wr[2], wr[3], wr[4], wr[5], wnew, wt > 0.0, B);
        }
        const double w0 = wr[3];        // w at (i, j, k)
        // ---- x: advecting flux of the own x-face from the rho_u ring; neighbour by shuffle ----
        double dx;
        {
            const double ut = (Bf == 3) ? bz_symm4(qu[0], qu[1], qu[2], qu[3]) : bz_symm2(qu[1], qu[2]);
            const double fx = ut * bz_up5(w[n - 3], w[n - 2], w[n - 1], w0, w[n + 1], w[n + 2], ut > 0.0);
            double nb = __shfl_down(fx, 1);
            const double e = __shfl(edge, src);
            if (lane == le) nb = e;
            dx = nb - fx;
        }
        // ---- y: both faces from the two rho_v rings ----
        double dy;
        {
            const double v0 = (Bf == 3) ? bz_symm4(qv0[0], qv0[1], qv0[2], qv0[3]) : bz_symm2(qv0[1], qv0[2]);
            const double v1 = (Bf == 3) ? bz_symm4(qv1[0], qv1[1], qv1[2], qv1[3]) : bz_symm2(qv1[1], qv1[2]);
            const double m3 = w[n - 3 * sy], m2 = w[n - 2 * sy], m1 = w[n - sy], p1 = w[n + sy], p2 = w[n + 2 * sy], p3 = w[n + 3 * sy];
            const double lo = v0 * bz_up5(m3, m2, m1, w0, p1, p2, v0 > 0.0);
            const double hi = v1 * bz_up5(m2, m1, w0, p1, p2, p3, v1 > 0.0);
            dy = hi - lo;
        }
        const double b_hi = bz_buoyancy(g, F.T, F.q, n, k);
        if (store)
            F.G[n] = bz_rk_apply(E.mode, E.dt, E.alpha, E.oma, E.u0, E.u0_out,
                                 -(g.Vinv_f[k] * (dx + dy + (fz_hi - fz_lo))) + 0.5 * (b_lo + b_hi), rw[n], n);
        fz_lo = fz_hi;
        b_lo = b_hi;
        // ---- advance the rings to level k+1 ----
        const double Axn = g.Ax[k + 2], Ayn = g.Ay[k + 2];
        const double qun = Axn * ru[n + 2 * sz], qv0n = Ayn * rv[n + 2 * sz], qv1n = Ayn * rv[n + sy + 2 * sz];
#pragma unroll
        for (int s = 0; s < 3; ++s) { qu[s] = qu[s + 1]; qv0[s] = qv0[s + 1]; qv1[s] = qv1[s + 1]; qw[s] = qw[s + 1]; }
        qu[3] = qun; qv0[3] = qv0n; qv1[3] = qv1n; qw[3] = qwnew;
#pragma unroll
        for (int s = 0; s < 5; ++s) wr[s] = wr[s + 1];
        wr[5] = wnew;
    }
}
