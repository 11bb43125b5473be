// bz_poisson.hip — anelastic pressure Poisson solve.
//   compute_anelastic_source_term!   /root/reference/src/AnelasticEquations/anelastic_pressure_solver.jl:90-105
//   tridiagonal coefficients         :32-78
//   solve!(phi, FourierTridiagonalPoissonSolver)  Oceananigans.Solvers (called at :86):
//     forward FFT in x,y (rocFFT, real-to-complex half spectrum) -> Thomas solve along z per
//     horizontal wavenumber -> inverse FFT -> zero mean -> real copy.
// The Thomas factors depend only on the grid and the reference density, so 1/beta_k and
// t_k = c_{k-1}/beta_{k-1} are tabulated once per (kx,ky,k) at bz_create.
#include <cmath>
#include <cstdlib>

#include "bz_internal.h"
#include "bz_weno.h"      // bz_recip

#define TX 64
#define TY 4

__global__ __launch_bounds__(TX *TY) void k_poisson_source(DevGrid g, double *__restrict__ rhs,
                                                          const double *__restrict__ ru,
                                                          const double *__restrict__ rv,
                                                          const double *__restrict__ rw, double dt)
{
    int i = blockIdx.x * TX + threadIdx.x, j = blockIdx.y * TY + threadIdx.y, k = blockIdx.z;
    if (i >= g.Nx || j >= g.Ny) return;
    long long n = g.idx(i, j, k);
    double Ax = g.Ax[k], Ay = g.Ay[k], Az = g.Az;
    double a = Ax * ru[n + 1] - Ax * ru[n];
    double b = g.flat_y ? 0.0 : Ay * rv[n + g.Sx] - Ay * rv[n];
    double c = Az * rw[n + g.Sxy] - Az * rw[n];
    double div = g.Vinv_c[k] * (a + b + c);
    rhs[(long long)i + (long long)g.Nx * ((long long)j + (long long)g.Ny * k)] = g.dzc[k] * div / dt;
}

struct TriCols {
    const double *lower;   // [Nz-1]  rho_f[k+1]/dzf[k+1]
    const double *diag0;   // [Nz]
    const double *mass;    // [Nz]    rho[k]*dzc[k]
    const double *lam_x;   // [NXH]
    const double *lam_y;   // [Ny]
};

// One thread per horizontal wavenumber: tabulate 1/beta_k and t_k of the Thomas forward sweep.
// NXH = number of kx columns held here (the whole half spectrum, or this rank's block [kx0, kx0+NXH) of the
// zero-padded half spectrum; columns with kx0+kx >= nxh_real are padding and get zero factors).
// ky_fastest: column index c = ky + Ny*kx (slab mode: the y transform runs along the contiguous dimension), else
// c = kx + NXH*ky.
__global__ __launch_bounds__(256) void k_tridiag_setup(int NXH, int Ny, int Nz, int kx0, int nxh_real, int ky_fastest, TriCols C,
                                                       double *__restrict__ ibeta,
                                                       double *__restrict__ tfac)
{
    long long c = (long long)blockIdx.x * 256 + threadIdx.x;
    long long plane = (long long)NXH * Ny;
    if (c >= plane) return;
    int kx = ky_fastest ? (int)(c / Ny) : (int)(c % NXH), ky = ky_fastest ? (int)(c % Ny) : (int)(c / NXH);
    if (kx0 + kx >= nxh_real) {
        for (int k = 0; k < Nz; ++k) { ibeta[c + plane * k] = 0.0; tfac[c + plane * k] = 0.0; }
        return;
    }
    double lam = C.lam_x[kx0 + kx] + C.lam_y[ky];
    double beta = C.diag0[0] - C.mass[0] * lam;
    ibeta[c] = 1.0 / beta;
    tfac[c] = 0.0;
    const double tiny = 10.0 * 2.220446049250313e-16;
    for (int k = 1; k < Nz; ++k) {
        double a = C.lower[k - 1];
        double t = a / beta;
        beta = (C.diag0[k] - C.mass[k] * lam) - a * t;
        tfac[c + plane * k] = t;
        // Oceananigans elides the forward update when |beta| <= 10 eps (singular (0,0) mode):
        // ibeta = 0 reproduces "keep the stale value", with stale value 0.
        ibeta[c + plane * k] = (fabs(beta) > tiny) ? 1.0 / beta : 0.0;
    }
}

// Thomas solve along z, in place on the half-spectrum.  scale = 1/(Nx*Ny) folds in the inverse-FFT
// normalisation.  Column (0,0) additionally gets its z-mean removed, which is the global mean of
// phi (mean removal of Oceananigans' solve!).
__global__ void __launch_bounds__(64) k_tridiag_solve(int NXH, int Ny, int Nz, const double *__restrict__ lower,
                                                      const double *__restrict__ ibeta,
                                                      const double *__restrict__ tfac,
                                                      double2 *__restrict__ hat, double scale, int mean_column)
{
    long long c = (long long)blockIdx.x * 64 + threadIdx.x;
    long long plane = (long long)NXH * Ny;
    if (c >= plane) return;
    double2 prev = make_double2(0.0, 0.0);
    double a = 0.0;
#pragma unroll 8
    for (int k = 0; k < Nz; ++k) {
        long long n = c + plane * k;
        double2 f = hat[n];
        double ib = ibeta[n];
        double2 v;
        v.x = (f.x * scale - a * prev.x) * ib;
        v.y = (f.y * scale - a * prev.y) * ib;
        hat[n] = v;
        prev = v;
        a = (k < Nz - 1) ? lower[k] : 0.0;
    }
    double2 next = prev;
    double sum = prev.x;
#pragma unroll 8
    for (int k = Nz - 2; k >= 0; --k) {
        long long n = c + plane * k;
        double2 v = hat[n];
        double t = tfac[n + plane];
        v.x -= t * next.x;
        v.y -= t * next.y;
        hat[n] = v;
        next = v;
        sum += v.x;
    }
    if (c == 0 && mean_column) {      // the (kx,ky) = (0,0) column lives here
        double mean = sum / Nz;
        for (int k = 0; k < Nz; ++k) hat[plane * k].x -= mean;
    }
}


// ---------------------------------------------------------------------------------------------------------------------------------
// Cooperative single-pass solve (round 2).  The Thomas kernel above reads and writes the spectrum twice (forward + backward sweep)
// and reads two factor tables: 5 words moved per 2 algorithmic ones (PMC 5.37 GB per launch at 512^3) on a 512-deep dependent chain
// with one thread per column.  Here a block of 512 threads owns 8 adjacent columns (one 128-byte line per level) and 64 segments of
// <= 8 levels each; thread (column c, segment s) keeps its rows in registers, so the spectrum is read once and written once and no
// table is read (the factors are recomputed: one division per row).  Partition method (Wang 1981):
//   A  local forward elimination started at the segment's first row, with a fill-in column g for the unknown above the segment:
//        x_k + cp_k x_{k+1} + g_k x_prev = dp_k
//   B  local backward elimination towards the segment's last unknown x_l:   x_k + g_k x_prev + h_k x_l = dp_k   (k < l)
//   C  the 64 last unknowns X_s of a column satisfy a tridiagonal system
//        g_l(s) X_{s-1} + (1 - cp_l(s) g_f(s+1)) X_s - cp_l(s) h_f(s+1) X_{s+1} = dp_l(s) - cp_l(s) dp_f(s+1)
//      (f: first row of the next segment), solved through LDS by one thread per column;
//   D  x_k = dp_k - g_k X_{s-1} - h_k X_s.
// The singular (kx, ky) = (0, 0) column is made regular the way the sequential kernel's |beta| guard does it — its last row is
// replaced by x = 0 — and then loses its mean (Oceananigans' solve! subtracts the mean of phi).
// Segments per column are a template parameter (64 / 16 / 8 for 128..512 / 32..127 / 16..31 levels, tridiag_coop_segs); shorter or
// longer columns use the sequential kernel.
// ---------------------------------------------------------------------------------------------------------------------------------
#define TCO_COLS 8
#define TCO_M 8
// minimum waves per SIMD the register allocation must leave room for (second argument of __launch_bounds__): 2 = ONE 512-thread
// workgroup per CU.  That is what is shipped and what every quoted figure was measured with (0.46 ms per launch at 512^3 in Float64,
// 0.28 in Float32): the persistent grid below is sized for one resident workgroup per CU (`resident`), each walking ngroups / 256
// column groups with the next group's rows in flight.  4 (two workgroups per CU, <= 128 VGPRs) was measured again in round 4 (ADVICE
// r03): the kernel spills and takes 0.98 ms — the rows of a group held in registers are the design, not an accident of the allocator.
#ifndef TCO_MIN_WAVES
#define TCO_MIN_WAVES 2
#endif
// Workgroup barrier that orders LDS traffic only.  __syncthreads() is a workgroup-scope fence over every address space: hipcc then waits
// for the global stores of the previous column group before it lets the next group's arithmetic start (s_waitcnt vmcnt(10) in the loop
// of k_tridiag_coop), i.e. the store drain of every group is exposed.  Every barrier of that kernel protects LDS exchange buffers.
__device__ __forceinline__ void tco_lds_barrier()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

// M rows per segment held in registers; EXACT: Nz == TCO_SEGS * M and the plane is a whole number of column groups, so every thread
// holds exactly M rows of a real column (no row or column predicates: straight-line loads and stores, whose completion the in-order
// memory counter can be waited on precisely — behind a predicate the compiler waits for everything outstanding, which drained every
// store of a group before the next group's loads could be used)
template <int TCO_SEGS, int M, bool EXACT>
__global__ void __launch_bounds__(TCO_COLS *TCO_SEGS, TCO_MIN_WAVES) k_tridiag_coop(int NXH, int Ny, int Nz, int kx0, int nxh_real, int ky_fastest, TriCols C,
                                                                      double2 *__restrict__ hat, double scale, int mean_column, int ngroups,
                                                                      long long lstride, long long kxs, int grp0)
{
    // Where (column c, level k) lives: hat[col_addr(c) + lstride k].  Level-major spectrum (kxs = 0): col_addr(c) = c, lstride = the plane;
    // kx-major spectrum of the chunked pipeline (round 6; ky fastest): wavenumber kx owns Nz Ny contiguous elements, col_addr(c) =
    // (c / Ny) kxs + c % Ny, lstride = Ny.  The launch covers the column groups [grp0, ngroups).
    __shared__ double sR[TCO_SEGS][TCO_COLS][6];      // g_f, h_f, g_l, cp_l of a segment; then sub / diag / sup of the reduced system
    __shared__ double2 sD[TCO_SEGS][TCO_COLS][2];     // dp_f, dp_l; then [0] = X_s
    __shared__ double sSum[TCO_SEGS];
    // column coefficients of every level, staged once per workgroup: read through LDS they stay off the vector-memory counter, which
    // returns in order — a coefficient load issued after the next group's rows would make its wait a wait for those rows as well
    __shared__ double sLow[TCO_SEGS * M + 1], sDiag[TCO_SEGS * M], sMass[TCO_SEGS * M];      // sLow[k] = lower[k - 1], 0 at both ends
    const int t = threadIdx.x, cc = t & (TCO_COLS - 1), s = t >> 3;
    for (int k = t; k <= Nz; k += TCO_COLS * TCO_SEGS) {
        sLow[k] = (k > 0 && k < Nz) ? C.lower[k - 1] : 0.0;
        if (k < Nz) { sDiag[k] = C.diag0[k]; sMass[k] = C.mass[k]; }
    }
    tco_lds_barrier();
    const long long plane = (long long)NXH * Ny;
    auto col_addr = [&](long long c) -> long long {
        if (!kxs) return c;
        const unsigned cu = (unsigned)min(c, plane - 1), quo = cu / (unsigned)Ny;
        return (long long)quo * kxs + (long long)(cu - quo * (unsigned)Ny);
    };
    const int q = Nz / TCO_SEGS, r = Nz % TCO_SEGS;
    const int L = EXACT ? M : q + (s < r ? 1 : 0), k0 = EXACT ? s * M : s * q + min(s, r);
    // A workgroup walks column groups grp = blockIdx.x, + gridDim.x, ... and requests the rows of its NEXT group before it starts on the
    // current one: with one group per workgroup (round 2) the two workgroups of a CU drifted into the same phase — both waiting for
    // their loads, then both in the dependent elimination chains — and the kernel took load time + arithmetic time (0.57 ms at 512^3
    // for 2.15 GB).  The loads of the next group now fly under the ~550 instructions of the current one.
    double2 dpn[M];
    double lxn, lyn;                                                     // lam_x, lam_y of the next group's column, fetched with its rows (unconditional
    auto lam_of = [&](long long c, double &lx, double &ly) {             // clamped loads, added where they are used: nothing here waits)
        const unsigned cu = (unsigned)min(c, plane - 1), den = (unsigned)(ky_fastest ? Ny : NXH), quo = cu / den, rem = cu - quo * den;      // plane < 2^31
        const int kx = ky_fastest ? (int)quo : (int)rem, ky = ky_fastest ? (int)rem : (int)quo;
        lx = C.lam_x[min(kx0 + kx, nxh_real - 1)];
        ly = C.lam_y[ky];
    };
    {
        const long long c = (long long)(grp0 + (int)blockIdx.x) * TCO_COLS + cc;
        const long long ca = col_addr(c);
        lam_of(c, lxn, lyn);
#pragma unroll
        for (int j = 0; j < M; ++j) {
            if constexpr (EXACT) dpn[j] = hat[ca + lstride * (k0 + j)];
            else if (j < L) dpn[j] = (c < plane) ? hat[ca + lstride * (k0 + j)] : make_double2(0.0, 0.0);
        }
    }
    auto trip = [&](const int grp) {
    const long long c = (long long)grp * TCO_COLS + cc;
    const bool live = EXACT || c < plane;                                // ragged last block (2-D grids: plane = Nx/2 + 1 columns)
    const unsigned den = (unsigned)(ky_fastest ? Ny : NXH), quo = (unsigned)c / den;
    const int kx = ky_fastest ? (int)quo : (int)((unsigned)c - quo * den);
    const bool padding = (kx0 + kx >= nxh_real) || !live;
    const bool pinned = mean_column && c == 0;                           // the (0, 0) column of the whole spectrum lives here
    const double lam = padding ? 0.0 : lxn + lyn;
    double2 dp[M];
    double g[M], h[M];
    double2 *col = hat + col_addr(c);
#pragma unroll
    for (int j = 0; j < M; ++j) dp[j] = dpn[j];
    if constexpr (EXACT) {                                               // the last trip re-reads its own group (in bounds, unused)
        const long long cn = (grp + (int)gridDim.x < ngroups) ? c + (long long)gridDim.x * TCO_COLS : c;
        const long long cna = col_addr(cn);
        lam_of(cn, lxn, lyn);
#pragma unroll
        for (int j = 0; j < M; ++j) dpn[j] = hat[cna + lstride * (k0 + j)];
    } else if (grp + (int)gridDim.x < ngroups) {                         // block-uniform
        const long long cn = c + (long long)gridDim.x * TCO_COLS;
        const long long cna = col_addr(cn);
        lam_of(cn, lxn, lyn);
#pragma unroll
        for (int j = 0; j < M; ++j)
            if (j < L) dpn[j] = (cn < plane) ? hat[cna + lstride * (k0 + j)] : make_double2(0.0, 0.0);
    }
    // ---- A: local forward elimination ----
    double cp_prev = 0.0, g_prev = 0.0;
    double2 d_prev = make_double2(0.0, 0.0);
#pragma unroll
    for (int j = 0; j < M; ++j) {
        if (j < L) {
            const int k = k0 + j;
            double a = sLow[k], cu = sLow[k + 1];
            double b = sDiag[k] - sMass[k] * lam;
            double2 d = make_double2(dp[j].x * scale, dp[j].y * scale);
            if (pinned && k == Nz - 1) { a = 0.0; b = 1.0; cu = 0.0; d = make_double2(0.0, 0.0); }
            if (padding) { a = 0.0; b = 1.0; cu = 0.0; d = make_double2(0.0, 0.0); }
            double inv, gk;
            if (j == 0) { inv = bz_recip<2>(b); gk = a * inv; d.x *= inv; d.y *= inv; }
            else {
                inv = bz_recip<2>(b - a * cp_prev);      // pivots of a diagonally dominant matrix: normal, never zero (reciprocal by v_rcp + Newton steps)
                gk = -(a * g_prev) * inv;
                d.x = (d.x - a * d_prev.x) * inv;
                d.y = (d.y - a * d_prev.y) * inv;
            }
            cp_prev = cu * inv;
            g_prev = gk;
            d_prev = d;
            dp[j] = d; g[j] = gk; h[j] = cp_prev;        // h holds cp until stage B
        }
    }
    // ---- B: local backward elimination towards the last unknown (row L-1 keeps its coupling cp to the next segment) ----
    const double cp_l = cp_prev;                       // cp of the last row
#pragma unroll
    for (int j = M - 3; j >= 0; --j) {
        if (j <= L - 3) {
            const double cpk = h[j];
            dp[j].x -= cpk * dp[j + 1].x;
            dp[j].y -= cpk * dp[j + 1].y;
            g[j] -= cpk * g[j + 1];
            h[j] = -cpk * h[j + 1];
        }
    }
    // row L-2 already reads x_k + cp_k x_l + g_k x_prev = dp_k: h[L-2] = cp[L-2] as stored
    // ---- C: reduced system of the segment-last unknowns ----
    {
        double gl = 0.0;
        double2 dl = make_double2(0.0, 0.0);
#pragma unroll
        for (int j = 0; j < M; ++j)
            if (j == L - 1) { gl = g[j]; dl = dp[j]; }
        sR[s][cc][0] = g[0]; sR[s][cc][1] = h[0]; sR[s][cc][2] = gl; sR[s][cc][3] = cp_l;
        sD[s][cc][0] = dp[0]; sD[s][cc][1] = dl;
    }
    tco_lds_barrier();
    // Parallel cyclic reduction over the TCO_SEGS reduced unknowns of a column, one thread per unknown (the threads of a column are
    // t = s * 8 + cc).  Rows are kept normalised, a X_{m-h} + X_m + c X_{m+h} = d; one step with stride h eliminates both neighbours:
    //   r = 1 / (1 - a c_{m-h} - c a_{m+h}),  a' = -a a_{m-h} r,  c' = -c c_{m+h} r,  d' = (d - a d_{m-h} - c d_{m+h}) r
    // (rows beyond the ends are identity rows: a = c = d = 0), and after log2(TCO_SEGS) steps X_m = d.  The system inherits the
    // diagonal dominance of the column's own tridiagonal matrix, so no pivoting is needed.  The serial Thomas sweep this replaces kept
    // 8 of the block's 512 threads busy for 2 x 64 dependent LDS round trips.
    double pa, pc;
    double2 pd;
    {
        const double glm = sR[s][cc][2], cpl = sR[s][cc][3];
        double diag = 1.0, sup = 0.0;
        double2 rhs = sD[s][cc][1];
        if (s + 1 < TCO_SEGS) {
            diag = 1.0 - cpl * sR[s + 1][cc][0];
            sup = -cpl * sR[s + 1][cc][1];
            const double2 df = sD[s + 1][cc][0];
            rhs.x -= cpl * df.x;
            rhs.y -= cpl * df.y;
        }
        const double r = bz_recip<2>(diag);
        pa = glm * r; pc = sup * r;
        pd = make_double2(rhs.x * r, rhs.y * r);
    }
    tco_lds_barrier();                                   // sR / sD are read; their storage becomes the two exchange buffers
    constexpr int NT = TCO_SEGS * TCO_COLS;
    static_assert(sizeof(sR) >= 4 * NT * sizeof(double) && sizeof(sD) >= 4 * NT * sizeof(double), "exchange buffers alias sR / sD");
    double *xb[2] = {&sR[0][0][0], (double *)&sD[0][0][0]};
    int pb = 0;
#pragma unroll
    for (int h = 1; h < TCO_SEGS; h <<= 1) {
        double *B = xb[pb];
        B[t] = pa; B[NT + t] = pc; B[2 * NT + t] = pd.x; B[3 * NT + t] = pd.y;
        tco_lds_barrier();
        const int tm = t - h * TCO_COLS, tp = t + h * TCO_COLS;
        double am = 0.0, cm = 0.0, ap = 0.0, cp = 0.0;
        double2 dm = make_double2(0.0, 0.0), dq = make_double2(0.0, 0.0);
        if (s - h >= 0) { am = B[tm]; cm = B[NT + tm]; dm = make_double2(B[2 * NT + tm], B[3 * NT + tm]); }
        if (s + h < TCO_SEGS) { ap = B[tp]; cp = B[NT + tp]; dq = make_double2(B[2 * NT + tp], B[3 * NT + tp]); }
        const double r = bz_recip<2>(1.0 - pa * cm - pc * ap);
        pd.x = (pd.x - pa * dm.x - pc * dq.x) * r;
        pd.y = (pd.y - pa * dm.y - pc * dq.y) * r;
        pa = -pa * am * r;
        pc = -pc * cp * r;
        pb ^= 1;
    }
    // ---- D: back substitution ----
    const double2 Xs = pd;
    double2 Xp = make_double2(0.0, 0.0);
    {
        double *B = xb[pb];
        B[2 * NT + t] = pd.x; B[3 * NT + t] = pd.y;
        tco_lds_barrier();
        if (s > 0) Xp = make_double2(B[2 * NT + t - TCO_COLS], B[3 * NT + t - TCO_COLS]);
    }
    double sum = 0.0;
#pragma unroll
    for (int j = 0; j < M; ++j) {
        if (j < L) {
            if (j == L - 1) dp[j] = Xs;
            else {
                dp[j].x -= g[j] * Xp.x + h[j] * Xs.x;
                dp[j].y -= g[j] * Xp.y + h[j] * Xs.y;
            }
            sum += dp[j].x;
        }
    }
    if (mean_column && grp == 0) {       // block-uniform branch: the global mean of phi is the z-mean of the (0, 0) column
        if (cc == 0) sSum[s] = sum;
        tco_lds_barrier();
        double tot = 0.0;
        for (int m = 0; m < TCO_SEGS; ++m) tot += sSum[m];
        if (cc == 0) {
            const double mean = tot / Nz;
#pragma unroll
            for (int j = 0; j < M; ++j)
                if (j < L) dp[j].x -= mean;
        }
    }
#pragma unroll
    for (int j = 0; j < M; ++j)
        if (j < L && live) col[lstride * (k0 + j)] = dp[j];
    tco_lds_barrier();                                   // the exchange buffers are rewritten by the next group
    };
    // EXACT: the first group runs outside the loop.  The memory counter returns in order and the compiler derives its waits from what
    // may be outstanding on ANY path into a point: entered straight from the prologue the loop head would see "rows still loading,
    // no stores behind them" on one path and "rows + 8 stores" on the other, and settle for waits that drain the stores of the
    // previous group in every trip (s_waitcnt vmcnt(17..10) through phase A).  Peeled, both paths into the head look alike.
    int grp = grp0 + (int)blockIdx.x;
    if constexpr (EXACT) { trip(grp); grp += gridDim.x; }
#pragma nounroll
    for (; grp < ngroups; grp += gridDim.x) trip(grp);
}

// segments per column: 64 for 128 <= Nz <= 512 (the tuned shape), 16 / 8 for shorter columns, where the sequential kernel's
// NXH * Ny threads of Nz dependent steps each leave the chip idle (64^3: 39 us sequential).  Every segment holds >= 2 rows.
static int tridiag_coop_segs(const bz_ctx *ctx, int)
{
    const int Nz = ctx->dg.Nz;
    if (ctx->tune.no_tridiag_coop) return 0;
    if (Nz >= 128 && Nz <= 64 * TCO_M) return 64;
    if (Nz >= 32 && Nz < 128) return 16;
    if (Nz >= 16 && Nz < 32) return 8;
    return 0;
}

// Thomas solve of this context's spectral block, in place: the cooperative kernel when the shape allows it, else the sequential one
// kx_lo / kx_hi (kx-major spectrum only, ctx->kxmajor): the wavenumber range of one chunk of the pipeline; default: every column
int bzi_tridiag_launch(bz_ctx *ctx, double *hat, double scale, int Ny, int mean_column, int kx_lo, int kx_hi)
{
    const long long plane = (long long)ctx->NXH * Ny;
    if (const int segs = tridiag_coop_segs(ctx, Ny)) {
        const int nxh_real = ctx->dg.bounded_x ? ctx->dg.Nx : ctx->dg.Nx / 2 + 1;      // as in bzi_poisson_setup
        double *d_cols = ctx->d_lower;
        const int Nz = ctx->dg.Nz;
        TriCols C{d_cols, d_cols + Nz, d_cols + 2 * Nz, d_cols + 3 * Nz, d_cols + 3 * Nz + nxh_real};
        const bool chunk = ctx->kxmajor && kx_hi > kx_lo;      // Ny is a multiple of TCO_COLS there: chunks are whole column groups
        const int grp0 = chunk ? (int)((long long)kx_lo * Ny / TCO_COLS) : 0;
        const int ngroups = chunk ? (int)((long long)kx_hi * Ny / TCO_COLS) : (int)((plane + TCO_COLS - 1) / TCO_COLS);
        const int nrun = ngroups - grp0;
        // 64 segments = 512 threads: one workgroup per CU resident (TCO_MIN_WAVES / 2), each walks ngroups / grid column groups (a multiple of the grid keeps the tail short)
        const int resident = (TCO_MIN_WAVES / 2) * ctx->num_cus * (segs == 64 ? 1 : 64 / segs);
        const int per_block = (nrun + resident - 1) / resident;
        dim3 grid((unsigned)((nrun + per_block - 1) / per_block)), block(TCO_COLS * segs);
        const int kyf = (ctx->slab_mode || ctx->xf) ? 1 : 0;
        const long long lstride = ctx->kxmajor ? (long long)Ny : plane, kxs = ctx->kxmajor ? (long long)(Nz + ctx->kx_pad) * Ny : 0;
        if (mean_column && grp0 != 0) mean_column = 0;
#define TCO_GO(SEGS, M, EXACT) hipLaunchKernelGGL((k_tridiag_coop<SEGS, M, EXACT>), grid, block, 0, ctx->stream, ctx->NXH, Ny, Nz, ctx->kx0, nxh_real, kyf, C, (double2 *)hat, scale, mean_column, ngroups, lstride, kxs, grp0)
        const bool whole = plane % TCO_COLS == 0;
        if (segs == 64 && Nz == 64 * 8 && whole) TCO_GO(64, 8, true);      // (128 segments of 4 rows, 1024 threads: the same 0.48 ms)
        else if (segs == 64 && Nz == 64 * 4 && whole) TCO_GO(64, 4, true);
        else if (segs == 64 && Nz == 64 * 2 && whole) TCO_GO(64, 2, true);
        else if (segs == 64) TCO_GO(64, TCO_M, false);
        else if (segs == 16) TCO_GO(16, TCO_M, false);
        else TCO_GO(8, TCO_M, false);
#undef TCO_GO
    } else {
        hipLaunchKernelGGL(k_tridiag_solve, dim3((unsigned)((plane + 63) / 64)), dim3(64), 0, ctx->stream, ctx->NXH, Ny, ctx->dg.Nz,
                           ctx->d_lower, ctx->d_ibeta, ctx->d_tfac, (double2 *)hat, scale, mean_column);
    }
    BZ_LAUNCH_CHECK();
    return BZ_OK;
}

__global__ __launch_bounds__(TX *TY) void k_phi_scatter(DevGrid g, double *__restrict__ phi,
                                                       const double *__restrict__ src)
{
    int i = blockIdx.x * TX + threadIdx.x, j = blockIdx.y * TY + threadIdx.y, k = blockIdx.z;
    if (i >= g.Nx || j >= g.Ny) return;
    phi[g.idx(i, j, k)] = src[(long long)i + (long long)g.Nx * ((long long)j + (long long)g.Ny * k)];
}

int bzi_poisson_setup(bz_ctx *ctx, const double *h_rho /* Nz+2Hz, halo-inclusive */)
{
    const DevGrid &g = ctx->dg;
    const int Nx = g.Nx, Nz = g.Nz, Hz = g.Hz;
    const bool slab = ctx->slab_mode;
    // Bounded x (2-D x-z models with walls): Nx cosine modes instead of the Nx / 2 + 1 wavenumbers of the half spectrum
    const int nxh_real = g.bounded_x ? Nx : Nx / 2 + 1;
    // hand-written x transforms + transposed spectrum: Nx a power of two in [16, 1024] (one team of Nx / 8 <= 128 threads per row, 8 rows per
    // workgroup in 45 KiB of LDS)
    const bool pow2 = (Nx & (Nx - 1)) == 0, three_pow2 = Nx % 3 == 0 && ((Nx / 3) & (Nx / 3 - 1)) == 0;      // 96, 192, 384, 768: one radix-3 stage
    const bool xf_shape = ((pow2 && Nx >= 16 && Nx <= 1024) || (three_pow2 && Nx >= 96 && Nx <= 768)) && g.Ny % 8 == 0 && !ctx->tune.no_xfft;
    ctx->xf = !slab && xf_shape && (g.wrap_y || g.bounded_y);
    if (g.bounded_y && (!ctx->xf || g.Ny > 4096 || (g.Ny & 1))) {
        ctx->last_error = "(Periodic, Bounded, Bounded): the cosine-transform solve rides the hand-written x transforms — Nx a power of two in "
                          "[16, 1024] or 3 * 2^m in [96, 768], Ny a multiple of 8";
        return BZ_ERR_UNSUPPORTED;
    }
    ctx->xf_slab = slab && xf_shape;
    int Ny = g.Ny;                       // rows of the spectral block: local rows, or ALL rows in slab mode
    if (slab) {
        ctx->nkx = (nxh_real + ctx->y_nranks - 1) / ctx->y_nranks;
        ctx->kx0 = ctx->y_rank * ctx->nkx;
        ctx->NXH = ctx->nkx;
        Ny = ctx->Ny_global;
    } else {
        ctx->NXH = nxh_real;
        ctx->nkx = nxh_real;
        ctx->kx0 = 0;
    }
    const size_t nreal = (size_t)Nx * g.Ny * Nz, nhat = (size_t)ctx->NXH * Ny * Nz;

    // ---- host column coefficients (anelastic_pressure_solver.jl:39-78) ----
    std::vector<double> dzc(Nz + 2 * Hz), dzf(Nz + 1 + 2 * Hz);
    BZ_HIP(hipMemcpy(dzc.data(), g.dzc - Hz, dzc.size() * sizeof(double), hipMemcpyDeviceToHost));
    BZ_HIP(hipMemcpy(dzf.data(), g.dzf - Hz, dzf.size() * sizeof(double), hipMemcpyDeviceToHost));
    const double *rho = h_rho + Hz;
    std::vector<double> lower(Nz, 0.0), diag0(Nz), mass(Nz), lam_x(nxh_real), lam_y(Ny);
    for (int k = 0; k < Nz - 1; ++k) lower[k] = (0.5 * (rho[k] + rho[k + 1])) / dzf[k + 1 + Hz];
    for (int k = 0; k < Nz; ++k) {
        double up = (k < Nz - 1) ? lower[k] : 0.0;
        double dn = (k > 0) ? lower[k - 1] : 0.0;
        diag0[k] = (k == 0) ? -up : (k == Nz - 1) ? -dn : -(up + dn);
        mass[k] = rho[k] * dzc[k + Hz];
    }
    const double pi = 3.14159265358979323846;
    for (int i = 0; i < nxh_real; ++i) { double s = 2.0 * std::sin(i * pi / (g.bounded_x ? 2.0 * Nx : (double)Nx)) / g.dx; lam_x[i] = s * s; }
    // Bounded y: the cosine modes of the staggered Neumann problem (Oceananigans poisson_eigenvalues(N, L, dim, ::Bounded))
    for (int j = 0; j < Ny; ++j) { double s = 2.0 * std::sin(j * pi / (g.bounded_y ? 2.0 * Ny : (double)Ny)) / g.dy; lam_y[j] = s * s; }

    double *d_cols = nullptr;
    size_t ncols = (size_t)3 * Nz + nxh_real + Ny;
    BZ_HIP(hipMalloc(&d_cols, ncols * sizeof(double)));
    ctx->d_lower = d_cols;
    double *d_diag0 = d_cols + Nz, *d_mass = d_cols + 2 * Nz, *d_lx = d_cols + 3 * Nz, *d_ly = d_lx + nxh_real;
    BZ_HIP(hipMemcpy(ctx->d_lower, lower.data(), Nz * sizeof(double), hipMemcpyHostToDevice));
    BZ_HIP(hipMemcpy(d_diag0, diag0.data(), Nz * sizeof(double), hipMemcpyHostToDevice));
    BZ_HIP(hipMemcpy(d_mass, mass.data(), Nz * sizeof(double), hipMemcpyHostToDevice));
    BZ_HIP(hipMemcpy(d_lx, lam_x.data(), nxh_real * sizeof(double), hipMemcpyHostToDevice));
    BZ_HIP(hipMemcpy(d_ly, lam_y.data(), Ny * sizeof(double), hipMemcpyHostToDevice));

    if (!slab) {
        BZ_HIP(hipMalloc(&ctx->d_rhs, nreal * sizeof(double)));
        // (+ room for the phantom lines of the kx-major layout: ctx->kx_pad lines per wavenumber, below)
        BZ_HIP(hipMalloc(&ctx->d_hat, (nhat + (size_t)ctx->NXH * 4 * Ny) * sizeof(hipfftDoubleComplex)));
        BZ_HIP(hipMemset(ctx->d_hat, 0, (nhat + (size_t)ctx->NXH * 4 * Ny) * sizeof(hipfftDoubleComplex)));
    }
    BZ_HIP(hipMalloc(&ctx->d_ibeta, nhat * sizeof(double)));
    BZ_HIP(hipMalloc(&ctx->d_tfac, nhat * sizeof(double)));

    TriCols C{ctx->d_lower, d_diag0, d_mass, d_lx, d_ly};
    long long plane = (long long)ctx->NXH * Ny;
    hipLaunchKernelGGL(k_tridiag_setup, dim3((unsigned)((plane + 255) / 256)), dim3(256), 0, 0, ctx->NXH, Ny, Nz, ctx->kx0, nxh_real, (slab || ctx->xf) ? 1 : 0, C,
                       ctx->d_ibeta, ctx->d_tfac);
    BZ_HIP(hipGetLastError());
    BZ_HIP(hipDeviceSynchronize());

    if (ctx->xf || ctx->xf_slab) {       // twiddles of the hand-written x transforms
        const int nw = 3 * Nx / 4;
        std::vector<double> w(2 * (size_t)nw);
        for (int t = 0; t < nw; ++t) {
            const double a = (double)(2.0 * pi * (double)t / (double)Nx);
            w[2 * t] = std::cos(a);
            w[2 * t + 1] = -std::sin(a);
        }
        BZ_HIP(hipMalloc(&ctx->d_wtab, w.size() * sizeof(double)));
        BZ_HIP(hipMemcpy(ctx->d_wtab, w.data(), w.size() * sizeof(double), hipMemcpyHostToDevice));
    }
    if (slab) return BZ_OK;          // horizontal transforms are the caller's (distributed) in slab mode
    // ---- rocFFT plans: 2-D (y,x) transforms batched over z ----
    int n[2] = {Ny, Nx};
    if (g.bounded_x)       // half spectrum of the permuted rows between the real transform and the cosine combination
        BZ_HIP(hipMalloc(&ctx->d_dctx, (size_t)(Nx / 2 + 1) * Nz * sizeof(hipfftDoubleComplex)));
    if (!ctx->xf && Ny == 1) {      // Flat y: rows only
        int n1[1] = {Nx};
        BZ_FFT(hipfftPlanMany(&ctx->plan_fwd, 1, n1, nullptr, 1, 0, nullptr, 1, 0, HIPFFT_D2Z, Nz));
        BZ_FFT(hipfftPlanMany(&ctx->plan_inv, 1, n1, nullptr, 1, 0, nullptr, 1, 0, HIPFFT_Z2D, Nz));
        ctx->plans_ok = true;
        BZ_FFT(hipfftSetStream(ctx->plan_fwd, ctx->stream));
        BZ_FFT(hipfftSetStream(ctx->plan_inv, ctx->stream));
    } else if (!ctx->xf) {
        BZ_FFT(hipfftPlanMany(&ctx->plan_fwd, 2, n, nullptr, 1, 0, nullptr, 1, 0, HIPFFT_D2Z, Nz));
        BZ_FFT(hipfftPlanMany(&ctx->plan_inv, 2, n, nullptr, 1, 0, nullptr, 1, 0, HIPFFT_Z2D, Nz));
        ctx->plans_ok = true;
        BZ_FFT(hipfftSetStream(ctx->plan_fwd, ctx->stream));
        BZ_FFT(hipfftSetStream(ctx->plan_inv, ctx->stream));
    }
    if (ctx->xf) {
        int ny[1] = {Ny};
        BZ_FFT(hipfftPlanMany(&ctx->plan_y, 1, ny, ny, 1, Ny, ny, 1, Ny, HIPFFT_Z2Z, ctx->NXH * Nz));
        BZ_FFT(hipfftSetStream(ctx->plan_y, ctx->stream));
        // Round 6: kx-major spectrum and the chunked middle of the solve.  The y transform, the vertical solves and the inverse y transform
        // each read and write the whole half spectrum (1.08 GB at 512^3); a range of wavenumbers of ~220 MB that takes the three passes
        // back to back stays in the 256 MB Infinity Cache between them (tools/mall_probe.hip: three in-place streaming passes over 1.03 GB
        // 1.24 ms whole, 0.84 ms in 128 - 256 MB chunks).  That needs a wavenumber range to be contiguous: hatT[(kx Nz + k) Ny + ky]
        // (xf_addr, k_tridiag_coop::col_addr).  Shapes: the cooperative tridiagonal kernel's, Ny a multiple of its column group.
        const size_t spec = (size_t)ctx->NXH * Nz * Ny * sizeof(hipfftDoubleComplex);
        const size_t chunk_bytes = ctx->tune.poisson_kx_chunk_kb > 0 ? (size_t)ctx->tune.poisson_kx_chunk_kb << 10      // (tests: small grids take the pipeline too)
                                   : (size_t)(ctx->tune.poisson_kx_chunk_mb > 0 ? ctx->tune.poisson_kx_chunk_mb : 256) << 20;
        if (ctx->tune.poisson_kxmajor && !g.bounded_y && tridiag_coop_segs(ctx, Ny) && Ny % TCO_COLS == 0 && spec > chunk_bytes + chunk_bytes / 2) {
            const int nch = (int)((spec + chunk_bytes - 1) / chunk_bytes);
            ctx->kx_cw = (ctx->NXH + nch - 1) / nch;
            ctx->kx_nch = (ctx->NXH + ctx->kx_cw - 1) / ctx->kx_cw;
            const int last = ctx->NXH - (ctx->kx_nch - 1) * ctx->kx_cw;
            // consecutive wavenumbers are Nz + kx_pad lines apart: with none, the 257 segments one block of the forward x transform writes
            // sit a power of two apart (4 MB at 512^3); the phantom lines (zeros, transformed along with the rest) break that
            ctx->kx_pad = std::max(0, std::min(4, ctx->tune.poisson_kx_pad));
            const int lines = Nz + ctx->kx_pad;
            BZ_FFT(hipfftPlanMany(&ctx->plan_yc, 1, ny, ny, 1, Ny, ny, 1, Ny, HIPFFT_Z2Z, ctx->kx_cw * lines));
            BZ_FFT(hipfftSetStream(ctx->plan_yc, ctx->stream));
            if (last != ctx->kx_cw) {
                BZ_FFT(hipfftPlanMany(&ctx->plan_yc_last, 1, ny, ny, 1, Ny, ny, 1, Ny, HIPFFT_Z2Z, last * lines));
                BZ_FFT(hipfftSetStream(ctx->plan_yc_last, ctx->stream));
            }
            ctx->kxmajor = true;
        }
    }
    // chunk plans of the L3-resident pipeline
    int ch = ctx->tune.poisson_chunk;
    if (ch > 0 && ch < Nz && Nz % ch == 0 && !ctx->xf) {
        BZ_FFT(hipfftPlanMany(&ctx->plan_fwd_c, 2, n, nullptr, 1, 0, nullptr, 1, 0, HIPFFT_D2Z, ch));
        BZ_FFT(hipfftPlanMany(&ctx->plan_inv_c, 2, n, nullptr, 1, 0, nullptr, 1, 0, HIPFFT_Z2D, ch));
        BZ_FFT(hipfftSetStream(ctx->plan_fwd_c, ctx->stream));
        BZ_FFT(hipfftSetStream(ctx->plan_inv_c, ctx->stream));
        ctx->pchunk = ch;
    }
    return BZ_OK;
}

// 2-D transforms of levels k0 .. k0 + pchunk - 1
int bzi_fft_chunk(bz_ctx *ctx, int k0, bool forward)
{
    const DevGrid &g = ctx->dg;
    double *real = ctx->d_rhs + (size_t)g.Nx * g.Ny * k0;
    hipfftDoubleComplex *hat = ctx->d_hat + (size_t)ctx->NXH * g.Ny * k0;
    if (forward) BZ_FFT(hipfftExecD2Z(ctx->plan_fwd_c, real, hat));
    else BZ_FFT(hipfftExecZ2D(ctx->plan_inv_c, hat, real));
    return BZ_OK;
}

void bzi_poisson_teardown(bz_ctx *ctx)
{
    if (ctx->slab_plans_ok) {
        hipfftDestroy(ctx->slab_plan_x_fwd);
        hipfftDestroy(ctx->slab_plan_x_inv);
        hipfftDestroy(ctx->slab_plan_y);
        ctx->slab_plans_ok = false;
    }
    if (ctx->xf && ctx->plan_y) hipfftDestroy(ctx->plan_y);
    if (ctx->plan_yc) hipfftDestroy(ctx->plan_yc);
    if (ctx->plan_yc_last) hipfftDestroy(ctx->plan_yc_last);
    ctx->plan_yc = ctx->plan_yc_last = 0; ctx->kxmajor = false;
    if (ctx->d_wtab) hipFree(ctx->d_wtab);
    ctx->plan_y = 0; ctx->d_wtab = nullptr; ctx->xf = ctx->xf_slab = false;
    if (ctx->plans_ok) {
        hipfftDestroy(ctx->plan_fwd);
        hipfftDestroy(ctx->plan_inv);
        if (ctx->pchunk) { hipfftDestroy(ctx->plan_fwd_c); hipfftDestroy(ctx->plan_inv_c); ctx->pchunk = 0; }
        ctx->plans_ok = false;
    }
    if (ctx->d_dctx) hipFree(ctx->d_dctx);
    ctx->d_dctx = nullptr;
    if (ctx->d_lower) hipFree(ctx->d_lower);
    if (ctx->d_rhs) hipFree(ctx->d_rhs);
    if (ctx->d_hat) hipFree(ctx->d_hat);
    if (ctx->d_ibeta) hipFree(ctx->d_ibeta);
    if (ctx->d_tfac) hipFree(ctx->d_tfac);
    ctx->d_lower = ctx->d_rhs = ctx->d_ibeta = ctx->d_tfac = nullptr;
    ctx->d_hat = nullptr;
}

// solve!(phi, FourierTridiagonalPoissonSolver) on the source term held in ctx->d_rhs; the zero-mean
// solution is left in ctx->d_rhs (contiguous Nx*Ny*Nz).
// Cosine transform along a Bounded y through the complex FFT of the same length (Makhoul 1980), on the ky-contiguous lines of the
// transposed spectrum.  The transform is real-linear and the lines are complex (x is already transformed), so the combination steps are
// written in their complex-linear form.  N = Ny (even), w_k = exp(-i pi k / (2N)):
//   forward  (DCT-II, X_k = 2 sum_n x_n cos(pi k (2n+1) / (2N))):  v_n = x_{2n}, v_{N-1-n} = x_{2n+1};  V = FFT(v);  X_k = w_k V_k + conj(w_k) V_{(N-k) mod N}
//   backward (its inverse up to N, like the unnormalised FFT pair): V_k = conj(w_k) (X_k - i X_{N-k}) / 2 (X_N = 0);  v = N IFFT(V);  x_{2n} = v_n, x_{2n+1} = v_{N-1-n}
// MODE 0: forward permutation, 1: forward combination, 2: backward combination, 3: backward permutation.  One line per workgroup, staged in LDS.
template <int MODE>
__global__ __launch_bounds__(256) void k_dct_line(double2 *__restrict__ hat, int N)
{
    extern __shared__ double2 dct_line[];
    double2 *line = hat + (long long)blockIdx.x * N;
    for (int m = threadIdx.x; m < N; m += 256) dct_line[m] = line[m];
    __syncthreads();
    for (int m = threadIdx.x; m < N; m += 256) {
        double2 out;
        if (MODE == 0) out = (m < N / 2) ? dct_line[2 * m] : dct_line[2 * (N - 1 - m) + 1];
        else if (MODE == 3) out = (m & 1) ? dct_line[N - 1 - (m - 1) / 2] : dct_line[m / 2];
        else {
            const double ang = 3.14159265358979323846 * (double)m / (2.0 * (double)N);      // w_m = cs - i sn
            const double sn = sin(ang), cs = cos(ang);
            if (MODE == 1) {
                const double2 a = dct_line[m], b = dct_line[m ? N - m : 0];
                // w a + conj(w) b
                out = make_double2(cs * a.x + sn * a.y + cs * b.x - sn * b.y, cs * a.y - sn * a.x + cs * b.y + sn * b.x);
            } else {
                const double2 a = dct_line[m], b = m ? dct_line[N - m] : make_double2(0.0, 0.0);
                const double2 d = make_double2(a.x + b.y, a.y - b.x);          // X_m - i X_{N-m}
                out = make_double2(0.5 * (cs * d.x - sn * d.y), 0.5 * (cs * d.y + sn * d.x));      // conj(w) d / 2
            }
        }
        line[m] = out;
    }
}

// The same along a Bounded x of a 2-D model (Flat y: rows of Nx real cells, one row per level): the permuted row goes through the real
// row transform the Flat-y solve already owns (half spectrum V_0 .. V_{N/2}, V_{N-k} = conj V_k), and the combination writes the N cosine
// coefficients as complex numbers with zero imaginary part — the columns the tridiagonal kernels take.
// MODE 0: permute a real row in place; 1: half spectrum -> N coefficients; 2: N coefficients -> half spectrum; 3: un-permute in place.
template <int MODE>
__global__ __launch_bounds__(256) void k_dctx_row(double *__restrict__ real, double2 *__restrict__ half, double2 *__restrict__ full, int N)
{
    extern __shared__ double2 dct_line[];
    const int NH = N / 2 + 1;
    const long long row = blockIdx.x;
    if (MODE == 0 || MODE == 3) {
        double *sm = (double *)dct_line, *x = real + row * N;
        for (int m = threadIdx.x; m < N; m += 256) sm[m] = x[m];
        __syncthreads();
        for (int m = threadIdx.x; m < N; m += 256)
            x[m] = (MODE == 0) ? ((m < N / 2) ? sm[2 * m] : sm[2 * (N - 1 - m) + 1]) : ((m & 1) ? sm[N - 1 - (m - 1) / 2] : sm[m / 2]);
    } else if (MODE == 1) {
        const double2 *V = half + row * NH;
        for (int m = threadIdx.x; m < N; m += 256) {
            const double ang = 3.14159265358979323846 * (double)m / (2.0 * (double)N);
            const double sn = sin(ang), cs = cos(ang);
            const double2 v = (m <= N / 2) ? V[m] : make_double2(V[N - m].x, -V[N - m].y);
            full[row * N + m] = make_double2(2.0 * (cs * v.x + sn * v.y), 0.0);      // X_m = w_m V_m + conj(w_m V_m) = 2 Re(w_m V_m)
        }
    } else {
        const double2 *X = full + row * N;
        for (int m = threadIdx.x; m < NH; m += 256) {
            const double ang = 3.14159265358979323846 * (double)m / (2.0 * (double)N);
            const double sn = sin(ang), cs = cos(ang);
            const double a = X[m].x, b = m ? X[N - m].x : 0.0;      // real coefficients: V_m = conj(w_m) (X_m - i X_{N-m}) / 2
            half[row * NH + m] = make_double2(0.5 * (cs * a + sn * b), 0.5 * (sn * a - cs * b));
        }
    }
}

int bzi_xf_y(bz_ctx *ctx, bool forward)
{
    const DevGrid &g = ctx->dg;
    if (!g.bounded_y) {
        BZ_FFT(hipfftExecZ2Z(ctx->plan_y, ctx->d_hat, ctx->d_hat, forward ? HIPFFT_FORWARD : HIPFFT_BACKWARD));
        return BZ_OK;
    }
    const unsigned lines = (unsigned)(ctx->NXH * g.Nz);
    const size_t lds = (size_t)g.Ny * sizeof(double2);
    double2 *hat = (double2 *)ctx->d_hat;
    if (forward) {
        hipLaunchKernelGGL(k_dct_line<0>, dim3(lines), dim3(256), lds, ctx->stream, hat, g.Ny);
        BZ_FFT(hipfftExecZ2Z(ctx->plan_y, ctx->d_hat, ctx->d_hat, HIPFFT_FORWARD));
        hipLaunchKernelGGL(k_dct_line<1>, dim3(lines), dim3(256), lds, ctx->stream, hat, g.Ny);
    } else {
        hipLaunchKernelGGL(k_dct_line<2>, dim3(lines), dim3(256), lds, ctx->stream, hat, g.Ny);
        BZ_FFT(hipfftExecZ2Z(ctx->plan_y, ctx->d_hat, ctx->d_hat, HIPFFT_BACKWARD));
        hipLaunchKernelGGL(k_dct_line<3>, dim3(lines), dim3(256), lds, ctx->stream, hat, g.Ny);
    }
    BZ_LAUNCH_CHECK();
    return BZ_OK;
}

// the middle of the solve on the transposed half spectrum of the hand-written x transforms: y transform, vertical solves, inverse y
// transform — whole-spectrum passes, or (ctx->kxmajor) chunk by chunk of wavenumbers with the three passes of a chunk back to back
int bzi_xf_middle(bz_ctx *ctx)
{
    const DevGrid &g = ctx->dg;
    const double scale = 1.0 / ((double)g.Nx * (double)g.Ny);
    int rc;
    if (ctx->kxmajor) {
        ProfileScope ps(ctx, "poisson_fft_y+tridiagonal");
        hipfftDoubleComplex *hat = (hipfftDoubleComplex *)ctx->d_hat;
        for (int c = 0; c < ctx->kx_nch; ++c) {
            const int kx_lo = c * ctx->kx_cw, kx_hi = std::min(ctx->NXH, kx_lo + ctx->kx_cw);
            hipfftHandle plan = (kx_hi - kx_lo == ctx->kx_cw) ? ctx->plan_yc : ctx->plan_yc_last;
            hipfftDoubleComplex *p = hat + (size_t)kx_lo * (g.Nz + ctx->kx_pad) * g.Ny;
            BZ_FFT(hipfftExecZ2Z(plan, p, p, HIPFFT_FORWARD));
            if ((rc = bzi_tridiag_launch(ctx, (double *)ctx->d_hat, scale, g.Ny, 1, kx_lo, kx_hi))) return rc;
            BZ_FFT(hipfftExecZ2Z(plan, p, p, HIPFFT_BACKWARD));
        }
        return BZ_OK;
    }
    {
        ProfileScope ps(ctx, "poisson_fft_y_forward");
        if ((rc = bzi_xf_y(ctx, true))) return rc;
    }
    {
        ProfileScope ps(ctx, "poisson_tridiagonal");
        if ((rc = bzi_tridiag_launch(ctx, (double *)ctx->d_hat, scale, g.Ny, 1))) return rc;
    }
    ProfileScope ps(ctx, "poisson_fft_y_inverse");
    return bzi_xf_y(ctx, false);
}

int bzi_poisson_spectral(bz_ctx *ctx)
{
    const DevGrid &g = ctx->dg;
    if (ctx->xf) {
        int rc;
        {
            ProfileScope ps(ctx, "poisson_fft_x_forward");
            if ((rc = bzi_xf_forward(ctx, nullptr, 1.0, nullptr))) return rc;
        }
        if ((rc = bzi_xf_middle(ctx))) return rc;
        ProfileScope ps(ctx, "poisson_fft_x_inverse");
        return bzi_xf_inverse(ctx);
    }
    const size_t lds_x = (size_t)g.Nx * sizeof(double);
    {
        ProfileScope ps(ctx, "poisson_fft_forward");
        if (g.bounded_x) {      // cosine transform of every row (Flat y: one row per level)
            hipLaunchKernelGGL(k_dctx_row<0>, dim3(g.Nz), dim3(256), lds_x, ctx->stream, ctx->d_rhs, nullptr, nullptr, g.Nx);
            BZ_FFT(hipfftExecD2Z(ctx->plan_fwd, ctx->d_rhs, ctx->d_dctx));
            hipLaunchKernelGGL(k_dctx_row<1>, dim3(g.Nz), dim3(256), 0, ctx->stream, nullptr, (double2 *)ctx->d_dctx, (double2 *)ctx->d_hat, g.Nx);
        } else BZ_FFT(hipfftExecD2Z(ctx->plan_fwd, ctx->d_rhs, ctx->d_hat));
    }
    {
        ProfileScope ps(ctx, "poisson_tridiagonal");
        int rct = bzi_tridiag_launch(ctx, (double *)ctx->d_hat, 1.0 / ((double)g.Nx * (double)g.Ny), g.Ny, 1);
        if (rct) return rct;
    }
    {
        ProfileScope ps(ctx, "poisson_fft_inverse");
        if (g.bounded_x) {
            hipLaunchKernelGGL(k_dctx_row<2>, dim3(g.Nz), dim3(256), 0, ctx->stream, nullptr, (double2 *)ctx->d_dctx, (double2 *)ctx->d_hat, g.Nx);
            BZ_FFT(hipfftExecZ2D(ctx->plan_inv, ctx->d_dctx, ctx->d_rhs));
            hipLaunchKernelGGL(k_dctx_row<3>, dim3(g.Nz), dim3(256), lds_x, ctx->stream, ctx->d_rhs, nullptr, nullptr, g.Nx);
        } else BZ_FFT(hipfftExecZ2D(ctx->plan_inv, ctx->d_hat, ctx->d_rhs));
    }
    BZ_LAUNCH_CHECK();
    return BZ_OK;
}

// compute_anelastic_source_term! + solve! from the momentum in `s` (predictor == nullptr) or from the predictor arrays, for the fused tiers
// of the whole-step seam: with the hand-written x transforms the source term is evaluated inside the forward x pass (no rhs round trip,
// as the lean seam does); otherwise source kernel + library transforms.  The zero-mean solution is left in ctx->d_rhs.
int bzi_poisson_from_momentum(bz_ctx *ctx, const bz_state *s, double dt, const bz_prognostic *predictor)
{
    const DevGrid &g = ctx->dg;
    int rc;
    if (ctx->xf && !ctx->pchunk) {
        {
            ProfileScope ps(ctx, "poisson_source_term+fft_x");
            if ((rc = bzi_xf_forward(ctx, s, dt, predictor))) return rc;
        }
        if ((rc = bzi_xf_middle(ctx))) return rc;
        ProfileScope ps(ctx, "poisson_fft_x_inverse");
        return bzi_xf_inverse(ctx);
    }
    if ((rc = bzi_poisson_source_fused(ctx, s, dt, nullptr, predictor))) return rc;
    return bzi_poisson_spectral(ctx);
}

// solve_for_anelastic_pressure!(phi, solver, rhoU, dt)  (anelastic_pressure_solver.jl:84-88)
int bzi_poisson_solve(bz_ctx *ctx, const bz_state *s, double dt)
{
    const DevGrid &g = ctx->dg;
    dim3 grid((g.Nx + TX - 1) / TX, (g.Ny + TY - 1) / TY, g.Nz), block(TX, TY);
    {
        ProfileScope ps(ctx, "poisson_source_term");
        hipLaunchKernelGGL(k_poisson_source, grid, block, 0, ctx->stream, g, ctx->d_rhs, s->rho_u, s->rho_v,
                           s->rho_w, dt);
        BZ_LAUNCH_CHECK();
    }
    int rc = bzi_poisson_spectral(ctx);
    if (rc) return rc;
    {
        ProfileScope ps(ctx, "poisson_phi_scatter");
        hipLaunchKernelGGL(k_phi_scatter, grid, block, 0, ctx->stream, g, s->phi, ctx->d_rhs);
        BZ_LAUNCH_CHECK();
    }
    return BZ_OK;
}

extern "C" int bz_compute_pressure_correction(bz_ctx *ctx, const bz_state *s, double dt)
{
    if (!ctx || !s) return BZ_ERR_INVALID;
    if (ctx->slab_mode) {
        ctx->last_error = "bz_compute_pressure_correction: y-slab contexts solve through bz_poisson_source_term / "
                          "bz_spectral_tridiagonal_solve / bz_project_and_diagnose";
        return BZ_ERR_UNSUPPORTED;
    }
    double *mf[3] = {s->rho_u, s->rho_v, s->rho_w};
    int mk[3] = {BZ_HALO_XFACE, BZ_HALO_YFACE, 1};
    int rc = bzi_fill_halos_multi(ctx, mf, mk, 3);   // anelastic_time_stepping.jl:29
    if (rc) return rc;
    rc = bzi_poisson_solve(ctx, s, dt);
    if (rc) return rc;
    return bzi_fill_halo(ctx, s->phi, 0);            // anelastic_time_stepping.jl:36
}
