// bz_closure.hip — closure = SmagorinskyLilly() on the anelastic potential-temperature model (BASELINE configs[2]).
//   Breeze side (followed line by line):
//     dynamic stresses T_ij = rho x viscous flux, scalar fluxes J = rho x diffusive flux
//                                           /root/reference/src/TurbulenceClosures/TurbulenceClosures.jl:44-101
//     - d_j T_1j, - d_j T_2j, - d_j T_3j    /root/reference/src/AtmosphereModels/dynamics_kernel_functions.jl:80,100,128
//     - div J for theta, moisture           /root/reference/src/PotentialTemperatureFormulations/potential_temperature_tendency.jl:102,
//                                           /root/reference/src/AtmosphereModels/dynamics_kernel_functions.jl:157
//     N^2 = g dz(log theta_v)               /root/reference/src/AtmosphereModels/atmosphere_model_buoyancy.jl:46-68
//     compute_closure_fields!               /root/reference/src/AtmosphereModels/update_atmosphere_model_state.jl:218
//   Oceananigans side (0.110.14, Project.toml:43, not vendored; restated from its published Smagorinsky-Lilly form — DESIGN.md,
//   "parity unpinned"): nu_e = varsigma (C_s Delta)^2 sqrt(2 Sigma^2), varsigma = sqrt(1 - min(1, C_b N^2+/Sigma^2)),
//   viscous flux_ij = -2 nu Sigma_ij with nu averaged to the flux location, diffusive flux = -(nu/Pr) grad c.
// Two kernels: the eddy viscosity (one centre array), then one pass that adds the five divergences to the tendencies.  Every
// flux is recomputed by the two cells that share it instead of being stored: 9 stress + 6 scalar-flux arrays would cost
// 30 words/cell of traffic, the recomputation reads u, v, w, nu, theta, q through L2 (7 words/cell) and is arithmetic otherwise.
#include "bz_internal.h"

struct ClosureFields {
    const double *u, *v, *w, *nu, *theta, *q;
    double C2, Cb, Pr;
    int jofs;      // y-slab contexts: the viscosity kernel also covers rows -1 and Ny (launched over Ny + 2 rows with jofs = -1), which
                   // the tendency kernel then reads in place of the periodic wrap
};

// strain components; (i, j, k) may sit one cell outside the interior in x / y (u, v, w carry periodic halos there)
__device__ __forceinline__ double S11(const DevGrid &g, const double *u, long long n) { return (u[n + 1] - u[n]) * g.rdx; }
__device__ __forceinline__ double S22(const DevGrid &g, const double *v, long long n) { return (v[n + g.Sx] - v[n]) * g.rdy; }
__device__ __forceinline__ double S33(const DevGrid &g, const double *w, long long n, int k) { return (w[n + g.Sxy] - w[n]) * g.rdzc[k]; }
__device__ __forceinline__ double S12(const DevGrid &g, const double *u, const double *v, long long n)
{
    return ((u[n] - u[n - g.Sx]) * g.rdy + (v[n] - v[n - 1]) * g.rdx) * 0.5;
}
// at (x face i, z face k): zero on the walls (u has no-flux z halos there, w = 0)
__device__ __forceinline__ double S13(const DevGrid &g, const double *u, const double *w, long long n, int k)
{
    if (k <= 0 || k >= g.Nz) return 0.0;
    return ((u[n] - u[n - g.Sxy]) * g.rdzf[k] + (w[n] - w[n - 1]) * g.rdx) * 0.5;
}
__device__ __forceinline__ double S23(const DevGrid &g, const double *v, const double *w, long long n, int k)
{
    if (k <= 0 || k >= g.Nz) return 0.0;
    return ((v[n] - v[n - g.Sxy]) * g.rdzf[k] + (w[n] - w[n - g.Sx]) * g.rdy) * 0.5;
}

// XCD-aware block order: hardware deals consecutive workgroup ids round-robin to the 8 XCDs (each with its own L2); give every
// XCD a contiguous slab of (row, level) space so that the j +- 1 / k +- 1 neighbour rows of a block are hits in its own L2
__device__ __forceinline__ void xcd_block(int &bx, int &by, int &bz)
{
    const int nb = gridDim.x * gridDim.y * gridDim.z;
    int b = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
    if (nb % 8 == 0) b = (b % 8) * (nb / 8) + b / 8;
    bx = b % gridDim.x;
    by = (b / gridDim.x) % gridDim.y;
    bz = b / (gridDim.x * gridDim.y);
}

// ipi[k] = (pst / p_r[k])^(Rd/cpd), k = -1 .. Nz (k_inverse_exner_column): the only pow of the buoyancy gradient depends on k alone
__device__ __forceinline__ double log_theta_v(const DevGrid &g, const double *ipi, const double *T, const double *qv, long long n, int k)
{
    const double q = qv[n];
    const double Rm = (1.0 - (q + 0.0 + 0.0)) * g.Rd + q * g.Rv;
    return log(Rm / g.Rd * T[n] * ipi[k]);
}

// ... and so does the filter width: delta2[k] = cbrt(dx dy dz_k)^2 (round 2 evaluated cbrt per cell)
__global__ void k_inverse_exner_column(DevGrid g, double *ipi, double *delta2)
{
    for (int k = (int)threadIdx.x - 1; k <= g.Nz; k += blockDim.x) {
        ipi[k] = pow(g.pst / g.p_r[k], g.Rd / g.cpd);
        if (k < 0 || k >= g.Nz) continue;
        const double delta = cbrt(g.dx * g.dy * g.dzc[k]);
        delta2[k] = delta * delta;
    }
}

// A workgroup walks `kchunk` levels of its row: log(theta_v) of a cell enters N^2 of three levels, and the march evaluates it once (ring of
// three) instead of three times — the logarithm is the most expensive thing in this kernel.
#define SMAG_KCHUNK 1      // 8 measured slower (0.36 -> 0.46 ms at 256x256x128 Float32): the strain loads bound this kernel, not the logarithms
__global__ __launch_bounds__(256) void k_smagorinsky_viscosity(DevGrid g, ClosureFields F, const double *__restrict__ T,
                                                               const double *__restrict__ qv, const double *__restrict__ ipi,
                                                               const double *__restrict__ delta2, double *__restrict__ nu)
{
    int bx, by, bz;
    xcd_block(bx, by, bz);
    const int i = bx * 256 + threadIdx.x, j = by + F.jofs, kbeg = bz * SMAG_KCHUNK, kend = min(kbeg + SMAG_KCHUNK, g.Nz);
    if (i >= g.Nx) return;
    const long long sx = 1, sy = g.Sx, sz = g.Sxy;
    const double *u = F.u, *v = F.v, *w = F.w;
    long long n = g.idx(i, j, kbeg);
    double lm = log_theta_v(g, ipi, T, qv, n - sz, kbeg - 1), lc = log_theta_v(g, ipi, T, qv, n, kbeg);
    for (int k = kbeg; k < kend; ++k, n += sz) {
    const double s11 = S11(g, u, n), s22 = S22(g, v, n), s33 = S33(g, w, n, k);
    auto sq = [](double a) { return a * a; };
    // a Flat y direction (Ny = 1, no halo rows, 1/dy stored as 0): the corners at j and j + 1 coincide
    const long long cy = g.flat_y ? 0 : sy;
    const double a12 = ((sq(S12(g, u, v, n)) + sq(S12(g, u, v, n + sx))) / 2 +
                        (sq(S12(g, u, v, n + cy)) + sq(S12(g, u, v, n + cy + sx))) / 2) / 2;
    const double a13 = ((sq(S13(g, u, w, n, k)) + sq(S13(g, u, w, n + sx, k))) / 2 +
                        (sq(S13(g, u, w, n + sz, k + 1)) + sq(S13(g, u, w, n + sz + sx, k + 1))) / 2) / 2;
    const double a23 = ((sq(S23(g, v, w, n, k)) + sq(S23(g, v, w, n + cy, k))) / 2 +
                        (sq(S23(g, v, w, n + sz, k + 1)) + sq(S23(g, v, w, n + sz + cy, k + 1))) / 2) / 2;
    const double Sig2 = (s11 * s11 + s22 * s22 + s33 * s33) + 2 * a12 + 2 * a13 + 2 * a23;
    // N^2: T, q^v carry no-flux z halos, the reference pressure column its own first halo cell
    const double lp = log_theta_v(g, ipi, T, qv, n + sz, k + 1);
    const double bdn = g.g * ((lc - lm) * g.rdzf[k]), bup = g.g * ((lp - lc) * g.rdzf[k + 1]);
    const double N2 = (bdn + bup) / 2;
    const double N2p = fmax(0.0, N2);
    const double stab = (Sig2 == 0.0) ? 0.0 : sqrt(1.0 - fmin(1.0, F.Cb * N2p / Sig2));
    nu[n] = (stab * F.C2) * delta2[k] * sqrt(2 * Sig2);
    lm = lc; lc = lp;
    }
}

__global__ __launch_bounds__(256) void k_closure_tendencies(DevGrid g, ClosureFields F, double *__restrict__ Gu,
                                                            double *__restrict__ Gv, double *__restrict__ Gw,
                                                            double *__restrict__ Gth, double *__restrict__ Gq, double scale)
{
    int bx, by, bz;
    xcd_block(bx, by, bz);
    const int i = bx * 256 + threadIdx.x, j = by, k = bz;
    if (i >= g.Nx) return;
    const long long n = g.idx(i, j, k), sx = 1, sy = g.Sx, sz = g.Sxy;
    const double *u = F.u, *v = F.v, *w = F.w, *nu = F.nu;
    const double dx = g.dx, dy = g.dy, dz = g.dzc[k];
    const double rVc = g.rdx * (1.0 / dy) * g.rdzc[k];      // not g.rdy: that is the derivative quotient, stored as 0 on Flat grids
    const double rho = g.rho[k];
    const double Ax = dy * dz, Ay = dx * dz, Az = dx * dy;
    // eddy viscosity of the 3 x 3 x 3 neighbourhood that the flux locations of this cell touch
    double nun[3][3][3];      // [dk+1][dj+1][di+1]; the 8 corners are never used
    {
        // walls in x: the columns -1 and Nx hold the no-flux copy (bz_compute_closure_fields fills them)
        const long long oi[3] = {(i == 0 && !g.bounded_x) ? (long long)(g.Nx - 1) : -1, 0, (i == g.Nx - 1 && !g.bounded_x) ? -(long long)(g.Nx - 1) : 1};
        // y-slab (wrap_y == 0): rows -1 and Ny hold the viscosity the extended launch of k_smagorinsky_viscosity computed there
        const long long oj[3] = {((j == 0 && g.wrap_y) ? (long long)(g.Ny - 1) : -1) * sy, 0,
                                 ((j == g.Ny - 1 && g.wrap_y) ? -(long long)(g.Ny - 1) : 1) * sy};
        const long long ok[3] = {(k == 0) ? 0 : -sz, 0, (k == g.Nz - 1) ? 0 : sz};
#pragma unroll
        for (int c = 0; c < 3; ++c)
#pragma unroll
            for (int b = 0; b < 3; ++b)
#pragma unroll
                for (int a = 0; a < 3; ++a)
                    nun[c][b][a] = ((a != 1) + (b != 1) + (c != 1) == 3) ? 0.0 : nu[n + ok[c] + oj[b] + oi[a]];
    }
    auto NU = [&](int di, int dj, int dk) { return nun[dk + 1][dj + 1][di + 1]; };
    auto nu_ffc = [&](int di, int dj) {      // corner (i + di, j + dj) at level k
        return ((NU(di - 1, dj - 1, 0) + NU(di, dj - 1, 0)) / 2 + (NU(di - 1, dj, 0) + NU(di, dj, 0)) / 2) / 2;
    };
    auto nu_fcf = [&](int di, int dk) {      // x face i + di, z face k + dk, row j
        return ((NU(di - 1, 0, dk - 1) + NU(di, 0, dk - 1)) / 2 + (NU(di - 1, 0, dk) + NU(di, 0, dk)) / 2) / 2;
    };
    auto nu_cff = [&](int dj, int dk) {      // y face j + dj, z face k + dk, column i
        return ((NU(0, dj - 1, dk - 1) + NU(0, dj, dk - 1)) / 2 + (NU(0, dj - 1, dk) + NU(0, dj, dk)) / 2) / 2;
    };
    auto T11 = [&](int di) { return rho * (-2 * NU(di, 0, 0) * S11(g, u, n + di * sx)); };
    auto T22 = [&](int dj) { return rho * (-2 * NU(0, dj, 0) * S22(g, v, n + dj * sy)); };
    auto T33 = [&](int dk) { return g.rho[k + dk] * (-2 * NU(0, 0, dk) * S33(g, w, n + dk * sz, k + dk)); };
    auto T12 = [&](int di, int dj) { return rho * (-2 * nu_ffc(di, dj) * S12(g, u, v, n + di * sx + dj * sy)); };
    auto T13 = [&](int di, int dk) { return g.rho_f[k + dk] * (-2 * nu_fcf(di, dk) * S13(g, u, w, n + di * sx + dk * sz, k + dk)); };
    auto T23 = [&](int dj, int dk) { return g.rho_f[k + dk] * (-2 * nu_cff(dj, dk) * S23(g, v, w, n + dj * sy + dk * sz, k + dk)); };

    const double t12_00 = T12(0, 0), t13_00 = T13(0, 0), t23_00 = T23(0, 0);
    if (!(g.bounded_x && i == 0)) {   // x momentum at face i (walls in x: the wall face is never updated)
        const double div = (Ax * T11(0) - Ax * T11(-1)) + (g.flat_y ? 0.0 : Ay * T12(0, 1) - Ay * t12_00) + (Az * T13(0, 1) - Az * t13_00);
        Gu[n] -= scale * (div * rVc);
    }
    if (!(g.bounded_y && j == 0)) {   // y momentum at face j (walls in y: the wall face is never updated)
        const double div = (Ax * T12(1, 0) - Ax * t12_00) + (Ay * T22(0) - Ay * T22(-1)) + (Az * T23(0, 1) - Az * t23_00);
        Gv[n] -= scale * (div * rVc);
    }
    if (k >= 1) {   // z momentum at the interior face k (between centres k-1 and k)
        const double dzf = g.dzf[k];
        const double Axf = dy * dzf, Ayf = dx * dzf;
        const double div = (Axf * T13(1, 0) - Axf * t13_00) + (g.flat_y ? 0.0 : Ayf * T23(1, 0) - Ayf * t23_00) + (Az * T33(0) - Az * T33(-1));
        Gw[n] -= scale * (div * (g.rdx * (1.0 / dy) * g.rdzf[k]));
    }
    // scalars: J = rho x (-(nu / Pr) grad c) on the six faces of the cell
    const double rPr = 1.0 / F.Pr;
    const double kc = NU(0, 0, 0) * rPr;
    const double kxm = (NU(-1, 0, 0) * rPr + kc) / 2, kxp = (kc + NU(1, 0, 0) * rPr) / 2;
    const double kym = (NU(0, -1, 0) * rPr + kc) / 2, kyp = (kc + NU(0, 1, 0) * rPr) / 2;
    const double kzm = (NU(0, 0, -1) * rPr + kc) / 2, kzp = (kc + NU(0, 0, 1) * rPr) / 2;
    const double rfm = g.rho_f[k], rfp = g.rho_f[k + 1];
    const double rdzfm = g.rdzf[k], rdzfp = g.rdzf[k + 1];
    auto scalar = [&](const double *c, double *G) {
        const double c0 = c[n];
        const double Jxm = rho * (-kxm * ((c0 - c[n - sx]) * g.rdx)), Jxp = rho * (-kxp * ((c[n + sx] - c0) * g.rdx));
        const double Jym = rho * (-kym * ((c0 - c[n - sy]) * g.rdy)), Jyp = rho * (-kyp * ((c[n + sy] - c0) * g.rdy));
        const double Jzm = (k == 0) ? 0.0 : rfm * (-kzm * ((c0 - c[n - sz]) * rdzfm));
        const double Jzp = (k == g.Nz - 1) ? 0.0 : rfp * (-kzp * ((c[n + sz] - c0) * rdzfp));
        const double div = (Ax * Jxp - Ax * Jxm) + (Ay * Jyp - Ay * Jym) + (Az * Jzp - Az * Jzm);
        G[n] -= scale * (div * rVc);
    };
    scalar(F.theta, Gth);
    scalar(F.q, Gq);
}

// ---------------------------------------------------------------------------------------------------------------------------------
// The same five divergences as a z-marching, LDS-tiled kernel (round 5).  k_closure_tendencies is one thread per
// cell: ~65 global loads with their 64-bit addresses and a wave that lives for one cell — 1.23 ms for 8.4 M cells (BOMEX 256 x 256 x 128,
// Float64: 0.11 of the roof in compulsory bytes), latency- and request-bound, not arithmetic-bound (~640 instructions per cell).  Here a
// workgroup owns a 64 x TY tile of columns and walks a chunk of levels; u, v, w, nu_e live in LDS for the levels k-1, k, k+1 (tiles with a
// one-cell frame, three rotating slots), theta and q for level k; every neighbour the stresses and fluxes touch is a ds_read with an
// immediate offset, and a level costs each thread six own-column loads (requested one level ahead), at most two frame cells and the five
// read-modify-writes.  Same expressions, same order of operations as k_closure_tendencies (which stays for walls, Flat y and ragged grids).
// grid (Nx / 64, Ny / TY, ceil(Nz / kchunk)), block (64, TY); Nx % 64 == 0, Ny % TY == 0.
#ifndef CL_KO
#define CL_KO 0      // timing experiments only (results WRONG): 1 no read-modify-write of the G arrays; 2 no frame staging; 4 one barrier per level
#endif
#ifndef CL_NCH
#define CL_NCH 2048
#endif
// Workgroup barrier that orders LDS traffic only (the idiom of k_tridiag_coop, bz_poisson.hip): __syncthreads() is a fence over every
// address space, so hipcc drains the level's five read-modify-write stores (s_waitcnt vmcnt(0)) in front of each of the two barriers of
// a level; the global traffic of this kernel is each thread's own column, nothing another thread reads.
__device__ __forceinline__ void closure_lds_barrier()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

template <int TY>
__global__ __launch_bounds__(64 * TY, 4) void k_closure_march(DevGrid g, ClosureFields F, double *__restrict__ Gu, double *__restrict__ Gv,
                                                          double *__restrict__ Gw, double *__restrict__ Gth, double *__restrict__ Gq,
                                                          double scale, int kchunk)
{
    constexpr int TR = TY + 2, TC = 68, NT = 64 * TY;
    constexpr int NF = 2 * 66 + 2 * TY;                  // frame cells of one tile
    constexpr int NFL = 6 * NF;                          // frame loads per level: u, v, w, nu of level k+2 and theta, q of level k+1
    constexpr int HPT = (NFL + NT - 1) / NT;
    constexpr int TS = TR * TC;                          // one tile
    __shared__ double S[14 * TS];                        // u, v, w, nu: three level slots each; theta, q: one
    double (*U)[TR][TC] = (double (*)[TR][TC])(S), (*V)[TR][TC] = (double (*)[TR][TC])(S + 3 * TS), (*W)[TR][TC] = (double (*)[TR][TC])(S + 6 * TS),
           (*NUt)[TR][TC] = (double (*)[TR][TC])(S + 9 * TS);
    double (*TH)[TC] = (double (*)[TC])(S + 12 * TS), (*Q)[TC] = (double (*)[TC])(S + 13 * TS);
    int bx, by, bz;
    xcd_block(bx, by, bz);
    const int tx = threadIdx.x, ty = threadIdx.y, t = ty * 64 + tx;
    const int i0 = bx * 64, j0 = by * TY, i = i0 + tx, j = j0 + ty;
    const int kbeg = bz * kchunk, kend = min(kbeg + kchunk, g.Nz);
    if (kbeg >= kend) return;
    const long long sy = g.Sx, sz = g.Sxy;
    const double *__restrict__ u = F.u, *__restrict__ v = F.v, *__restrict__ w = F.w, *__restrict__ nu = F.nu;
    const double *__restrict__ th = F.theta, *__restrict__ q = F.q;
    const int r = ty + 1, c = tx + 1;
    // level l of a z-face field / centre field sits in slot (l - kbeg + 1) % 3; nu beyond the walls mirrors the wall level (no-flux)
    auto nu_level = [&](int l) { return min(max(l, 0), g.Nz - 1); };
    // nu_e carries no halos: its x / y neighbours wrap (y-slab contexts: rows -1 and Ny were computed by the extended viscosity launch)
    auto nu_col = [&](int ii) { return ii < 0 ? ii + g.Nx : ii >= g.Nx ? ii - g.Nx : ii; };
    auto nu_row = [&](int jj) { return !g.wrap_y ? jj : jj < 0 ? jj + g.Ny : jj >= g.Ny ? jj - g.Ny : jj; };
    const long long n0 = g.idx(i, j, 0);                 // own column, level 0
    // frame cells of this thread: (field, tile row, tile column) for q = 0 .. HPT-1
    int hf[HPT], hoff[HPT];                              // field; offset of the tile cell in S without the slot term
    long long hn[HPT];                                   // index at level 0 (nu: wrapped column / row)
    bool hok[HPT];
#pragma unroll
    for (int qq = 0; qq < HPT; ++qq) {
        const int sidx = t + qq * NT;
        hok[qq] = sidx < NFL;
        const int f = hok[qq] ? sidx / NF : 0, cell = hok[qq] ? sidx - f * NF : 0;
        int rr, cc;
        if (cell < 132) { rr = (cell < 66) ? 0 : TR - 1; cc = (cell < 66) ? cell : cell - 66; }
        else { const int m = cell - 132; rr = 1 + (m >> 1); cc = (m & 1) ? 65 : 0; }
        hf[qq] = f; hoff[qq] = (f < 4 ? 3 * f : 8 + f) * TS + rr * TC + cc;
        const int ii = i0 - 1 + cc, jj = j0 - 1 + rr;
        hn[qq] = (f == 3) ? g.idx(nu_col(ii), nu_row(jj), 0) : g.idx(ii, jj, 0);
    }
    auto frame_src = [&](int f) -> const double * { return f == 0 ? u : f == 1 ? v : f == 2 ? w : f == 3 ? nu : f == 4 ? th : q; };
    // level a frame load of field f fetches while the compute of level k runs: k + 2 (u, v, w, nu), k + 1 (theta, q)
    auto frame_load = [&](int qq, int k) -> double {
        const int f = hf[qq];
        const int l = (f < 4) ? k + 2 : k + 1;
        return frame_src(f)[hn[qq] + (long long)(f == 3 ? nu_level(l) : l) * sz];
    };
    auto frame_store = [&](int qq, int k, double val) {
        const int sl = (k + 2 - kbeg + 1) % 3;
        S[hoff[qq] + (hf[qq] < 4 ? sl * TS : 0)] = val;
    };
    // ---- prologue: levels kbeg-1, kbeg, kbeg+1 of u, v, w, nu and level kbeg of theta, q ----
    for (int l = kbeg - 1; l <= kbeg + 1; ++l) {
        const int sl = (l - kbeg + 1) % 3;
        const long long n = n0 + (long long)l * sz;
        U[sl][r][c] = u[n]; V[sl][r][c] = v[n]; W[sl][r][c] = w[n];
        NUt[sl][r][c] = nu[n0 + (long long)nu_level(l) * sz];
#pragma unroll
        for (int qq = 0; qq < HPT; ++qq)
            if (hok[qq] && hf[qq] < 4) frame_store(qq, l - 2, frame_load(qq, l - 2));
    }
    double th_m = th[n0 + (long long)(kbeg - 1) * sz], th_c = th[n0 + (long long)kbeg * sz], th_p = th[n0 + (long long)(kbeg + 1) * sz];
    double q_m = q[n0 + (long long)(kbeg - 1) * sz], q_c = q[n0 + (long long)kbeg * sz], q_p = q[n0 + (long long)(kbeg + 1) * sz];
    TH[r][c] = th_c; Q[r][c] = q_c;
#pragma unroll
    for (int qq = 0; qq < HPT; ++qq)
        if (hok[qq] && hf[qq] >= 4) frame_store(qq, kbeg - 1, frame_load(qq, kbeg - 1));
    __syncthreads();

    const double dx = g.dx, dy = g.dy;
    const double rPr = 1.0 / F.Pr;
    long long n = n0 + (long long)kbeg * sz;
    for (int k = kbeg; k < kend; ++k, n += sz) {
        const int sm = (k - kbeg) % 3, sc = (k - kbeg + 1) % 3, sp = (k - kbeg + 2) % 3;      // slots of k-1, k, k+1
        // ---- requests for the next iteration (consumed after the arithmetic) ----
        const long long n2 = n + 2 * sz;
        const double pu = u[n2], pv = v[n2], pw = w[n2], pnu = nu[n0 + (long long)nu_level(k + 2) * sz];
        const double pth = th[n2], pq = q[n2];
        double hnext[HPT];
#pragma unroll
        for (int qq = 0; qq < HPT; ++qq) hnext[qq] = (hok[qq] && !(CL_KO & 2)) ? frame_load(qq, k) : 0.0;

        const double dz = g.dzc[k];
        const double rVc = g.rdx * (1.0 / dy) * g.rdzc[k];
        const double rho = g.rho[k];
        const double Ax = dy * dz, Ay = dx * dz, Az = dx * dy;
        auto Uc = [&](int di, int dj, int dk) { return U[dk < 0 ? sm : dk > 0 ? sp : sc][r + dj][c + di]; };
        auto Vc = [&](int di, int dj, int dk) { return V[dk < 0 ? sm : dk > 0 ? sp : sc][r + dj][c + di]; };
        auto Wc = [&](int di, int dj, int dk) { return W[dk < 0 ? sm : dk > 0 ? sp : sc][r + dj][c + di]; };
        auto NU = [&](int di, int dj, int dk) { return NUt[dk < 0 ? sm : dk > 0 ? sp : sc][r + dj][c + di]; };
        // strains at the offsets the stresses of this cell need (S11 .. S23 of this file, on the tiles)
        auto s11 = [&](int di) { return (Uc(di + 1, 0, 0) - Uc(di, 0, 0)) * g.rdx; };
        auto s22 = [&](int dj) { return (Vc(0, dj + 1, 0) - Vc(0, dj, 0)) * g.rdy; };
        auto s33 = [&](int dk) { return (Wc(0, 0, dk + 1) - Wc(0, 0, dk)) * g.rdzc[k + dk]; };
        auto s12 = [&](int di, int dj) { return ((Uc(di, dj, 0) - Uc(di, dj - 1, 0)) * g.rdy + (Vc(di, dj, 0) - Vc(di - 1, dj, 0)) * g.rdx) * 0.5; };
        auto s13 = [&](int di, int dk) {
            const int kk = k + dk;
            if (kk <= 0 || kk >= g.Nz) return 0.0;
            return ((Uc(di, 0, dk) - Uc(di, 0, dk - 1)) * g.rdzf[kk] + (Wc(di, 0, dk) - Wc(di - 1, 0, dk)) * g.rdx) * 0.5;
        };
        auto s23 = [&](int dj, int dk) {
            const int kk = k + dk;
            if (kk <= 0 || kk >= g.Nz) return 0.0;
            return ((Vc(0, dj, dk) - Vc(0, dj, dk - 1)) * g.rdzf[kk] + (Wc(0, dj, dk) - Wc(0, dj - 1, dk)) * g.rdy) * 0.5;
        };
        auto nu_ffc = [&](int di, int dj) { return ((NU(di - 1, dj - 1, 0) + NU(di, dj - 1, 0)) / 2 + (NU(di - 1, dj, 0) + NU(di, dj, 0)) / 2) / 2; };
        auto nu_fcf = [&](int di, int dk) { return ((NU(di - 1, 0, dk - 1) + NU(di, 0, dk - 1)) / 2 + (NU(di - 1, 0, dk) + NU(di, 0, dk)) / 2) / 2; };
        auto nu_cff = [&](int dj, int dk) { return ((NU(0, dj - 1, dk - 1) + NU(0, dj, dk - 1)) / 2 + (NU(0, dj - 1, dk) + NU(0, dj, dk)) / 2) / 2; };
        auto T11 = [&](int di) { return rho * (-2 * NU(di, 0, 0) * s11(di)); };
        auto T22 = [&](int dj) { return rho * (-2 * NU(0, dj, 0) * s22(dj)); };
        auto T33 = [&](int dk) { return g.rho[k + dk] * (-2 * NU(0, 0, dk) * s33(dk)); };
        auto T12 = [&](int di, int dj) { return rho * (-2 * nu_ffc(di, dj) * s12(di, dj)); };
        // (dk = 1 needs nu of level k+1 and k: nu_fcf / nu_cff with dk - 1 = 0, dk = 1; dk = 0 needs k-1 and k)
        auto T13 = [&](int di, int dk) { return g.rho_f[k + dk] * (-2 * nu_fcf(di, dk) * s13(di, dk)); };
        auto T23 = [&](int dj, int dk) { return g.rho_f[k + dk] * (-2 * nu_cff(dj, dk) * s23(dj, dk)); };

        const double t12_00 = T12(0, 0), t13_00 = T13(0, 0), t23_00 = T23(0, 0);
        {
            const double div = (Ax * T11(0) - Ax * T11(-1)) + (Ay * T12(0, 1) - Ay * t12_00) + (Az * T13(0, 1) - Az * t13_00);
            if (!(CL_KO & 1)) Gu[n] -= scale * (div * rVc); else if (div == 1.2345) Gu[n] = 0.0;
        }
        {
            const double div = (Ax * T12(1, 0) - Ax * t12_00) + (Ay * T22(0) - Ay * T22(-1)) + (Az * T23(0, 1) - Az * t23_00);
            if (!(CL_KO & 1)) Gv[n] -= scale * (div * rVc); else if (div == 1.2345) Gv[n] = 0.0;
        }
        if (k >= 1) {
            const double dzf = g.dzf[k];
            const double Axf = dy * dzf, Ayf = dx * dzf;
            const double div = (Axf * T13(1, 0) - Axf * t13_00) + (Ayf * T23(1, 0) - Ayf * t23_00) + (Az * T33(0) - Az * T33(-1));
            if (!(CL_KO & 1)) Gw[n] -= scale * (div * (g.rdx * (1.0 / dy) * g.rdzf[k])); else if (div == 1.2345) Gw[n] = 0.0;
        }
        const double kc = NU(0, 0, 0) * rPr;
        const double kxm = (NU(-1, 0, 0) * rPr + kc) / 2, kxp = (kc + NU(1, 0, 0) * rPr) / 2;
        const double kym = (NU(0, -1, 0) * rPr + kc) / 2, kyp = (kc + NU(0, 1, 0) * rPr) / 2;
        const double kzm = (NU(0, 0, -1) * rPr + kc) / 2, kzp = (kc + NU(0, 0, 1) * rPr) / 2;
        const double rfm = g.rho_f[k], rfp = g.rho_f[k + 1];
        const double rdzfm = g.rdzf[k], rdzfp = g.rdzf[k + 1];
        auto scalar = [&](const double (*Tl)[TC], double c0, double cm, double cp, double *G) {
            const double Jxm = rho * (-kxm * ((c0 - Tl[r][c - 1]) * g.rdx)), Jxp = rho * (-kxp * ((Tl[r][c + 1] - c0) * g.rdx));
            const double Jym = rho * (-kym * ((c0 - Tl[r - 1][c]) * g.rdy)), Jyp = rho * (-kyp * ((Tl[r + 1][c] - c0) * g.rdy));
            const double Jzm = (k == 0) ? 0.0 : rfm * (-kzm * ((c0 - cm) * rdzfm));
            const double Jzp = (k == g.Nz - 1) ? 0.0 : rfp * (-kzp * ((cp - c0) * rdzfp));
            const double div = (Ax * Jxp - Ax * Jxm) + (Ay * Jyp - Ay * Jym) + (Az * Jzp - Az * Jzm);
            if (!(CL_KO & 1)) G[n] -= scale * (div * rVc); else if (div == 1.2345) G[n] = 0.0;
        };
        scalar(TH, th_c, th_m, th_p, Gth);
        scalar(Q, q_c, q_m, q_p, Gq);
        if (!(CL_KO & 4)) closure_lds_barrier();                           // every wave is done with slot k-1 and with the theta / q tile of level k
        // ---- stage level k+2 into the slot level k-1 leaves, and theta / q of level k+1 ----
        U[sm][r][c] = pu; V[sm][r][c] = pv; W[sm][r][c] = pw; NUt[sm][r][c] = pnu;
        TH[r][c] = th_p; Q[r][c] = q_p;
#pragma unroll
        for (int qq = 0; qq < HPT; ++qq)
            if (hok[qq]) frame_store(qq, k, hnext[qq]);
        th_m = th_c; th_c = th_p; th_p = pth;
        q_m = q_c; q_c = q_p; q_p = pq;
        closure_lds_barrier();
    }
}

// The eddy viscosity the same way (round 5): u, v, w tiles of the levels k-1, k, k+1 in LDS (one-cell frame, three rotating slots), the
// logarithms of theta_v of the own column in a ring of three — one per level instead of three.  Same expressions as
// k_smagorinsky_viscosity, which stays for walls, Flat y, ragged grids and the two rows beyond a y-slab.
template <int TY>
__global__ __launch_bounds__(64 * TY, 4) void k_smagorinsky_march(DevGrid g, ClosureFields F, const double *__restrict__ T,
                                                                  const double *__restrict__ qv, const double *__restrict__ ipi,
                                                                  const double *__restrict__ delta2, double *__restrict__ nu, int kchunk)
{
    constexpr int TR = TY + 2, TC = 68, NT = 64 * TY, TS = TR * TC;
    constexpr int NF = 2 * 66 + 2 * TY, NFL = 3 * NF, HPT = (NFL + NT - 1) / NT;
    __shared__ double S[9 * TS];
    double (*U)[TR][TC] = (double (*)[TR][TC])(S), (*V)[TR][TC] = (double (*)[TR][TC])(S + 3 * TS), (*W)[TR][TC] = (double (*)[TR][TC])(S + 6 * TS);
    int bx, by, bz;
    xcd_block(bx, by, bz);
    const int tx = threadIdx.x, ty = threadIdx.y, t = ty * 64 + tx;
    const int i0 = bx * 64, j0 = by * TY, i = i0 + tx, j = j0 + ty;
    const int kbeg = bz * kchunk, kend = min(kbeg + kchunk, g.Nz);
    if (kbeg >= kend) return;
    const long long sz = g.Sxy;
    const double *__restrict__ u = F.u, *__restrict__ v = F.v, *__restrict__ w = F.w;
    const int r = ty + 1, c = tx + 1;
    const long long n0 = g.idx(i, j, 0);
    int hf[HPT], hoff[HPT];
    long long hn[HPT];
    bool hok[HPT];
#pragma unroll
    for (int qq = 0; qq < HPT; ++qq) {
        const int sidx = t + qq * NT;
        hok[qq] = sidx < NFL;
        const int f = hok[qq] ? sidx / NF : 0, cell = hok[qq] ? sidx - f * NF : 0;
        int rr, cc;
        if (cell < 132) { rr = (cell < 66) ? 0 : TR - 1; cc = (cell < 66) ? cell : cell - 66; }
        else { const int m = cell - 132; rr = 1 + (m >> 1); cc = (m & 1) ? 65 : 0; }
        hf[qq] = f; hoff[qq] = 3 * f * TS + rr * TC + cc;
        hn[qq] = g.idx(i0 - 1 + cc, j0 - 1 + rr, 0);
    }
    auto frame_src = [&](int f) -> const double * { return f == 0 ? u : f == 1 ? v : w; };
    for (int l = kbeg - 1; l <= kbeg + 1; ++l) {
        const int sl = (l - kbeg + 1) % 3;
        const long long n = n0 + (long long)l * sz;
        U[sl][r][c] = u[n]; V[sl][r][c] = v[n]; W[sl][r][c] = w[n];
#pragma unroll
        for (int qq = 0; qq < HPT; ++qq)
            if (hok[qq]) S[hoff[qq] + sl * TS] = frame_src(hf[qq])[hn[qq] + (long long)l * sz];
    }
    long long n = n0 + (long long)kbeg * sz;
    double lm = log_theta_v(g, ipi, T, qv, n - sz, kbeg - 1), lc = log_theta_v(g, ipi, T, qv, n, kbeg);
    closure_lds_barrier();
    for (int k = kbeg; k < kend; ++k, n += sz) {
        const int sm = (k - kbeg) % 3, sc = (k - kbeg + 1) % 3, sp = (k - kbeg + 2) % 3;
        const long long n2 = n + 2 * sz;
        const double pu = u[n2], pv = v[n2], pw = w[n2];
        double hnext[HPT];
#pragma unroll
        for (int qq = 0; qq < HPT; ++qq) hnext[qq] = hok[qq] ? frame_src(hf[qq])[hn[qq] + (long long)(k + 2) * sz] : 0.0;
        const double Tp = T[n + sz], qp = qv[n + sz];
        auto Uc = [&](int di, int dj, int dk) { return U[dk < 0 ? sm : dk > 0 ? sp : sc][r + dj][c + di]; };
        auto Vc = [&](int di, int dj, int dk) { return V[dk < 0 ? sm : dk > 0 ? sp : sc][r + dj][c + di]; };
        auto Wc = [&](int di, int dj, int dk) { return W[dk < 0 ? sm : dk > 0 ? sp : sc][r + dj][c + di]; };
        auto sq = [](double a) { return a * a; };
        // strains at (cell + (di, dj), level k + dk): S11 .. S23 of this file on the tiles
        auto s12 = [&](int di, int dj) { return ((Uc(di, dj, 0) - Uc(di, dj - 1, 0)) * g.rdy + (Vc(di, dj, 0) - Vc(di - 1, dj, 0)) * g.rdx) * 0.5; };
        auto s13 = [&](int di, int dk) {
            const int kk = k + dk;
            if (kk <= 0 || kk >= g.Nz) return 0.0;
            return ((Uc(di, 0, dk) - Uc(di, 0, dk - 1)) * g.rdzf[kk] + (Wc(di, 0, dk) - Wc(di - 1, 0, dk)) * g.rdx) * 0.5;
        };
        auto s23 = [&](int dj, int dk) {
            const int kk = k + dk;
            if (kk <= 0 || kk >= g.Nz) return 0.0;
            return ((Vc(0, dj, dk) - Vc(0, dj, dk - 1)) * g.rdzf[kk] + (Wc(0, dj, dk) - Wc(0, dj - 1, dk)) * g.rdy) * 0.5;
        };
        const double s11 = (Uc(1, 0, 0) - Uc(0, 0, 0)) * g.rdx, s22 = (Vc(0, 1, 0) - Vc(0, 0, 0)) * g.rdy, s33 = (Wc(0, 0, 1) - Wc(0, 0, 0)) * g.rdzc[k];
        const double a12 = ((sq(s12(0, 0)) + sq(s12(1, 0))) / 2 + (sq(s12(0, 1)) + sq(s12(1, 1))) / 2) / 2;
        const double a13 = ((sq(s13(0, 0)) + sq(s13(1, 0))) / 2 + (sq(s13(0, 1)) + sq(s13(1, 1))) / 2) / 2;
        const double a23 = ((sq(s23(0, 0)) + sq(s23(1, 0))) / 2 + (sq(s23(0, 1)) + sq(s23(1, 1))) / 2) / 2;
        const double Sig2 = (s11 * s11 + s22 * s22 + s33 * s33) + 2 * a12 + 2 * a13 + 2 * a23;
        const double Rm = (1.0 - (qp + 0.0 + 0.0)) * g.Rd + qp * g.Rv;
        const double lp = log(Rm / g.Rd * Tp * ipi[k + 1]);
        const double bdn = g.g * ((lc - lm) * g.rdzf[k]), bup = g.g * ((lp - lc) * g.rdzf[k + 1]);
        const double N2 = (bdn + bup) / 2;
        const double N2p = fmax(0.0, N2);
        const double stab = (Sig2 == 0.0) ? 0.0 : sqrt(1.0 - fmin(1.0, F.Cb * N2p / Sig2));
        nu[n] = (stab * F.C2) * delta2[k] * sqrt(2 * Sig2);
        lm = lc; lc = lp;
        closure_lds_barrier();
        U[sm][r][c] = pu; V[sm][r][c] = pv; W[sm][r][c] = pw;
#pragma unroll
        for (int qq = 0; qq < HPT; ++qq)
            if (hok[qq]) S[hoff[qq] + sm * TS] = hnext[qq];
        closure_lds_barrier();
    }
}

// - div J^c of one more scalar (user tracers): the `scalar` part of k_closure_tendencies for a field of its own
__global__ __launch_bounds__(256) void k_closure_scalar(DevGrid g, const double *__restrict__ nu, double rPr, const double *__restrict__ c,
                                                        double *__restrict__ G, double scale)
{
    int bx, by, bz;
    xcd_block(bx, by, bz);
    const int i = bx * 256 + threadIdx.x, j = by, k = bz;
    if (i >= g.Nx) return;
    const long long n = g.idx(i, j, k), sy = g.Sx, sz = g.Sxy;
    const long long oxm = (i == 0 && !g.bounded_x) ? (long long)(g.Nx - 1) : -1, oxp = (i == g.Nx - 1 && !g.bounded_x) ? -(long long)(g.Nx - 1) : 1;
    const long long oym = ((j == 0 && g.wrap_y) ? (long long)(g.Ny - 1) : -1) * sy, oyp = ((j == g.Ny - 1 && g.wrap_y) ? -(long long)(g.Ny - 1) : 1) * sy;
    const long long ozm = (k == 0) ? 0 : -sz, ozp = (k == g.Nz - 1) ? 0 : sz;
    const double dx = g.dx, dy = g.dy, dz = g.dzc[k];
    const double rVc = g.rdx * (1.0 / dy) * g.rdzc[k], rho = g.rho[k];
    const double Ax = dy * dz, Ay = dx * dz, Az = dx * dy;
    const double kc = nu[n] * rPr;
    const double kxm = (nu[n + oxm] * rPr + kc) / 2, kxp = (kc + nu[n + oxp] * rPr) / 2;
    const double kym = (nu[n + oym] * rPr + kc) / 2, kyp = (kc + nu[n + oyp] * rPr) / 2;
    const double kzm = (nu[n + ozm] * rPr + kc) / 2, kzp = (kc + nu[n + ozp] * rPr) / 2;
    const double c0 = c[n];
    const double Jxm = rho * (-kxm * ((c0 - c[n - 1]) * g.rdx)), Jxp = rho * (-kxp * ((c[n + 1] - c0) * g.rdx));
    const double Jym = rho * (-kym * ((c0 - c[n - sy]) * g.rdy)), Jyp = rho * (-kyp * ((c[n + sy] - c0) * g.rdy));
    const double Jzm = (k == 0) ? 0.0 : g.rho_f[k] * (-kzm * ((c0 - c[n - sz]) * g.rdzf[k]));
    const double Jzp = (k == g.Nz - 1) ? 0.0 : g.rho_f[k + 1] * (-kzp * ((c[n + sz] - c0) * g.rdzf[k + 1]));
    const double div = (Ax * Jxp - Ax * Jxm) + (Ay * Jyp - Ay * Jym) + (Az * Jzp - Az * Jzm);
    G[n] -= scale * (div * rVc);
}

extern "C" int bz_set_closure(bz_ctx *ctx, const bz_smagorinsky_lilly *closure, double *eddy_viscosity)
{
    if (ctx) ++ctx->config_epoch;      // captured steps (bz_graph.hip) belong to one configuration
    if (!ctx) return BZ_ERR_INVALID;
    if (!closure) { ctx->has_closure = false; ctx->closure_nu = nullptr; return BZ_OK; }
    if (!eddy_viscosity) return BZ_ERR_INVALID;
    if (ctx->compressible || ctx->dg.formulation != 0 || ctx->dg.microphysics == 2) {      // y-slab contexts: through the library-owned distributed step (bz_comm.hip)
        ctx->last_error = "bz_set_closure: SmagorinskyLilly is implemented for the anelastic "
                          "potential-temperature model (microphysics nothing or SaturationAdjustment)";
        return BZ_ERR_UNSUPPORTED;
    }
    if (ctx->dg.Hx < 1 || (ctx->dg.Hy < 1 && !ctx->dg.flat_y) || ctx->dg.Hz < 1) { ctx->last_error = "bz_set_closure: needs halos >= 1"; return BZ_ERR_UNSUPPORTED; }
    ctx->closure = *closure;
    ctx->closure_nu = eddy_viscosity;
    if (!ctx->d_closure_ipi) BZ_HIP(hipMalloc(&ctx->d_closure_ipi, (size_t)2 * (ctx->dg.Nz + 2) * sizeof(double)));      // ipi, delta2: Nz + 2 entries each
    hipLaunchKernelGGL(k_inverse_exner_column, dim3(1), dim3(256), 0, ctx->stream, ctx->dg, ctx->d_closure_ipi + 1, ctx->d_closure_ipi + (ctx->dg.Nz + 2) + 1);
    BZ_LAUNCH_CHECK();
    ctx->has_closure = true;
    return BZ_OK;
}

static ClosureFields closure_fields(bz_ctx *ctx, const bz_state *s)
{
    ClosureFields F;
    F.u = s->u; F.v = s->v; F.w = s->w; F.nu = ctx->closure_nu; F.theta = s->theta; F.q = s->q;
    F.C2 = ctx->closure.smagorinsky_coefficient * ctx->closure.smagorinsky_coefficient;
    F.Cb = ctx->closure.reduction_factor;
    F.Pr = ctx->closure.prandtl_number;
    F.jofs = 0;
    return F;
}

// compute_closure_fields!(model.closure_fields, model.closure, model): nu_e on the interior
extern "C" int bz_compute_closure_fields(bz_ctx *ctx, const bz_state *s)
{
    if (!ctx || !s) return BZ_ERR_INVALID;
    if (!ctx->has_closure) return BZ_OK;
    { const int rcs = bzi_refresh_diagnostics(ctx, s, "bz_compute_closure_fields"); if (rcs) return rcs; }
    const DevGrid &g = ctx->dg;
    ProfileScope ps(ctx, "smagorinsky_viscosity");
    const double *qv = (g.microphysics == 1) ? g.qv_field : s->q;      // specific_humidity(model)
    ClosureFields F = closure_fields(ctx, s);
    // y-slabs: one more row on each side (u, v, w, T, q^v carry Hy >= 2 exchanged rows there), so nu_e needs no exchange of its own
    F.jofs = ctx->slab_mode ? -1 : 0;
    constexpr int CTY = 8;
    const bool march = !ctx->tune.no_closure_march && !g.flat_y && !g.bounded_x && !g.bounded_y && g.Nx % 64 == 0 && g.Ny % CTY == 0 && g.Nz >= 4 &&
                       g.Hx >= 1 && g.Hy >= 1 && g.Hz >= 2;
    const double *ipi = ctx->d_closure_ipi + 1, *delta2 = ctx->d_closure_ipi + (g.Nz + 2) + 1;
    if (march) {
        const long long tiles = (long long)(g.Nx / 64) * (g.Ny / CTY);
        int nch = (int)((CL_NCH + tiles - 1) / tiles);
        if (nch > g.Nz / 16) nch = g.Nz / 16;
        if (nch < 1) nch = 1;
        const int kc = (g.Nz + nch - 1) / nch;
        F.jofs = 0;
        hipLaunchKernelGGL(k_smagorinsky_march<CTY>, dim3(g.Nx / 64, g.Ny / CTY, (g.Nz + kc - 1) / kc), dim3(64, CTY), 0, ctx->stream, g, F,
                           s->T, qv, ipi, delta2, ctx->closure_nu, kc);
        if (ctx->slab_mode) {      // the rows beyond the slab (-1 and Ny), which the tendency kernels read in place of the periodic wrap
            for (int side = 0; side < 2; ++side) {
                F.jofs = side ? g.Ny : -1;
                hipLaunchKernelGGL(k_smagorinsky_viscosity, dim3((g.Nx + 255) / 256, 1, (g.Nz + SMAG_KCHUNK - 1) / SMAG_KCHUNK), dim3(256), 0, ctx->stream,
                                   g, F, s->T, qv, ipi, delta2, ctx->closure_nu);
            }
        }
    } else
    hipLaunchKernelGGL(k_smagorinsky_viscosity, dim3((g.Nx + 255) / 256, g.Ny + (ctx->slab_mode ? 2 : 0), (g.Nz + SMAG_KCHUNK - 1) / SMAG_KCHUNK),
                       dim3(256), 0, ctx->stream, g, F, s->T, qv, ipi, delta2, ctx->closure_nu);
    BZ_LAUNCH_CHECK();
    // walls in y (or x): nu_e is a centre field with the default no-flux condition — its first halo rows mirror the wall rows (the tendency
    // kernels reach rows -1 and Ny through the halo, as on y-slabs, where the extended launch above computes them)
    if (g.bounded_y || g.bounded_x) return bzi_fill_halo(ctx, ctx->closure_nu, 0);
    return BZ_OK;
}

// closure terms of compute_tendencies!; scale = 1 subtracts the divergences from tendencies, the whole-step seam passes
// alpha dt and the arrays its fused RK update just wrote (see bzi_apply_forcings)
void bzi_closure_teardown(bz_ctx *ctx)
{
    if (ctx->d_closure_ipi) hipFree(ctx->d_closure_ipi);
    ctx->d_closure_ipi = nullptr;
}

int bzi_apply_closure(bz_ctx *ctx, const bz_state *s, double *Gu, double *Gv, double *Gw, double *Gth, double *Gq, double scale)
{
    int rc = bz_compute_closure_fields(ctx, s);
    if (rc) return rc;
    const DevGrid &g = ctx->dg;
    ProfileScope ps(ctx, "closure_tendencies");
    // the z-marching LDS-tiled kernel on whole tiles of a periodic (or y-slab) grid; walls, Flat y and ragged grids keep the cell-per-thread one
    constexpr int CTY = 8;
    const bool march = !ctx->tune.no_closure_march && !g.flat_y && !g.bounded_x && !g.bounded_y && g.Nx % 64 == 0 && g.Ny % CTY == 0 && g.Nz >= 4 &&
                       g.Hx >= 1 && g.Hy >= 1 && g.Hz >= 2;
    if (march) {
        // chunks of >= 16 levels, enough workgroups to fill the chip (every chunk restages three levels)
        const long long tiles = (long long)(g.Nx / 64) * (g.Ny / CTY);
        int nch = (int)((CL_NCH + tiles - 1) / tiles);
        if (nch > g.Nz / 16) nch = g.Nz / 16;
        if (nch < 1) nch = 1;
        const int kc = (g.Nz + nch - 1) / nch;
        hipLaunchKernelGGL(k_closure_march<CTY>, dim3(g.Nx / 64, g.Ny / CTY, (g.Nz + kc - 1) / kc), dim3(64, CTY), 0, ctx->stream, g,
                           closure_fields(ctx, s), Gu, Gv, Gw, Gth, Gq, scale, kc);
    } else
    hipLaunchKernelGGL(k_closure_tendencies, dim3((g.Nx + 255) / 256, g.Ny, g.Nz), dim3(256), 0, ctx->stream, g,
                       closure_fields(ctx, s), Gu, Gv, Gw, Gth, Gq, scale);
    // user tracers diffuse like every other scalar.  The whole-step seam updates rho c in place (bzi_tracer_rk3), so there the
    // divergence goes to the density array with the stage weight, exactly as for rho theta / rho q; per-operator callers pass scale = 1
    // and the tendency arrays
    const ClosureFields F = closure_fields(ctx, s);
    for (int t = 0; t < ctx->n_tracers; ++t)
        hipLaunchKernelGGL(k_closure_scalar, dim3((g.Nx + 255) / 256, g.Ny, g.Nz), dim3(256), 0, ctx->stream, g, F.nu, 1.0 / F.Pr,
                           (const double *)ctx->tracers[t].specific, (Gth == s->rho_theta) ? ctx->tracers[t].density : ctx->tracers[t].G, scale);
    BZ_LAUNCH_CHECK();
    return BZ_OK;
}
