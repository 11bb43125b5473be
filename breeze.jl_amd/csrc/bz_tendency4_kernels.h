// bz_tendency4_kernels.h — LDS-tiled tendency kernels (fourth generation).
//
// What the measurements said (tools/tendbench, rocprofv3 PMC): the free-running kernels issue
// ~27 vector loads per thread and level, two thirds of them x/y stencil neighbours that other threads of the same
// block already hold in their vertical register rings; those re-reads miss the 32 KB L1, cost L2 requests and
// 1.6-3.3x over-fetch at the L2-fabric boundary, and the kernels sit at 2x their pure-streaming time.
// Here the block stages the current level of the advected fields in LDS:
//   * interior of the tile: written from the threads' register rings (no load at all);
//   * 3-wide halo frame of the tile: ~1 global load per thread and level, issued one level ahead (prefetched into
//     registers before the arithmetic of the current level, stored to the other LDS buffer after it);
//   * x/y stencils are ds_read_b64 from the tile;  y-face fluxes are shared through a second double-buffered LDS
//     array (one barrier per level covers both), x-face fluxes through a wave shuffle + the batched out-of-wave
//     flux of k_tend3.
// Loads per thread and level drop from ~27 to ~9 (fused theta+q kernel), reconstructions from 8 to 6 + 2/TY.
#pragma once
#include "bz_internal.h"
#include "bz_weno.h"

#ifndef BZ4_XCD
#define BZ4_XCD 1
#endif
template <int TY>
__global__ __launch_bounds__(64 * TY) void k_scalar_pair_lds(DevGrid g, const double *__restrict__ u,
                                                            const double *__restrict__ v,
                                                            const double *__restrict__ w,
                                                            const double *__restrict__ ca,
                                                            const double *__restrict__ cb, double *Ga,
                                                            double *Gb, int kchunk, RKEpilogue E,
                                                            const double *pa, const double *pb)
{   // pa, pb: prognostic rho*theta, rho*q (read only when the RK epilogue is on; then Ga = pa, Gb = pb in place)
    constexpr int TR = TY + 6, TC = 72;                 // tile rows, padded row length (70 used)
    constexpr int NHALO = TR * 70 - TY * 64;            // frame cells per field
    constexpr int NT = 64 * TY;
    constexpr int HPT = (NHALO + NT - 1) / NT;          // frame cells per thread (1 for TY = 8, 2 for TY = 4)
    __shared__ double T[2][2][TR][TC];
    __shared__ double FY[2][2][TY + 1][64];

    const int tx = threadIdx.x, ty = threadIdx.y, t = ty * 64 + tx;
    // XCD-contiguous block order where the grid allows it (as bz_block5 of the fifth-generation kernels): consecutive workgroup ids go
    // round-robin to the 8 XCDs, so in launch order the x and y neighbours of a tile sit behind other L2s
    int bx = blockIdx.x, by = blockIdx.y, bz = blockIdx.z;
    {
        const unsigned gx = gridDim.x, gy = gridDim.y, W = gx * gy * gridDim.z;
        if (BZ4_XCD && (W & 7u) == 0) {
            unsigned w = bx + gx * (by + gy * bz);
            w = (w & 7u) * (W >> 3) + (w >> 3);
            bx = (int)(w % gx); by = (int)((w / gx) % gy); bz = (int)(w / (gx * gy));
        }
        bx = __builtin_amdgcn_readfirstlane(bx); by = __builtin_amdgcn_readfirstlane(by); bz = __builtin_amdgcn_readfirstlane(bz);
    }
    const int i0 = bx * 64, j0 = by * TY;
    const int i = i0 + tx, j = j0 + ty;
    // ragged tiles: out-of-range threads still stage true halo values (never clamp into the interior)
    const int ic = min(i, g.Nx + 2), jc = min(j, g.Ny + 2);
    const int nact = min(64, g.Nx - i0);
    const int ie = i0 + nact, le = nact - 1;
    const int kbeg = bz * kchunk, kend = min(kbeg + kchunk, g.Nz);
    if (kbeg >= kend) return;                           // block-uniform
    const long long sz = g.Sxy;
    const bool store = (i < g.Nx) && (j < g.Ny);
    long long n = g.idx(ic, jc, kbeg);

    // frame cells handled by this thread: LDS slot (row, col) and global index at level kbeg
    int hr[HPT], hc[HPT];
    long long hn[HPT];
    bool hok[HPT];
#pragma unroll
    for (int q = 0; q < HPT; ++q) {
        const int h = t + q * NT;
        hok[q] = h < NHALO;
        int r, c;
        if (h < 6 * 70) {                               // three rows below and three above the tile, full width
            const int rr = h / 70;
            c = h - rr * 70;
            r = (rr < 3) ? rr : TY + rr;
        } else {                                        // side columns of the TY interior rows
            const int hh = h - 6 * 70;
            const int rr = hh / 6, cc = hh - rr * 6;
            r = 3 + rr;
            c = (cc < 3) ? cc : 64 + cc;
        }
        if (!hok[q]) { r = 0; c = 0; }
        hr[q] = r; hc[q] = c;
        const int gi = min(i0 - 3 + c, g.Nx + 2), gj = min(j0 - 3 + r, g.Ny + 2);
        hn[q] = g.idx(gi, gj, kbeg);
    }

    // vertical rings
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
    // tile of level kbeg
    T[0][0][ty + 3][tx + 3] = a[3];
    T[0][1][ty + 3][tx + 3] = b[3];
#pragma unroll
    for (int q = 0; q < HPT; ++q)
        if (hok[q]) { T[0][0][hr[q]][hc[q]] = ca[hn[q]]; T[0][1][hr[q]][hc[q]] = cb[hn[q]]; }
    __syncthreads();

    double ea = 0.0, eb = 0.0;
    int buf = 0;
    for (int k = kbeg; k < kend; ++k, n += sz) {
        // ---- loads: next level's frame cells, ring tops, velocities of this level ----
        double ha[HPT], hb[HPT];
        const long long lev = (long long)(k + 1 - kbeg) * sz;
#pragma unroll
        for (int q = 0; q < HPT; ++q) { ha[q] = hok[q] ? ca[hn[q] + lev] : 0.0; hb[q] = hok[q] ? cb[hn[q] + lev] : 0.0; }
        const double ta = ca[n + 3 * sz], tb = cb[n + 3 * sz];
        const double ut = u[n], vt = v[n], wt = w[n + sz];
        const double vtop = (ty == 0) ? v[g.idx(ic, min(j0 + TY, g.Ny), k)] : 0.0;
        if (((k - kbeg) & 63) == 0) {       // out-of-wave x flux for the next 64 levels (lane l <-> level k + l)
            const int kk = min(k + tx, kend - 1);
            const long long ne = g.idx(ie, jc, kk);
            const double ue = u[ne];
            const bool le_ = ue > 0.0;
            const double cf = g.Ax[kk] * ue, rho = g.rho[kk];
            ea = rho * (cf * bz_up5(ca[ne - 3], ca[ne - 2], ca[ne - 1], ca[ne], ca[ne + 1], ca[ne + 2], le_));
            eb = rho * (cf * bz_up5(cb[ne - 3], cb[ne - 2], cb[ne - 1], cb[ne], cb[ne + 1], cb[ne + 2], le_));
        }
        const int src = (k - kbeg) & 63;
        const double rho = g.rho[k];
        const double(*Ta)[TC] = T[buf][0];
        const double(*Tb)[TC] = T[buf][1];

        // ---- z ----
        double fza_hi, fzb_hi;
        {
            const bool left = wt > 0.0;
            const int B = bz_buffer_face(k + 1, g.Nz);
            const double cf = g.Az * wt, rf = g.rho_f[k + 1];
            fza_hi = rf * (cf * bz_upB(a[1], a[2], a[3], a[4], a[5], ta, left, B));
            fzb_hi = rf * (cf * bz_upB(b[1], b[2], b[3], b[4], b[5], tb, left, B));
        }
        // ---- x: own low face from the tile row ----
        double fxa, fxb;
        {
            const bool left = ut > 0.0;
            const double cf = g.Ax[k] * ut;
            const double *ra = Ta[ty + 3] + tx, *rb = Tb[ty + 3] + tx;
            fxa = rho * (cf * bz_up5(ra[0], ra[1], ra[2], a[3], ra[4], ra[5], left));
            fxb = rho * (cf * bz_up5(rb[0], rb[1], rb[2], b[3], rb[4], rb[5], left));
        }
        // ---- y: own low face from the tile column; wave 0 also does the face above the tile ----
        double fya, fyb;
        {
            const bool left = vt > 0.0;
            const double cf = g.Ay[k] * vt;
            const int c = tx + 3;
            fya = rho * (cf * bz_up5(Ta[ty][c], Ta[ty + 1][c], Ta[ty + 2][c], a[3], Ta[ty + 4][c], Ta[ty + 5][c], left));
            fyb = rho * (cf * bz_up5(Tb[ty][c], Tb[ty + 1][c], Tb[ty + 2][c], b[3], Tb[ty + 4][c], Tb[ty + 5][c], left));
            FY[buf][0][ty][tx] = fya;
            FY[buf][1][ty][tx] = fyb;
            if (ty == 0) {
                const bool l2 = vtop > 0.0;
                const double c2 = g.Ay[k] * vtop;
                FY[buf][0][TY][tx] = rho * (c2 * bz_up5(Ta[TY][c], Ta[TY + 1][c], Ta[TY + 2][c], Ta[TY + 3][c], Ta[TY + 4][c], Ta[TY + 5][c], l2));
                FY[buf][1][TY][tx] = rho * (c2 * bz_up5(Tb[TY][c], Tb[TY + 1][c], Tb[TY + 2][c], Tb[TY + 3][c], Tb[TY + 4][c], Tb[TY + 5][c], l2));
            }
        }
        // ---- stage level k+1 in the other buffer ----
        T[buf ^ 1][0][ty + 3][tx + 3] = a[4];
        T[buf ^ 1][1][ty + 3][tx + 3] = b[4];
#pragma unroll
        for (int q = 0; q < HPT; ++q)
            if (hok[q]) { T[buf ^ 1][0][hr[q]][hc[q]] = ha[q]; T[buf ^ 1][1][hr[q]][hc[q]] = hb[q]; }
        __syncthreads();
        // ---- combine ----
        {
            double na = __shfl_down(fxa, 1), nb = __shfl_down(fxb, 1);
            const double xa = __shfl(ea, src), xb = __shfl(eb, src);
            if (tx == le) { na = xa; nb = xb; }
            const double dya = FY[buf][0][ty + 1][tx] - fya;
            const double dyb = FY[buf][1][ty + 1][tx] - fyb;
            const double Vi = g.Vinv_c[k];
            if (store) {
                const double ga = -(Vi * ((na - fxa) + dya + (fza_hi - fza)));
                const double gb = -(Vi * ((nb - fxb) + dyb + (fzb_hi - fzb)));
                if (E.mode == 0) { Ga[n] = ga; Gb[n] = gb; }
                else {
                    Ga[n] = bz_rk_apply(E.mode, E.dt, E.alpha, E.oma, E.u0, E.u0_out, ga, pa[n], n);
                    Gb[n] = bz_rk_apply(E.mode, E.dt, E.alpha, E.oma, E.u0b, E.u0b_out, gb, pb[n], n);
                }
            }
        }
        fza = fza_hi; fzb = fzb_hi;
#pragma unroll
        for (int s = 0; s < 5; ++s) { a[s] = a[s + 1]; b[s] = b[s + 1]; }
        a[5] = ta; b[5] = tb;
        buf ^= 1;
    }
}

// ===================================================================================================
// Momentum kernels, same idea restricted to what the probes showed to matter: x-stencil loads are L1 hits and
// cost nothing measurable, the y-stencil of the advected velocity is what thrashes L1/L2.  So only that y-stencil
// goes through an LDS tile (rows j0-3 .. j0+TY+2 of the block's 64 columns, interior rows from the register
// rings, the six frame rows prefetched one level ahead), y-face fluxes are shared through LDS (wave 0 also
// evaluates the face above the tile), x-face/centre fluxes through a wave shuffle + batched out-of-wave flux.
// ===================================================================================================
#include "bz_tendency3_kernels.h"

template <int TY>
__global__ __launch_bounds__(64 * TY) void k_u_tend_lds(DevGrid g, Tend3Fields F, int kchunk, RKEpilogue E)
{
    constexpr int TR = TY + 6;
    __shared__ double T[2][TR][64];
    __shared__ double FY[2][TY + 1][64];
    const int tx = threadIdx.x, ty = threadIdx.y, t = ty * 64 + tx;
    const int i0 = blockIdx.x * 64, j0 = blockIdx.y * TY;
    const int i = i0 + tx, j = j0 + ty;
    const int ic = min(i, g.Nx + 2), jc = min(j, g.Ny + 2);
    const int ie = i0 - 1, le = 0;                      // centre-type x flux: lane 0 needs the flux of centre i0-1
    const int kbeg = blockIdx.z * kchunk, kend = min(kbeg + kchunk, g.Nz);
    if (kbeg >= kend) return;
    const long long sz = g.Sxy;
    const bool store = (i < g.Nx) && (j < g.Ny);
    const double *u = F.c, *ru = F.ru, *rv = F.rv;
    long long n = g.idx(ic, jc, kbeg);
    // frame row handled by this thread (6 rows x 64 columns)
    // frame rows (3 below + 3 above the tile, 64 columns): HPT cells per thread
    constexpr int NT = 64 * TY, HPT = (6 * 64 + NT - 1) / NT;
    bool hok[HPT];
    int hr[HPT], hcol[HPT];
    long long hn[HPT];
#pragma unroll
    for (int q = 0; q < HPT; ++q) {
        const int h = t + q * NT;
        hok[q] = h < 6 * 64;
        const int hrr = h >> 6;
        hcol[q] = h & 63;
        hr[q] = hok[q] ? ((hrr < 3) ? hrr : TY + hrr) : 0;
        hn[q] = g.idx(min(i0 + hcol[q], g.Nx + 2), min(j0 - 3 + hr[q], g.Ny + 2), kbeg);
    }
    const long long ntop0 = g.idx(ic, min(j0 + TY, g.Ny), kbeg);      // row of the face above the tile (wave 0)

    double r[6];
#pragma unroll
    for (int s = 0; s < 6; ++s) r[s] = u[n + (s - 3) * sz];
    double fz_lo = vflux<T3_U>(g, F, n, kbeg, r[0], r[1], r[2], r[3], r[4], r[5]);
    T[0][ty + 3][tx] = r[3];
#pragma unroll
    for (int q = 0; q < HPT; ++q)
        if (hok[q]) T[0][hr[q]][hcol[q]] = u[hn[q]];
    __syncthreads();

    double edge = 0.0;
    int buf = 0;
    for (int k = kbeg; k < kend; ++k, n += sz) {
        const long long lev = (long long)(k - kbeg) * sz;
        double hnext[HPT];
#pragma unroll
        for (int q = 0; q < HPT; ++q) hnext[q] = hok[q] ? u[hn[q] + lev + sz] : 0.0;
        const double tnew = u[n + 3 * sz];
        if (((k - kbeg) & 63) == 0) {
            const int kk = min(k + tx, kend - 1);
            edge = flux_x_at<T3_U>(g, F, ie, jc, kk);
        }
        const int src = (k - kbeg) & 63;
        const double Ax = g.Ax[k], Ay = g.Ay[k];
        const double c0 = r[3];
        // ---- z ----
        const double fz_hi = vflux<T3_U>(g, F, n + sz, k + 1, r[1], r[2], r[3], r[4], r[5], tnew);
        // ---- x: flux at centre i ----
        const double ax = bz_symm4(Ax * ru[n - 1], Ax * ru[n], Ax * ru[n + 1], Ax * ru[n + 2]);
        const double fx = ax * bz_up5(u[n - 2], u[n - 1], c0, u[n + 1], u[n + 2], u[n + 3], ax > 0.0);
        // ---- y: flux at the own low y-face, advected stencil from the tile column ----
        const double ay = bz_symm4(Ay * rv[n - 2], Ay * rv[n - 1], Ay * rv[n], Ay * rv[n + 1]);
        const double(*Tk)[64] = T[buf];
        const double fy = ay * bz_up5(Tk[ty][tx], Tk[ty + 1][tx], Tk[ty + 2][tx], c0, Tk[ty + 4][tx], Tk[ty + 5][tx], ay > 0.0);
        FY[buf][ty][tx] = fy;
        if (ty == 0) {
            const long long nt = ntop0 + lev;
            const double at = bz_symm4(Ay * rv[nt - 2], Ay * rv[nt - 1], Ay * rv[nt], Ay * rv[nt + 1]);
            FY[buf][TY][tx] = at * bz_up5(Tk[TY][tx], Tk[TY + 1][tx], Tk[TY + 2][tx], Tk[TY + 3][tx], Tk[TY + 4][tx], Tk[TY + 5][tx], at > 0.0);
        }
        // ---- stage level k+1 ----
        T[buf ^ 1][ty + 3][tx] = r[4];
#pragma unroll
        for (int q = 0; q < HPT; ++q)
            if (hok[q]) T[buf ^ 1][hr[q]][hcol[q]] = hnext[q];
        __syncthreads();
        {
            double nb = __shfl_up(fx, 1);
            const double e = __shfl(edge, src);
            if (tx == le) nb = e;
            const double dx = fx - nb;
            const double dy = FY[buf][ty + 1][tx] - fy;
            if (store)
                F.G[n] = bz_rk_apply(E.mode, E.dt, E.alpha, E.oma, E.u0, E.u0_out,
                                     -(g.Vinv_c[k] * (dx + dy + (fz_hi - fz_lo))), ru[n], n);
        }
        fz_lo = fz_hi;
#pragma unroll
        for (int s = 0; s < 5; ++s) r[s] = r[s + 1];
        r[5] = tnew;
        buf ^= 1;
    }
}

// z-momentum: k_w_tend_ring with the w y-stencil in an LDS tile and shared y-face fluxes.
// BM (buoyancy mode) 0: anelastic buoyancy from T, q;  3: the same with the diagnosed q^v, q^l of saturation adjustment;
// 1: none (SlowTendencyMode);  2: compressible slow vertical
// momentum  G^s = G_adv - dz(p - p_r) - g Iz(rho - rho_r)  with F.T = pressure, F.q = total density
// (acoustic_substepping.jl:727-752; the bottom face is never stored, the caller keeps it at 0).
template <int TY, int BM = 0>
__global__ __launch_bounds__(64 * TY) void k_w_tend_lds(DevGrid g, Tend3Fields F, int kchunk, RKEpilogue E)
{
    constexpr int TR = TY + 6;
    __shared__ double T[2][TR][64];
    __shared__ double FY[2][TY + 1][64];
    const int tx = threadIdx.x, ty = threadIdx.y, t = ty * 64 + tx;
    const int i0 = blockIdx.x * 64, j0 = blockIdx.y * TY;
    const int i = i0 + tx, j = j0 + ty;
    const int ic = min(i, g.Nx + 2), jc = min(j, g.Ny + 2);
    const int nact = min(64, g.Nx - i0);
    const int ie = i0 + nact, le = nact - 1;
    const int kbeg = 1 + blockIdx.z * kchunk, kend = min(kbeg + kchunk, g.Nz);
    if (kbeg >= kend) return;
    const long long sz = g.Sxy;
    const bool store = (i < g.Nx) && (j < g.Ny);
    const double *w = F.c, *ru = F.ru, *rv = F.rv, *rw = F.rw;
    const double Az = g.Az;
    long long n = g.idx(ic, jc, kbeg);
    // frame rows (3 below + 3 above the tile, 64 columns): HPT cells per thread
    constexpr int NT = 64 * TY, HPT = (6 * 64 + NT - 1) / NT;
    bool hok[HPT];
    int hr[HPT], hcol[HPT];
    long long hn[HPT];
#pragma unroll
    for (int q = 0; q < HPT; ++q) {
        const int h = t + q * NT;
        hok[q] = h < 6 * 64;
        const int hrr = h >> 6;
        hcol[q] = h & 63;
        hr[q] = hok[q] ? ((hrr < 3) ? hrr : TY + hrr) : 0;
        hn[q] = g.idx(min(i0 + hcol[q], g.Nx + 2), min(j0 - 3 + hr[q], g.Ny + 2), kbeg);
    }
    const long long ntop0 = g.idx(ic, min(j0 + TY, g.Ny), kbeg);
    const bool top = (ty == 0);

    double wr[6], qu[4], qv[4], qt[4], qw[4];       // qt: rho_v ring of the row above the tile (wave 0 only)
#pragma unroll
    for (int s = 0; s < 6; ++s) wr[s] = w[n + (s - 3) * sz];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        const int kk = kbeg - 2 + s;
        qu[s] = g.Ax[kk] * ru[n + (s - 2) * sz];
        qv[s] = g.Ay[kk] * rv[n + (s - 2) * sz];
        qt[s] = top ? g.Ay[kk] * rv[ntop0 + (s - 2) * sz] : 0.0;
        qw[s] = Az * rw[n + (s - 2) * sz];
    }
    double fz_lo, b_lo, r_lo = 0.0;
    {
        const int B = bz_buffer_center(kbeg - 1, g.Nz);
        const double wt = (B == 3) ? bz_symm4(qw[0], qw[1], qw[2], qw[3]) : bz_symm2(qw[1], qw[2]);
        fz_lo = wt * bz_upB(wr[0], wr[1], wr[2], wr[3], wr[4], wr[5], wt > 0.0, B);
        if (BM == 0) b_lo = buoyancy3(g, F.T[n - sz], F.q[n - sz], kbeg - 1);
        else if (BM == 3) b_lo = bz_buoyancy(g, F.T, F.q, n - sz, kbeg - 1);
        else if (BM == 2) { b_lo = F.T[n - sz] - g.p_r[kbeg - 1]; r_lo = F.q[n - sz] - g.rho[kbeg - 1]; }
        else b_lo = 0.0;
    }
    T[0][ty + 3][tx] = wr[3];
#pragma unroll
    for (int q = 0; q < HPT; ++q)
        if (hok[q]) T[0][hr[q]][hcol[q]] = w[hn[q]];
    __syncthreads();

    double edge = 0.0;
    int buf = 0;
    for (int k = kbeg; k < kend; ++k, n += sz) {
        const long long lev = (long long)(k - kbeg) * sz;
        double hnext[HPT];
#pragma unroll
        for (int q = 0; q < HPT; ++q) hnext[q] = hok[q] ? w[hn[q] + lev + sz] : 0.0;
        const double wnew = w[n + 3 * sz];
        const double qwnew = Az * rw[n + 2 * sz];
        if (((k - kbeg) & 63) == 0) {
            const int kk = min(k + tx, kend - 1);
            edge = flux_x_at<T3_W>(g, F, ie, jc, kk);
        }
        const int src = (k - kbeg) & 63;
        const int Bf = bz_buffer_face(k, g.Nz);
        double fz_hi;
        {
            const int B = bz_buffer_center(k, g.Nz);
            const double wt = (B == 3) ? bz_symm4(qw[1], qw[2], qw[3], qwnew) : bz_symm2(qw[2], qw[3]);
            fz_hi = wt * bz_upB(wr[1], wr[2], wr[3], wr[4], wr[5], wnew, wt > 0.0, B);
        }
        const double w0 = wr[3];
        const double ut = (Bf == 3) ? bz_symm4(qu[0], qu[1], qu[2], qu[3]) : bz_symm2(qu[1], qu[2]);
        const double fx = ut * bz_up5(w[n - 3], w[n - 2], w[n - 1], w0, w[n + 1], w[n + 2], ut > 0.0);
        const double vt = (Bf == 3) ? bz_symm4(qv[0], qv[1], qv[2], qv[3]) : bz_symm2(qv[1], qv[2]);
        const double(*Tk)[64] = T[buf];
        const double fy = vt * bz_up5(Tk[ty][tx], Tk[ty + 1][tx], Tk[ty + 2][tx], w0, Tk[ty + 4][tx], Tk[ty + 5][tx], vt > 0.0);
        FY[buf][ty][tx] = fy;
        if (top) {
            const double v2 = (Bf == 3) ? bz_symm4(qt[0], qt[1], qt[2], qt[3]) : bz_symm2(qt[1], qt[2]);
            FY[buf][TY][tx] = v2 * bz_up5(Tk[TY][tx], Tk[TY + 1][tx], Tk[TY + 2][tx], Tk[TY + 3][tx], Tk[TY + 4][tx], Tk[TY + 5][tx], v2 > 0.0);
        }
        T[buf ^ 1][ty + 3][tx] = wr[4];
#pragma unroll
        for (int q = 0; q < HPT; ++q)
            if (hok[q]) T[buf ^ 1][hr[q]][hcol[q]] = hnext[q];
        double b_hi = 0.0, r_hi = 0.0;
        if (BM == 0) b_hi = buoyancy3(g, F.T[n], F.q[n], k);
        else if (BM == 3) b_hi = bz_buoyancy(g, F.T, F.q, n, k);
        else if (BM == 2) { b_hi = F.T[n] - g.p_r[k]; r_hi = F.q[n] - g.rho[k]; }
        // ring advance loads (level k+2)
        const double Axn = g.Ax[k + 2], Ayn = g.Ay[k + 2];
        const double qun = Axn * ru[n + 2 * sz], qvn = Ayn * rv[n + 2 * sz];
        const double qtn = top ? Ayn * rv[ntop0 + lev + 2 * sz] : 0.0;
        __syncthreads();
        {
            double nb = __shfl_down(fx, 1);
            const double e = __shfl(edge, src);
            if (tx == le) nb = e;
            const double dx = nb - fx;
            const double dy = FY[buf][ty + 1][tx] - fy;
            const double adv = -(g.Vinv_f[k] * (dx + dy + (fz_hi - fz_lo)));
            double Gval;
            if (BM == 0 || BM == 3) Gval = adv + 0.5 * (b_lo + b_hi);
            else if (BM == 2) Gval = adv - (b_hi - b_lo) * g.rdzf[k] - g.g * ((r_hi + r_lo) / 2.0);
            else Gval = adv;
            if (store)
                F.G[n] = bz_rk_apply(E.mode, E.dt, E.alpha, E.oma, E.u0, E.u0_out, Gval, rw[n], n);
        }
        fz_lo = fz_hi;
        b_lo = b_hi;
        r_lo = r_hi;
#pragma unroll
        for (int s = 0; s < 3; ++s) { qu[s] = qu[s + 1]; qv[s] = qv[s + 1]; qt[s] = qt[s + 1]; qw[s] = qw[s + 1]; }
        qu[3] = qun; qv[3] = qvn; qt[3] = qtn; qw[3] = qwnew;
#pragma unroll
        for (int s = 0; s < 5; ++s) wr[s] = wr[s + 1];
        wr[5] = wnew;
        buf ^= 1;
    }
}

// y-momentum: every y-stencil through LDS.  The v kernel is the one whose advecting fluxes are centred-in-y
// interpolations (rho_u, rho_w over rows j-2..j+1, rho_v over j-1..j+2) on top of the y-WENO stencil of v itself:
// rocprofv3 PMC showed 12.7 GB fetched per launch for 5.4 GB of input (2.4x).  Four tiles per level, all staged
// one level ahead: v (rows j0-3..j0+TY+2), Ax*rho_u, Ay*rho_v, Az*rho_w(k+1) (rows j0-2..j0+TY+1); interiors come
// from the threads' own (coalesced) loads / the v ring, the 16 frame rows from two prefetched loads per thread.
// y fluxes live at cell centres here (G(j) = F(j) - F(j-1)): wave 0 also evaluates the centre below the tile.
template <int TY>
__global__ __launch_bounds__(64 * TY) void k_v_tend_lds(DevGrid g, Tend3Fields F, int kchunk, RKEpilogue E)
{
    constexpr int RV = TY + 6, RM = TY + 4;               // rows of the v tile / of the momentum tiles
    __shared__ double Tv[2][RV][64];
    __shared__ double Tm[2][3][RM][64];                   // 0: Ax*rho_u, 1: Ay*rho_v, 2: Az*rho_w at the upper z-face
    __shared__ double FY[2][TY + 1][64];
    constexpr int NT = 64 * TY, NFR = 16 * 64, HPT = (NFR + NT - 1) / NT;
    const int tx = threadIdx.x, ty = threadIdx.y, t = ty * 64 + tx;
    const int i0 = blockIdx.x * 64, j0 = blockIdx.y * TY;
    const int i = i0 + tx, j = j0 + ty;
    const int ic = min(i, g.Nx + 2), jc = min(j, g.Ny + 2);
    const int nact = min(64, g.Nx - i0);
    const int ie = i0 + nact, le = nact - 1;              // x flux at x-faces: last lane needs face i0+nact
    const int kbeg = blockIdx.z * kchunk, kend = min(kbeg + kchunk, g.Nz);
    if (kbeg >= kend) return;
    const long long sz = g.Sxy;
    const bool store = (i < g.Nx) && (j < g.Ny);
    const double *v = F.c, *ru = F.ru, *rv = F.rv, *rw = F.rw;
    const double Az = g.Az;
    long long n = g.idx(ic, jc, kbeg);

    // frame rows: id 0..5 v tile rows {0,1,2,TY+3,TY+4,TY+5}; 6..8 rho_u rows {0,1,TY+2}; 9..12 rho_v rows
    // {0,1,TY+2,TY+3}; 13..15 rho_w rows {0,1,TY+2}   (momentum-tile row r' <-> grid row j0-2+r')
    bool hok[HPT];
    int hsel[HPT], hrow[HPT], hcol[HPT];
    long long hn[HPT];
#pragma unroll
    for (int q = 0; q < HPT; ++q) {
        const int h = t + q * NT;
        hok[q] = h < NFR;
        const int id = hok[q] ? (h >> 6) : 0;
        hcol[q] = h & 63;
        int sel, row, grow;                                // sel: 0 v, 1 rho_u, 2 rho_v, 3 rho_w
        if (id < 6) { sel = 0; row = (id < 3) ? id : TY + id; grow = j0 - 3 + row; }
        else if (id < 9) { sel = 1; const int m = id - 6; row = (m < 2) ? m : TY + 2; grow = j0 - 2 + row; }
        else if (id < 13) { sel = 2; const int m = id - 9; row = (m < 2) ? m : TY + m; grow = j0 - 2 + row; }
        else { sel = 3; const int m = id - 13; row = (m < 2) ? m : TY + 2; grow = j0 - 2 + row; }
        hsel[q] = sel; hrow[q] = row;
        hn[q] = g.idx(min(i0 + hcol[q], g.Nx + 2), min(grow, g.Ny + 2), kbeg) + (sel == 3 ? sz : 0);
    }
    auto frame_load = [&](int q, long long lev, int klev) -> double {      // value to stage for level klev
        const double *src = (hsel[q] == 0) ? v : (hsel[q] == 1) ? ru : (hsel[q] == 2) ? rv : rw;
        const double fac = (hsel[q] == 0) ? 1.0 : (hsel[q] == 1) ? g.Ax[klev] : (hsel[q] == 2) ? g.Ay[klev] : Az;
        return fac * src[hn[q] + lev];
    };
    auto frame_store = [&](int b, int q, double val) {
        if (hsel[q] == 0) Tv[b][hrow[q]][hcol[q]] = val;
        else Tm[b][hsel[q] - 1][hrow[q]][hcol[q]] = val;
    };

    double r[6];
#pragma unroll
    for (int s = 0; s < 6; ++s) r[s] = v[n + (s - 3) * sz];
    // lower z-face flux of the chunk, straight from global memory (once per chunk)
    double fz_lo = vflux<T3_V>(g, F, n, kbeg, r[0], r[1], r[2], r[3], r[4], r[5]);
    double rv_cur = rv[n];
    Tv[0][ty + 3][tx] = r[3];
    Tm[0][0][ty + 2][tx] = g.Ax[kbeg] * ru[n];
    Tm[0][1][ty + 2][tx] = g.Ay[kbeg] * rv_cur;
    Tm[0][2][ty + 2][tx] = Az * rw[n + sz];
#pragma unroll
    for (int q = 0; q < HPT; ++q)
        if (hok[q]) frame_store(0, q, frame_load(q, 0, kbeg));
    __syncthreads();

    double edge = 0.0;
    int buf = 0;
    for (int k = kbeg; k < kend; ++k, n += sz) {
        const long long lev = (long long)(k + 1 - kbeg) * sz;
        double hnext[HPT];
#pragma unroll
        for (int q = 0; q < HPT; ++q) hnext[q] = hok[q] ? frame_load(q, lev, k + 1) : 0.0;
        const double tnew = v[n + 3 * sz];
        const double ru_n = g.Ax[k + 1] * ru[n + sz], rv_nraw = rv[n + sz], rw_n = Az * rw[n + 2 * sz];
        if (((k - kbeg) & 63) == 0) {
            const int kk = min(k + tx, kend - 1);
            edge = flux_x_at<T3_V>(g, F, ie, jc, kk);
        }
        const int src = (k - kbeg) & 63;
        const double c0 = r[3];
        const double(*V)[64] = Tv[buf];
        const double(*MU)[64] = Tm[buf][0];
        const double(*MV)[64] = Tm[buf][1];
        const double(*MW)[64] = Tm[buf][2];
        // ---- z: advecting flux at (y-face j, z-face k+1) from the rho_w tile rows j-2..j+1 ----
        const double wt = bz_symm4(MW[ty][tx], MW[ty + 1][tx], MW[ty + 2][tx], MW[ty + 3][tx]);
        const double fz_hi = wt * bz_upB(r[1], r[2], r[3], r[4], r[5], tnew, wt > 0.0, bz_buffer_face(k + 1, g.Nz));
        // ---- x: flux at (x-face i, y-face j): rho_u rows j-2..j+1, v x-stencil from global (L1) ----
        const double ut = bz_symm4(MU[ty][tx], MU[ty + 1][tx], MU[ty + 2][tx], MU[ty + 3][tx]);
        const double fx = ut * bz_up5(v[n - 3], v[n - 2], v[n - 1], c0, v[n + 1], v[n + 2], ut > 0.0);
        // ---- y: flux at centre j: rho_v rows j-1..j+2, v rows j-2..j+3 ----
        const double vt = bz_symm4(MV[ty + 1][tx], MV[ty + 2][tx], MV[ty + 3][tx], MV[ty + 4][tx]);
        const double fy = vt * bz_up5(V[ty + 1][tx], V[ty + 2][tx], c0, V[ty + 4][tx], V[ty + 5][tx], V[ty + 6][tx], vt > 0.0);
        FY[buf][ty + 1][tx] = fy;
        if (ty == 0) {       // centre j0-1: rho_v rows j0-2..j0+1, v rows j0-3..j0+2
            const double vb = bz_symm4(MV[0][tx], MV[1][tx], MV[2][tx], MV[3][tx]);
            FY[buf][0][tx] = vb * bz_up5(V[0][tx], V[1][tx], V[2][tx], V[3][tx], V[4][tx], V[5][tx], vb > 0.0);
        }
        // ---- stage level k+1 ----
        Tv[buf ^ 1][ty + 3][tx] = r[4];
        Tm[buf ^ 1][0][ty + 2][tx] = ru_n;
        Tm[buf ^ 1][1][ty + 2][tx] = g.Ay[k + 1] * rv_nraw;
        Tm[buf ^ 1][2][ty + 2][tx] = rw_n;
#pragma unroll
        for (int q = 0; q < HPT; ++q)
            if (hok[q]) frame_store(buf ^ 1, q, hnext[q]);
        __syncthreads();
        {
            double nb = __shfl_down(fx, 1);
            const double e = __shfl(edge, src);
            if (tx == le) nb = e;
            const double dx = nb - fx;
            const double dy = fy - FY[buf][ty][tx];
            if (store)
                F.G[n] = bz_rk_apply(E.mode, E.dt, E.alpha, E.oma, E.u0, E.u0_out,
                                     -(g.Vinv_c[k] * (dx + dy + (fz_hi - fz_lo))), rv_cur, n);
        }
        fz_lo = fz_hi;
        rv_cur = rv_nraw;
#pragma unroll
        for (int s = 0; s < 5; ++s) r[s] = r[s + 1];
        r[5] = tnew;
        buf ^= 1;
    }
}
