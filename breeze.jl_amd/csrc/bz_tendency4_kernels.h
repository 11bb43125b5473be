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
    const int i0 = blockIdx.x * 64, j0 = blockIdx.y * TY;
    const int i = i0 + tx, j = j0 + ty;
    // ragged tiles: out-of-range threads still stage true halo values (never clamp into the interior)
    const int ic = min(i, g.Nx + 2), jc = min(j, g.Ny + 2);
    const int nact = min(64, g.Nx - i0);
    const int ie = i0 + nact, le = nact - 1;
    const int kbeg = blockIdx.z * kchunk, kend = min(kbeg + kchunk, g.Nz);
    if (kbeg >= kend) return;                           // block-uniform
    const long long sy = g.Sx, sz = g.Sxy;
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
