// bz_tendency.hip — WENO-5 flux-form tendencies of the dry anelastic AtmosphereModel.
//   scalar_tendency / potential_temperature_tendency: G = -div_rhoUc
//       /root/reference/src/Advection.jl:20-35, src/AtmosphereModels/dynamics_kernel_functions.jl:132-159,
//       src/PotentialTemperatureFormulations/potential_temperature_tendency.jl:66-106
//   x/y/z_momentum_tendency: G = -div(rhoU (x) u) [+ Iz(buoyancy) for z]
//       src/AtmosphereModels/dynamics_kernel_functions.jl:54-130, src/AnelasticEquations/anelastic_buoyancy.jl:36-72
//   launcher: compute_tendencies!  src/AtmosphereModels/update_atmosphere_model_state.jl:294-387
//
// Kernel shape (all four kernels): a block owns a 64 x TYB tile of columns and marches upward
// through a chunk of z levels.  The 6-value vertical stencil of the advected quantity lives in a
// register ring, the vertical flux through the lower face is carried from the previous level, and
// horizontal stencils are read straight from global memory (L1/L2 resident: the tile's rows were
// touched by the neighbouring lanes / rows a few instructions earlier).
#include <cstdlib>

#include "bz_internal.h"
#include "bz_weno.h"

#define TYB 4

// Bounded y ((Periodic, Bounded, Bounded)): the WENO buffer that fits at y-face / y-centre j — wave-uniform (a wavefront is one row) —
// and the Centered advecting-flux interpolation of the same order, as in z (symm_z_face below); 3 everywhere on a periodic y
// Bounded x ((Bounded, Flat, Bounded)): the same by lane — the buffer differs between the lanes next to a wall, so the reconstruction
// branches diverge in the first and last wavefront of a row only
__device__ __forceinline__ int bx_face(const DevGrid &g, int i) { return g.bounded_x ? bz_buffer_face(i, g.Nx) : 3; }
__device__ __forceinline__ int bx_center(const DevGrid &g, int i) { return g.bounded_x ? bz_buffer_center(i, g.Nx) : 3; }
__device__ __forceinline__ int by_face(const DevGrid &g, int j) { return g.bounded_y ? bz_buffer_face(j, g.Ny) : 3; }
__device__ __forceinline__ int by_center(const DevGrid &g, int j) { return g.bounded_y ? bz_buffer_center(j, g.Ny) : 3; }
__device__ __forceinline__ double symm_y(double qm2, double qm1, double q0, double qp1, int B)
{
    return (B == 3) ? bz_symm4(qm2, qm1, q0, qp1) : bz_symm2(qm1, q0);
}

__device__ __forceinline__ double flux_z_scalar(const DevGrid &g, double wt, double m3, double m2,
                                                double m1, double p0, double p1, double p2, int kface)
{
    int B = bz_buffer_face(kface, g.Nz);
    double cR = bz_upB(m3, m2, m1, p0, p1, p2, wt > 0.0, B);
    return g.rho_f[kface] * ((g.Az * wt) * cR);
}

__global__ __launch_bounds__(64 * TYB) void k_scalar_tendency(DevGrid g, double *__restrict__ Gc,
                                                             const double *__restrict__ u,
                                                             const double *__restrict__ v,
                                                             const double *__restrict__ w,
                                                             const double *__restrict__ c, int kchunk)
{
    const int i = blockIdx.x * 64 + threadIdx.x;
    const int j = blockIdx.y * TYB + threadIdx.y;
    if (i >= g.Nx || j >= g.Ny) return;
    const int k0 = blockIdx.z * kchunk;
    const int k1 = min(k0 + kchunk, g.Nz);
    const long long sy = g.Sx, sz = g.Sxy;
    long long n = g.idx(i, j, k0);

    double zm3 = c[n - 3 * sz], zm2 = c[n - 2 * sz], zm1 = c[n - sz], z0 = c[n], zp1 = c[n + sz], zp2 = c[n + 2 * sz];
    double Fz_lo = flux_z_scalar(g, w[n], zm3, zm2, zm1, z0, zp1, zp2, k0);

    for (int k = k0; k < k1; ++k, n += sz) {
        double zp3 = c[n + 3 * sz];
        double Fz_hi = flux_z_scalar(g, w[n + sz], zm2, zm1, z0, zp1, zp2, zp3, k + 1);

        const double rho = g.rho[k], Ax = g.Ax[k], Ay = g.Ay[k];
        double xm3 = c[n - 3], xm2 = c[n - 2], xm1 = c[n - 1], xp1 = c[n + 1], xp2 = c[n + 2], xp3 = c[n + 3];
        double u0 = u[n], u1 = u[n + 1];
        double Fx_lo = rho * ((Ax * u0) * bz_upB(xm3, xm2, xm1, z0, xp1, xp2, u0 > 0.0, bx_face(g, i)));
        double Fx_hi = rho * ((Ax * u1) * bz_upB(xm2, xm1, z0, xp1, xp2, xp3, u1 > 0.0, bx_face(g, i + 1)));

        double Fy_lo = 0.0, Fy_hi = 0.0;
        if (!g.flat_y) {      // a Flat y direction has no faces
            double ym3 = c[n - 3 * sy], ym2 = c[n - 2 * sy], ym1 = c[n - sy], yp1 = c[n + sy], yp2 = c[n + 2 * sy], yp3 = c[n + 3 * sy];
            double v0 = v[n], v1 = v[n + sy];
            Fy_lo = rho * ((Ay * v0) * bz_upB(ym3, ym2, ym1, z0, yp1, yp2, v0 > 0.0, by_face(g, j)));
            Fy_hi = rho * ((Ay * v1) * bz_upB(ym2, ym1, z0, yp1, yp2, yp3, v1 > 0.0, by_face(g, j + 1)));
        }

        Gc[n] = -(g.Vinv_c[k] * ((Fx_hi - Fx_lo) + (Fy_hi - Fy_lo) + (Fz_hi - Fz_lo)));

        zm3 = zm2; zm2 = zm1; zm1 = z0; z0 = zp1; zp1 = zp2; zp2 = zp3;
        Fz_lo = Fz_hi;
    }
}

// ---------------------------------------------------------------------------------------------
// x-momentum
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ double flux_Wu(const DevGrid &g, const double *__restrict__ rw, long long nf,
                                          double m3, double m2, double m1, double p0, double p1, double p2,
                                          int kface, int i)
{   // at (f,c,f): advecting flux = Centered4 in x of Az*rho_w to x-face i
    double Az = g.Az;
    double wt = symm_y(Az * rw[nf - 2], Az * rw[nf - 1], Az * rw[nf], Az * rw[nf + 1], bx_face(g, i));
    double uR = bz_upB(m3, m2, m1, p0, p1, p2, wt > 0.0, bz_buffer_face(kface, g.Nz));
    return wt * uR;
}

__global__ __launch_bounds__(64 * TYB) void k_u_tendency(DevGrid g, double *__restrict__ Gu,
                                                        const double *__restrict__ ru,
                                                        const double *__restrict__ rv,
                                                        const double *__restrict__ rw,
                                                        const double *__restrict__ u, int kchunk, RKEpilogue E)
{
    const int i = blockIdx.x * 64 + threadIdx.x;
    const int j = blockIdx.y * TYB + threadIdx.y;
    if (i >= g.Nx || j >= g.Ny) return;
    if (g.bounded_x && i == 0) return;               // the wall face is never updated
    const int k0 = blockIdx.z * kchunk;
    const int k1 = min(k0 + kchunk, g.Nz);
    const long long sy = g.Sx, sz = g.Sxy;
    long long n = g.idx(i, j, k0);
    const int Bxc = bx_center(g, i), Bxcm = bx_center(g, i - 1), Bxf = bx_face(g, i);

    double zm3 = u[n - 3 * sz], zm2 = u[n - 2 * sz], zm1 = u[n - sz], z0 = u[n], zp1 = u[n + sz], zp2 = u[n + 2 * sz];
    double Fz_lo = flux_Wu(g, rw, n, zm3, zm2, zm1, z0, zp1, zp2, k0, i);

    for (int k = k0; k < k1; ++k, n += sz) {
        double zp3 = u[n + 3 * sz];
        double Fz_hi = flux_Wu(g, rw, n + sz, zm2, zm1, z0, zp1, zp2, zp3, k + 1, i);
        const double Ax = g.Ax[k], Ay = g.Ay[k];

        // x: F_Uu at centres i (hi) and i-1 (lo)
        double q_m2 = Ax * ru[n - 2], q_m1 = Ax * ru[n - 1], q_0 = Ax * ru[n], q_p1 = Ax * ru[n + 1], q_p2 = Ax * ru[n + 2];
        double xm3 = u[n - 3], xm2 = u[n - 2], xm1 = u[n - 1], xp1 = u[n + 1], xp2 = u[n + 2], xp3 = u[n + 3];
        double ut_hi = (Bxc == 3) ? bz_symm4(q_m1, q_0, q_p1, q_p2) : bz_symm2(q_0, q_p1);      // centre targets: order 2 next to an x wall
        double ut_lo = (Bxcm == 3) ? bz_symm4(q_m2, q_m1, q_0, q_p1) : bz_symm2(q_m1, q_0);
        double Fx_hi = ut_hi * bz_upB(xm2, xm1, z0, xp1, xp2, xp3, ut_hi > 0.0, Bxc);
        double Fx_lo = ut_lo * bz_upB(xm3, xm2, xm1, z0, xp1, xp2, ut_lo > 0.0, Bxcm);

        // y: F_Vu at y-faces j (lo) and j+1 (hi)
        double Fy_lo = 0.0, Fy_hi = 0.0;
        if (!g.flat_y) {
            double vt_lo = symm_y(Ay * rv[n - 2], Ay * rv[n - 1], Ay * rv[n], Ay * rv[n + 1], Bxf);
            double vt_hi = symm_y(Ay * rv[n + sy - 2], Ay * rv[n + sy - 1], Ay * rv[n + sy], Ay * rv[n + sy + 1], Bxf);
            double ym3 = u[n - 3 * sy], ym2 = u[n - 2 * sy], ym1 = u[n - sy], yp1 = u[n + sy], yp2 = u[n + 2 * sy], yp3 = u[n + 3 * sy];
            Fy_lo = vt_lo * bz_upB(ym3, ym2, ym1, z0, yp1, yp2, vt_lo > 0.0, by_face(g, j));
            Fy_hi = vt_hi * bz_upB(ym2, ym1, z0, yp1, yp2, yp3, vt_hi > 0.0, by_face(g, j + 1));
        }

        Gu[n] = bz_rk_apply(E.mode, E.dt, E.alpha, E.oma, E.u0, E.u0_out,
                            -(g.Vinv_c[k] * ((Fx_hi - Fx_lo) + (Fy_hi - Fy_lo) + (Fz_hi - Fz_lo))), ru[n], n);

        zm3 = zm2; zm2 = zm1; zm1 = z0; z0 = zp1; zp1 = zp2; zp2 = zp3;
        Fz_lo = Fz_hi;
    }
}

// ---------------------------------------------------------------------------------------------
// y-momentum
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ double flux_Wv(const DevGrid &g, const double *__restrict__ rw, long long nf,
                                          double m3, double m2, double m1, double p0, double p1, double p2,
                                          int kface, int j)
{   // at (c,f,f): Centered4 in y of Az*rho_w to y-face j
    double Az = g.Az;
    long long sy = g.Sx;
    double wt = g.flat_y ? Az * rw[nf] : symm_y(Az * rw[nf - 2 * sy], Az * rw[nf - sy], Az * rw[nf], Az * rw[nf + sy], by_face(g, j));
    double vR = bz_upB(m3, m2, m1, p0, p1, p2, wt > 0.0, bz_buffer_face(kface, g.Nz));
    return wt * vR;
}

__global__ __launch_bounds__(64 * TYB) void k_v_tendency(DevGrid g, double *__restrict__ Gv,
                                                        const double *__restrict__ ru,
                                                        const double *__restrict__ rv,
                                                        const double *__restrict__ rw,
                                                        const double *__restrict__ v, int kchunk, RKEpilogue E)
{
    const int i = blockIdx.x * 64 + threadIdx.x;
    const int j = blockIdx.y * TYB + threadIdx.y;
    if (i >= g.Nx || j >= g.Ny) return;
    if (g.bounded_y && j == 0) return;               // the wall face is never updated (like w at k = 0)
    const int k0 = blockIdx.z * kchunk;
    const int k1 = min(k0 + kchunk, g.Nz);
    const long long sy = g.Sx, sz = g.Sxy;
    long long n = g.idx(i, j, k0);
    const int Bf = by_face(g, j), Bc = by_center(g, j), Bcm = by_center(g, j - 1);

    double zm3 = v[n - 3 * sz], zm2 = v[n - 2 * sz], zm1 = v[n - sz], z0 = v[n], zp1 = v[n + sz], zp2 = v[n + 2 * sz];
    double Fz_lo = flux_Wv(g, rw, n, zm3, zm2, zm1, z0, zp1, zp2, k0, j);

    for (int k = k0; k < k1; ++k, n += sz) {
        double zp3 = v[n + 3 * sz];
        double Fz_hi = flux_Wv(g, rw, n + sz, zm2, zm1, z0, zp1, zp2, zp3, k + 1, j);
        const double Ax = g.Ax[k], Ay = g.Ay[k];

        // x: F_Uv at x-faces i (lo) and i+1 (hi); advecting flux = Centered4 in y of Ax*rho_u
        // (Flat y: the y face of the v cell coincides with its centre, no interpolation)
        double ut_lo = g.flat_y ? Ax * ru[n] : symm_y(Ax * ru[n - 2 * sy], Ax * ru[n - sy], Ax * ru[n], Ax * ru[n + sy], Bf);
        double ut_hi = g.flat_y ? Ax * ru[n + 1] : symm_y(Ax * ru[n + 1 - 2 * sy], Ax * ru[n + 1 - sy], Ax * ru[n + 1], Ax * ru[n + 1 + sy], Bf);
        double xm3 = v[n - 3], xm2 = v[n - 2], xm1 = v[n - 1], xp1 = v[n + 1], xp2 = v[n + 2], xp3 = v[n + 3];
        double Fx_lo = ut_lo * bz_upB(xm3, xm2, xm1, z0, xp1, xp2, ut_lo > 0.0, bx_face(g, i));
        double Fx_hi = ut_hi * bz_upB(xm2, xm1, z0, xp1, xp2, xp3, ut_hi > 0.0, bx_face(g, i + 1));

        // y: F_Vv at centres j (hi) and j-1 (lo)
        double Fy_hi = 0.0, Fy_lo = 0.0;
        if (!g.flat_y) {
            double q_m2 = Ay * rv[n - 2 * sy], q_m1 = Ay * rv[n - sy], q_0 = Ay * rv[n], q_p1 = Ay * rv[n + sy], q_p2 = Ay * rv[n + 2 * sy];
            double ym3 = v[n - 3 * sy], ym2 = v[n - 2 * sy], ym1 = v[n - sy], yp1 = v[n + sy], yp2 = v[n + 2 * sy], yp3 = v[n + 3 * sy];
            // centre targets: faces j-1 .. j+2 around centre j (hi), j-2 .. j+1 around centre j-1 (lo); order 2 next to a wall
            double vt_hi = (Bc == 3) ? bz_symm4(q_m1, q_0, q_p1, q_p2) : bz_symm2(q_0, q_p1);
            double vt_lo = (Bcm == 3) ? bz_symm4(q_m2, q_m1, q_0, q_p1) : bz_symm2(q_m1, q_0);
            Fy_hi = vt_hi * bz_upB(ym2, ym1, z0, yp1, yp2, yp3, vt_hi > 0.0, Bc);
            Fy_lo = vt_lo * bz_upB(ym3, ym2, ym1, z0, yp1, yp2, vt_lo > 0.0, Bcm);
        }

        Gv[n] = bz_rk_apply(E.mode, E.dt, E.alpha, E.oma, E.u0, E.u0_out,
                            -(g.Vinv_c[k] * ((Fx_hi - Fx_lo) + (Fy_hi - Fy_lo) + (Fz_hi - Fz_lo))), rv[n], n);

        zm3 = zm2; zm2 = zm1; zm1 = z0; z0 = zp1; zp1 = zp2; zp2 = zp3;
        Fz_lo = Fz_hi;
    }
}

// ---------------------------------------------------------------------------------------------
// z-momentum (+ buoyancy), interior faces k = 1..Nz-1 only
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ double buoyancy_ccc(const DevGrid &g, double T, double q, int k)
{   // anelastic_buoyancy.jl:36-72, dry reference (R_m,r = Rd)
    double Rmr = g.Rd;
    double Rm = (1.0 - q) * g.Rd + q * g.Rv;
    double rhop = g.rho[k] * (Rmr * g.T_r[k] / (Rm * T) - 1.0);
    return -g.g * rhop;
}

// Centered in z (order 4 where the WENO5 buffer fits, else 2) of A(k)*M to z-face k
__device__ __forceinline__ double symm_z_face(const DevGrid &g, const double *__restrict__ A,
                                              const double *__restrict__ M, long long n, int k, int B)
{
    long long sz = g.Sxy;
    if (B == 3) return bz_symm4(A[k - 2] * M[n - 2 * sz], A[k - 1] * M[n - sz], A[k] * M[n], A[k + 1] * M[n + sz]);
    return bz_symm2(A[k - 1] * M[n - sz], A[k] * M[n]);
}

__device__ __forceinline__ double flux_Ww(const DevGrid &g, const double *__restrict__ rw, long long n, int k,
                                          double m3, double m2, double m1, double p0, double p1, double p2)
{   // at (c,c,c) centre k: faces k-1..k+2 for the advecting flux, (k-2..k+3) for w
    long long sz = g.Sxy;
    int B = bz_buffer_center(k, g.Nz);
    double Az = g.Az;
    double wt = (B == 3) ? bz_symm4(Az * rw[n - sz], Az * rw[n], Az * rw[n + sz], Az * rw[n + 2 * sz])
                         : bz_symm2(Az * rw[n], Az * rw[n + sz]);
    double wR = bz_upB(m3, m2, m1, p0, p1, p2, wt > 0.0, B);
    return wt * wR;
}

template <bool BUOY = true>
__global__ __launch_bounds__(64 * TYB) void k_w_tendency(DevGrid g, double *__restrict__ Gw,
                                                        const double *__restrict__ ru,
                                                        const double *__restrict__ rv,
                                                        const double *__restrict__ rw,
                                                        const double *__restrict__ w,
                                                        const double *__restrict__ T,
                                                        const double *__restrict__ qv, int kchunk)
{
    const int i = blockIdx.x * 64 + threadIdx.x;
    const int j = blockIdx.y * TYB + threadIdx.y;
    if (i >= g.Nx || j >= g.Ny) return;
    const int k0 = 1 + blockIdx.z * kchunk;          // first face of this chunk
    const int k1 = min(k0 + kchunk, g.Nz);           // faces k0 .. k1-1  (<= Nz-1)
    if (k0 >= k1) return;
    const long long sy = g.Sx, sz = g.Sxy;
    long long n = g.idx(i, j, k0);

    // ring of w around centre k-1 (faces k-3 .. k+2) to start; centre k needs faces k-2..k+3
    double zm3 = w[n - 3 * sz], zm2 = w[n - 2 * sz], zm1 = w[n - sz], z0 = w[n], zp1 = w[n + sz], zp2 = w[n + 2 * sz];
    double Fz_lo = flux_Ww(g, rw, n - sz, k0 - 1, zm3, zm2, zm1, z0, zp1, zp2);
    double b_lo = BUOY ? bz_buoyancy(g, T, qv, n - sz, k0 - 1) : 0.0;

    for (int k = k0; k < k1; ++k, n += sz) {
        double zp3 = w[n + 3 * sz];
        double Fz_hi = flux_Ww(g, rw, n, k, zm2, zm1, z0, zp1, zp2, zp3);
        double b_hi = BUOY ? bz_buoyancy(g, T, qv, n, k) : 0.0;
        const int Bf = bz_buffer_face(k, g.Nz);

        // x: F_Uw at x-faces i (lo), i+1 (hi): advecting flux = centred-in-z of Ax(k)*rho_u to face k
        double ut_lo = symm_z_face(g, g.Ax, ru, n, k, Bf);
        double ut_hi = symm_z_face(g, g.Ax, ru, n + 1, k, Bf);
        double xm3 = w[n - 3], xm2 = w[n - 2], xm1 = w[n - 1], xp1 = w[n + 1], xp2 = w[n + 2], xp3 = w[n + 3];
        double Fx_lo = ut_lo * bz_upB(xm3, xm2, xm1, z0, xp1, xp2, ut_lo > 0.0, bx_face(g, i));
        double Fx_hi = ut_hi * bz_upB(xm2, xm1, z0, xp1, xp2, xp3, ut_hi > 0.0, bx_face(g, i + 1));

        // y: F_Vw at y-faces j (lo), j+1 (hi)
        double Fy_lo = 0.0, Fy_hi = 0.0;
        if (!g.flat_y) {
            double vt_lo = symm_z_face(g, g.Ay, rv, n, k, Bf);
            double vt_hi = symm_z_face(g, g.Ay, rv, n + sy, k, Bf);
            double ym3 = w[n - 3 * sy], ym2 = w[n - 2 * sy], ym1 = w[n - sy], yp1 = w[n + sy], yp2 = w[n + 2 * sy], yp3 = w[n + 3 * sy];
            Fy_lo = vt_lo * bz_upB(ym3, ym2, ym1, z0, yp1, yp2, vt_lo > 0.0, by_face(g, j));
            Fy_hi = vt_hi * bz_upB(ym2, ym1, z0, yp1, yp2, yp3, vt_hi > 0.0, by_face(g, j + 1));
        }

        const double adv = -(g.Vinv_f[k] * ((Fx_hi - Fx_lo) + (Fy_hi - Fy_lo) + (Fz_hi - Fz_lo)));
        Gw[n] = BUOY ? adv + 0.5 * (b_lo + b_hi) : adv;

        zm3 = zm2; zm2 = zm1; zm1 = z0; z0 = zp1; zp1 = zp2; zp2 = zp3;
        Fz_lo = Fz_hi;
        b_lo = b_hi;
    }
}

// StaticEnergy: G_rho_e -= Iz_c(w * Iz_f(buoyancy))  (static_energy_tendency.jl:55-67, dynamics_kernel_functions.jl:40-51)
__global__ __launch_bounds__(256) void k_energy_buoyancy_flux(DevGrid g, double *__restrict__ Ge, const double *__restrict__ w,
                                                              const double *__restrict__ T, const double *__restrict__ qv)
{
    const int i = blockIdx.x * 256 + threadIdx.x, j = blockIdx.y, k = blockIdx.z;
    if (i >= g.Nx) return;
    const long long n = g.idx(i, j, k), sz = g.Sxy;
    const double b_m = bz_buoyancy(g, T, qv, n - sz, k - 1);
    const double b_0 = bz_buoyancy(g, T, qv, n, k);
    const double b_p = bz_buoyancy(g, T, qv, n + sz, k + 1);
    const double f_lo = ((b_0 + b_m) / 2.0) * w[n];
    const double f_hi = ((b_p + b_0) / 2.0) * w[n + sz];
    Ge[n] = Ge[n] - (f_hi + f_lo) / 2.0;
}

static int pick_kchunk(const DevGrid &g, int nlev)
{
    long long tiles = (long long)((g.Nx + 63) / 64) * ((g.Ny + TYB - 1) / TYB);
    long long want = (4096 + tiles - 1) / tiles;          // aim for >= 4096 blocks
    if (want < 1) want = 1;
    long long maxchunks = nlev / 8 > 0 ? nlev / 8 : 1;    // keep >= 8 levels per chunk
    if (want > maxchunks) want = maxchunks;
    return (int)((nlev + want - 1) / want);
}

// momentum advection by the per-operator kernels alone (no buoyancy): the slow momentum tendencies of the split-explicit compressible
// model on (Periodic, Flat, Bounded) grids, where the LDS-tiled kernels (y halo rows in their tiles) do not apply
int bzi_momentum_advection_gen1(bz_ctx *ctx, const bz_state *s, const bz_prognostic *G)
{
    const DevGrid &g = ctx->dg;
    const dim3 block(64, TYB);
    const int kc = pick_kchunk(g, g.Nz), kcw = pick_kchunk(g, g.Nz - 1);
    const dim3 grid((g.Nx + 63) / 64, (g.Ny + TYB - 1) / TYB, (g.Nz + kc - 1) / kc);
    const dim3 gridw(grid.x, grid.y, (g.Nz - 1 + kcw - 1) / kcw);
    {
        ProfileScope ps(ctx, "x_momentum_tendency");
        hipLaunchKernelGGL(k_u_tendency, grid, block, 0, ctx->stream, g, G->rho_u, s->rho_u, s->rho_v, s->rho_w, s->u, kc, RKEpilogue());
    }
    {
        ProfileScope ps(ctx, "y_momentum_tendency");
        hipLaunchKernelGGL(k_v_tendency, grid, block, 0, ctx->stream, g, G->rho_v, s->rho_u, s->rho_v, s->rho_w, s->v, kc, RKEpilogue());
    }
    {
        ProfileScope ps(ctx, "z_momentum_tendency");
        hipLaunchKernelGGL(k_w_tendency<false>, gridw, block, 0, ctx->stream, g, G->rho_w, s->rho_u, s->rho_v, s->rho_w, s->w,
                           (const double *)nullptr, (const double *)nullptr, kcw);
    }
    BZ_LAUNCH_CHECK();
    return BZ_OK;
}

// momentum_advection and scalar_advection of different orders (atmosphere_model.jl:80-82,148-158; examples/tropical_cyclone_world.jl:167-169,
// examples/prescribed_sea_surface_temperature.jl:72-73: momentum WENO(order = 9), scalars WENO(order = 5)): the momentum kernels of the
// momentum order, one scalar kernel per scalar of the scalars' order, then the order-independent terms — operator by operator, as the
// reference launches them (update_atmosphere_model_state.jl:294-387)
static int compute_tendencies_mixed(bz_ctx *ctx, const bz_state *s, const bz_prognostic *G)
{
    const DevGrid &g = ctx->dg;
    int rc;
    if (ctx->bounded_mask && ctx->scalar_R != 3) {
        ctx->last_error = "bounds-preserving advection is a WENO(order = 5) scalar scheme";
        return BZ_ERR_UNSUPPORTED;
    }
    const dim3 block(64, TYB);
    const int kc = pick_kchunk(g, g.Nz), kcw = pick_kchunk(g, g.Nz - 1);
    const dim3 grid((g.Nx + 63) / 64, (g.Ny + TYB - 1) / TYB, (g.Nz + kc - 1) / kc);
    if (ctx->weno_R != 3) {
        if ((rc = bzi_momentum_tendencies_generic(ctx, s, G))) return rc;
    } else if (ctx->tend_gen >= 2 && ctx->tend_lds) {      // the order-5 kernels bz_compute_tendencies itself would pick
        if ((rc = bzi_u_tendency_lds(ctx, s, G))) return rc;
        if ((rc = bzi_v_tendency_lds(ctx, s, G))) return rc;
        if ((rc = bzi_w_tendency_ring(ctx, s, G))) return rc;
    } else {
        {
            ProfileScope ps(ctx, "x_momentum_tendency");
            hipLaunchKernelGGL(k_u_tendency, grid, block, 0, ctx->stream, g, G->rho_u, s->rho_u, s->rho_v, s->rho_w, s->u, kc, RKEpilogue());
        }
        {
            ProfileScope ps(ctx, "y_momentum_tendency");
            hipLaunchKernelGGL(k_v_tendency, grid, block, 0, ctx->stream, g, G->rho_v, s->rho_u, s->rho_v, s->rho_w, s->v, kc, RKEpilogue());
        }
        if (g.Nz > 1) {
            ProfileScope ps(ctx, "z_momentum_tendency");
            const dim3 gridw(grid.x, grid.y, (g.Nz - 1 + kcw - 1) / kcw);
            hipLaunchKernelGGL(k_w_tendency<true>, gridw, block, 0, ctx->stream, g, G->rho_w, s->rho_u, s->rho_v, s->rho_w, s->w, s->T, s->q, kcw);
        }
    }
    auto scalar = [&](const char *name, double *Gc, const double *c) -> int {
        ProfileScope ps(ctx, name);
        if (ctx->scalar_R != 3) return bzi_scalar_tendency_generic(ctx, Gc, s->u, s->v, s->w, c);
        hipLaunchKernelGGL(k_scalar_tendency, grid, block, 0, ctx->stream, g, Gc, s->u, s->v, s->w, c, kc);
        return BZ_OK;
    };
    if (ctx->scalar_R == 3 && ctx->tend_gen >= 2) {      // theta and moisture together in the LDS-tiled pair kernel, as in the single-order path
        if ((rc = bzi_scalar_pair_tendency(ctx, s, G))) return rc;
    } else {
        if ((rc = scalar("potential_temperature_tendency", G->rho_theta, s->theta))) return rc;
        if ((rc = scalar("moisture_tendency", G->rho_q, s->q))) return rc;
    }
    if (g.microphysics == 2) {
        if ((rc = scalar("kessler_species_tendencies", ctx->kessler.G_cloud_liquid_density, ctx->kessler.cloud_liquid_mass_fraction))) return rc;
        if ((rc = scalar("kessler_species_tendencies", ctx->kessler.G_rain_density, ctx->kessler.rain_mass_fraction))) return rc;
    }
    for (int t = 0; t < ctx->n_tracers; ++t)
        if ((rc = scalar("tracer_tendencies", ctx->tracers[t].G, ctx->tracers[t].specific))) return rc;
    if (g.formulation == 1) {
        ProfileScope ps(ctx, "static_energy_buoyancy_flux");
        hipLaunchKernelGGL(k_energy_buoyancy_flux, dim3((g.Nx + 255) / 256, g.Ny, g.Nz), dim3(256), 0, ctx->stream, g, G->rho_theta, s->w, s->T, s->q);
    }
    if (ctx->bounded_mask && (rc = bzi_bounded_tendencies(ctx, s, G))) return rc;
    if (ctx->has_closure && (rc = bzi_apply_closure(ctx, s, G->rho_u, G->rho_v, G->rho_w, G->rho_theta, G->rho_q, 1.0))) return rc;
    if (ctx->has_forcings && (rc = bzi_apply_forcings(ctx, s, G->rho_u, G->rho_v, G->rho_theta, G->rho_q, 1.0))) return rc;
    if ((rc = bzi_apply_relaxation(ctx, s, G))) return rc;
    BZ_LAUNCH_CHECK();
    return BZ_OK;
}

// the scalars' scheme where it differs from the momentum scheme the context was created with (include/breeze_hip.h)
extern "C" int bz_set_scalar_advection_order(bz_ctx *ctx, int order)
{
    if (!ctx) return BZ_ERR_INVALID;
    ++ctx->config_epoch;
#ifdef BZ_CENTERED2
    ctx->last_error = "bz_set_scalar_advection_order: this build is Centered(order = 2)";
    return BZ_ERR_UNSUPPORTED;
#else
    const DevGrid &g = ctx->dg;
    if (order != 5 && order != 7 && order != 9) { ctx->last_error = "bz_set_scalar_advection_order: WENO order 5, 7 or 9"; return BZ_ERR_UNSUPPORTED; }
    const int R = (order + 1) / 2;
    if (g.Hx < R || (!g.flat_y && g.Hy < R) || g.Hz < R) { ctx->last_error = "bz_set_scalar_advection_order: halos narrower than the scheme"; return BZ_ERR_UNSUPPORTED; }
    if (R != ctx->weno_R && ctx->compressible) {
        ctx->last_error = "bz_set_scalar_advection_order: a scalar order that differs from the momentum order is implemented for anelastic contexts";
        return BZ_ERR_UNSUPPORTED;
    }
    ctx->scalar_R = R;
    return BZ_OK;
#endif
}

extern "C" int bz_compute_tendencies(bz_ctx *ctx, const bz_state *s, const bz_prognostic *G)
{
    if (!ctx || !s || !G) return BZ_ERR_INVALID;
    { const int rcs = bzi_refresh_diagnostics(ctx, s, "bz_compute_tendencies"); if (rcs) return rcs; }
    ctx->G_is_predictor = false;
    const DevGrid &g = ctx->dg;
    dim3 block(64, TYB);
    if (ctx->scalar_R != ctx->weno_R) return compute_tendencies_mixed(ctx, s, G);
    if (ctx->weno_R != 3) {      // WENO(order = 7 / 9): generic kernels for the five prognostic fields, then the order-independent terms
        if (ctx->bounded_mask) {
            ctx->last_error = "WENO(order = 7 / 9) does not implement bounds-preserving advection";
            return BZ_ERR_UNSUPPORTED;
        }
        int rcg = bzi_compute_tendencies_generic(ctx, s, G);
        if (rcg) return rcg;
        if (g.microphysics == 2) {      // Kessler condensate species (bzi_kessler_tendencies, order-generic)
            ProfileScope ps(ctx, "kessler_species_tendencies");
            if ((rcg = bzi_scalar_tendency_generic(ctx, ctx->kessler.G_cloud_liquid_density, s->u, s->v, s->w, ctx->kessler.cloud_liquid_mass_fraction))) return rcg;
            if ((rcg = bzi_scalar_tendency_generic(ctx, ctx->kessler.G_rain_density, s->u, s->v, s->w, ctx->kessler.rain_mass_fraction))) return rcg;
        }
        for (int t = 0; t < ctx->n_tracers; ++t) {
            ProfileScope ps(ctx, "tracer_tendencies");
            if ((rcg = bzi_scalar_tendency_generic(ctx, ctx->tracers[t].G, s->u, s->v, s->w, ctx->tracers[t].specific))) return rcg;
        }
        if (g.formulation == 1) {      // StaticEnergy (examples/dry_thermal_bubble.jl:25): the buoyancy flux term has no advection scheme in it
            ProfileScope ps(ctx, "static_energy_buoyancy_flux");
            hipLaunchKernelGGL(k_energy_buoyancy_flux, dim3((g.Nx + 255) / 256, g.Ny, g.Nz), dim3(256), 0, ctx->stream, g,
                               G->rho_theta, s->w, s->T, s->q);
        }
        if (ctx->has_closure && (rcg = bzi_apply_closure(ctx, s, G->rho_u, G->rho_v, G->rho_w, G->rho_theta, G->rho_q, 1.0))) return rcg;
        if (ctx->has_forcings && (rcg = bzi_apply_forcings(ctx, s, G->rho_u, G->rho_v, G->rho_theta, G->rho_q, 1.0))) return rcg;
        return bzi_apply_relaxation(ctx, s, G);
    }
    if (ctx->tend_gen >= 3 && g.formulation != 0) {
        ctx->last_error = "BZ_TEND_GEN >= 3 implements the potential-temperature formulation only";
        return BZ_ERR_UNSUPPORTED;
    }
    if (ctx->tend_gen >= 3) {
        // gen-3 kernels for the scalars and horizontal momentum; w stays gen-1 unless BZ_TEND_GEN=4
        if (ctx->bounded_mask) { ctx->last_error = "BZ_TEND_GEN >= 3 does not implement bounds-preserving advection"; return BZ_ERR_UNSUPPORTED; }
        int rc = bzi_compute_tendencies3(ctx, s, G, ctx->tend_gen >= 4);
        if (rc || ctx->tend_gen >= 4) return rc;
        ProfileScope ps(ctx, "z_momentum_tendency");
        int kcw = pick_kchunk(g, g.Nz - 1);
        dim3 gridw((g.Nx + 63) / 64, (g.Ny + TYB - 1) / TYB, (g.Nz - 1 + kcw - 1) / kcw);
        hipLaunchKernelGGL(k_w_tendency<true>, gridw, block, 0, ctx->stream, g, G->rho_w, s->rho_u, s->rho_v, s->rho_w, s->w,
                           s->T, s->q, kcw);
        BZ_LAUNCH_CHECK();
        return BZ_OK;
    }
    int kc = pick_kchunk(g, g.Nz);
    dim3 grid((g.Nx + 63) / 64, (g.Ny + TYB - 1) / TYB, (g.Nz + kc - 1) / kc);
    if (ctx->tend_gen >= 2 && ctx->tend_lds) {
        int rc = bzi_u_tendency_lds(ctx, s, G);
        if (rc) return rc;
    } else {
        ProfileScope ps(ctx, "x_momentum_tendency");
        hipLaunchKernelGGL(k_u_tendency, grid, block, 0, ctx->stream, g, G->rho_u, s->rho_u, s->rho_v, s->rho_w, s->u, kc, RKEpilogue());
    }
    if (ctx->tend_gen >= 2 && ctx->tend_lds) {
        int rc = bzi_v_tendency_lds(ctx, s, G);
        if (rc) return rc;
    } else {
        ProfileScope ps(ctx, "y_momentum_tendency");
        hipLaunchKernelGGL(k_v_tendency, grid, block, 0, ctx->stream, g, G->rho_v, s->rho_u, s->rho_v, s->rho_w, s->v, kc, RKEpilogue());
    }
    if (ctx->tend_gen >= 2) {
        int rc = bzi_w_tendency_ring(ctx, s, G);
        if (rc) return rc;
    } else {
        ProfileScope ps(ctx, "z_momentum_tendency");
        int kcw = pick_kchunk(g, g.Nz - 1);
        dim3 gridw(grid.x, grid.y, (g.Nz - 1 + kcw - 1) / kcw);
        hipLaunchKernelGGL(k_w_tendency<true>, gridw, block, 0, ctx->stream, g, G->rho_w, s->rho_u, s->rho_v, s->rho_w, s->w,
                           s->T, s->q, kcw);
    }
    if (ctx->tend_gen >= 2) {
        int rc = bzi_scalar_pair_tendency(ctx, s, G);
        if (rc) return rc;
    } else {
        {
            ProfileScope ps(ctx, "potential_temperature_tendency");
            hipLaunchKernelGGL(k_scalar_tendency, grid, block, 0, ctx->stream, g, G->rho_theta, s->u, s->v, s->w, s->theta, kc);
        }
        {
            ProfileScope ps(ctx, "moisture_tendency");
            hipLaunchKernelGGL(k_scalar_tendency, grid, block, 0, ctx->stream, g, G->rho_q, s->u, s->v, s->w, s->q, kc);
        }
    }
    if (g.microphysics == 2) {
        int rck = bzi_kessler_tendencies(ctx, s);
        if (rck) return rck;
    }
    if (g.formulation == 1) {
        ProfileScope ps(ctx, "static_energy_buoyancy_flux");
        hipLaunchKernelGGL(k_energy_buoyancy_flux, dim3((g.Nx + 255) / 256, g.Ny, g.Nz), dim3(256), 0, ctx->stream, g,
                           G->rho_theta, s->w, s->T, s->q);
    }
    if (ctx->n_tracers) {
        int rct = bzi_tracer_tendencies(ctx, s);
        if (rct) return rct;
    }
    if (ctx->bounded_mask) {
        int rcb = bzi_bounded_tendencies(ctx, s, G);
        if (rcb) return rcb;
    }
    if (ctx->has_closure) {
        int rcc = bzi_apply_closure(ctx, s, G->rho_u, G->rho_v, G->rho_w, G->rho_theta, G->rho_q, 1.0);
        if (rcc) return rcc;
    }
    if (ctx->has_forcings) {
        int rcf = bzi_apply_forcings(ctx, s, G->rho_u, G->rho_v, G->rho_theta, G->rho_q, 1.0);
        if (rcf) return rcf;
    }
    {
        int rcr = bzi_apply_relaxation(ctx, s, G);
        if (rcr) return rcr;
    }
    BZ_LAUNCH_CHECK();
    return BZ_OK;
}

// compute_tendencies! with the next ssp_rk3_substep! folded into the kernels' store (whole-step seam):
//   G.rho_u/v/w <- predictor momentum,  rho_theta, rho_q updated in place;  stage 1 also fills U0.
int bzi_tendencies_fused_rk(bz_ctx *ctx, const bz_state *s, const bz_prognostic *U0, const bz_prognostic *G, double dt,
                            double alpha, bool first)
{
    if (ctx->weno_R != 3) return bzi_generic_tendencies_fused_rk(ctx, s, U0, G, dt, alpha, first);      // WENO 7 / 9: bz_tendency_generic.hip
    const DevGrid &g = ctx->dg;
    dim3 block(64, TYB);
    int kc = pick_kchunk(g, g.Nz);
    dim3 grid((g.Nx + 63) / 64, (g.Ny + TYB - 1) / TYB, (g.Nz + kc - 1) / kc);
    RKEpilogue E;
    E.mode = first ? 1 : 2; E.dt = dt; E.alpha = alpha; E.oma = 1.0 - alpha;
    if (ctx->tend_lds) {
        int rcu = bzi_u_tendency_lds(ctx, s, G, U0, &E);
        if (rcu) return rcu;
    } else {
        ProfileScope ps(ctx, "x_momentum_tendency+rk3");
        E.u0 = U0->rho_u; E.u0_out = U0->rho_u;
        hipLaunchKernelGGL(k_u_tendency, grid, block, 0, ctx->stream, g, G->rho_u, s->rho_u, s->rho_v, s->rho_w, s->u, kc, E);
    }
    if (ctx->tend_lds) {
        int rcv = bzi_v_tendency_lds(ctx, s, G, U0, &E);
        if (rcv) return rcv;
    } else {
        ProfileScope ps(ctx, "y_momentum_tendency+rk3");
        E.u0 = U0->rho_v; E.u0_out = U0->rho_v;
        hipLaunchKernelGGL(k_v_tendency, grid, block, 0, ctx->stream, g, G->rho_v, s->rho_u, s->rho_v, s->rho_w, s->v, kc, E);
    }
    int rc = bzi_w_tendency_ring(ctx, s, G, U0, &E);
    if (rc) return rc;
    rc = bzi_scalar_pair_tendency(ctx, s, G, U0, &E);
    if (rc) return rc;
    BZ_LAUNCH_CHECK();
    return BZ_OK;
}

// one scalar: Gc = -div_rhoUc(c) (include/breeze_hip.h)
extern "C" int bz_compute_scalar_tendency(bz_ctx *ctx, const double *u, const double *v, const double *w, const double *c, double *Gc)
{
    if (!ctx || !u || !v || !w || !c || !Gc) return BZ_ERR_INVALID;
    if (ctx->compressible) { ctx->last_error = "bz_compute_scalar_tendency: anelastic contexts (the compressible model: bz_compute_moisture_tendency)"; return BZ_ERR_UNSUPPORTED; }
    ProfileScope ps(ctx, "scalar_tendency");
    if (ctx->scalar_R != 3) return bzi_scalar_tendency_generic(ctx, Gc, u, v, w, c);
    const DevGrid &g = ctx->dg;
    dim3 block(64, TYB);
    const int kc = pick_kchunk(g, g.Nz);
    dim3 grid((g.Nx + 63) / 64, (g.Ny + TYB - 1) / TYB, (g.Nz + kc - 1) / kc);
    hipLaunchKernelGGL(k_scalar_tendency, grid, block, 0, ctx->stream, g, Gc, u, v, w, c, kc);
    BZ_LAUNCH_CHECK();
    return BZ_OK;
}

// scalar tendencies of the user tracers: G = -div_rhoUc(c)  (update_atmosphere_model_state.jl:352-372)
int bzi_tracer_tendencies(bz_ctx *ctx, const bz_state *s)
{
    const DevGrid &g = ctx->dg;
    ProfileScope ps(ctx, "tracer_tendencies");
    dim3 block(64, TYB);
    int kc = pick_kchunk(g, g.Nz);
    dim3 grid((g.Nx + 63) / 64, (g.Ny + TYB - 1) / TYB, (g.Nz + kc - 1) / kc);
    for (int t = 0; t < ctx->n_tracers; ++t)
        hipLaunchKernelGGL(k_scalar_tendency, grid, block, 0, ctx->stream, g, ctx->tracers[t].G, s->u, s->v, s->w,
                           ctx->tracers[t].specific, kc);
    BZ_LAUNCH_CHECK();
    return BZ_OK;
}

// scalar tendencies of the Kessler condensate species: G = -div_rhoUc(q^cl), -div_rhoUc(q^r)
// (update_atmosphere_model_state.jl:352-372 with prognostic_field_names(::DCMIP2016KM), dcmip2016_kessler.jl:216)
int bzi_kessler_tendencies(bz_ctx *ctx, const bz_state *s)
{
    const DevGrid &g = ctx->dg;
    ProfileScope ps(ctx, "kessler_species_tendencies");
    dim3 block(64, TYB);
    int kc = pick_kchunk(g, g.Nz);
    dim3 grid((g.Nx + 63) / 64, (g.Ny + TYB - 1) / TYB, (g.Nz + kc - 1) / kc);
    hipLaunchKernelGGL(k_scalar_tendency, grid, block, 0, ctx->stream, g, ctx->kessler.G_cloud_liquid_density, s->u, s->v, s->w,
                       ctx->kessler.cloud_liquid_mass_fraction, kc);
    hipLaunchKernelGGL(k_scalar_tendency, grid, block, 0, ctx->stream, g, ctx->kessler.G_rain_density, s->u, s->v, s->w,
                       ctx->kessler.rain_mass_fraction, kc);
    BZ_LAUNCH_CHECK();
    return BZ_OK;
}
