// bz_state.hip — streaming kernels of the anelastic step: velocity / thermodynamic diagnosis,
// SSP-RK3 stage update, pressure projection.
#include "bz_internal.h"

#define TX 64
#define TY 4

// ---- _compute_velocities! (/root/reference/src/AtmosphereModels/update_atmosphere_model_state.jl:248-254)
// launch covers k = 0..Nz (Face length on Bounded z, :138-145).
__global__ __launch_bounds__(TX *TY) void k_velocities(DevGrid g, double *__restrict__ u,
                                                      double *__restrict__ v, double *__restrict__ w,
                                                      const double *__restrict__ ru,
                                                      const double *__restrict__ rv,
                                                      const double *__restrict__ rw)
{
    int i = blockIdx.x * TX + threadIdx.x, j = blockIdx.y * TY + threadIdx.y, k = blockIdx.z;
    if (i >= g.Nx || j >= g.Ny) return;
    long long n = g.idx(i, j, k);
    double rc = g.rho[k], rf = g.rho_f[k];
    u[n] = ru[n] / rc;
    v[n] = rv[n] / rc;
    w[n] = rw[n] / rf;
}

// ---- _compute_auxiliary_thermodynamic_variables! (:256-292) with
// potential_temperature_formulation.jl:115-145 and Thermodynamics/dynamic_states.jl:31-58.
__global__ __launch_bounds__(TX *TY) void k_thermo(DevGrid g, double *__restrict__ theta,
                                                  double *__restrict__ qv, double *__restrict__ T,
                                                  const double *__restrict__ rtheta,
                                                  const double *__restrict__ rq)
{
    int i = blockIdx.x * TX + threadIdx.x, j = blockIdx.y * TY + threadIdx.y, k = blockIdx.z;
    if (i >= g.Nx || j >= g.Ny) return;
    long long n = g.idx(i, j, k);
    double rho = g.rho[k];
    double th = rtheta[n] / rho;
    double q = rq[n] / rho;
    theta[n] = th;
    qv[n] = q;
    double qd = 1.0 - q;
    double Rm = qd * g.Rd + q * g.Rv;
    double cpm = qd * g.cpd + q * g.cpv;
    if (g.formulation == 1) {       // StaticEnergyState: T = (e - g z) / c_pm  (dynamic_states.jl:283-298)
        T[n] = (th - g.g * g.zc[k]) / cpm;
        return;
    }
    if (g.microphysics == 2) {      // DCMIP2016 Kessler: moisture fractions from the prognostic condensate densities
        // reference behaviour (update_atmosphere_model_state.jl:276-291): grid_moisture_fractions reads the diagnostic
        // mu.q^cl, mu.q^r *before* update_microphysical_auxiliaries! refreshes them, so T carries the previous condensate
        T[n] = bz_kessler_T(g, th, q, g.qcl_field[n] + g.qr_field[n], g.p_r[k]);
        const double qcl = g.rqcl_field[n] / rho, qr = g.rqr_field[n] / rho;
        g.qcl_field[n] = qcl;
        g.qr_field[n] = qr;
        g.qv_field[n] = q;
        return;
    }
    if (g.microphysics == 1) {      // maybe_adjust_thermodynamic_state(SaturationAdjustment) + update_microphysical_fields!
        double qvv, qll;
        T[n] = bz_sa_diagnose(g, th, q, g.p_r[k], qvv, qll);
        g.qv_field[n] = qvv;
        g.ql_field[n] = qll;
        return;
    }
    T[n] = bz_exner_factor(g, k, q, cpm) * th;
}

struct RKFields {
    double *u[5];
    const double *u0[5];
    const double *G[5];
};

// ---- _ssp_rk3_substep! (/root/reference/src/TimeSteppers/ssp_runge_kutta_3.jl:167-173), all five
// prognostic fields in one pass.  Field 2 is rho_w: wall faces (k = 0, Nz) are never updated.
__global__ __launch_bounds__(TX *TY) void k_rk3_substep(DevGrid g, RKFields F, double dt, double alpha)
{
    int i = blockIdx.x * TX + threadIdx.x, j = blockIdx.y * TY + threadIdx.y, k = blockIdx.z;
    if (i >= g.Nx || j >= g.Ny) return;
    long long n = g.idx(i, j, k);
    double oma = 1.0 - alpha;
#pragma unroll
    for (int f = 0; f < 5; ++f) {
        if (f == 2 && k == 0) continue;
        F.u[f][n] = oma * F.u0[f][n] + alpha * (F.u[f][n] + dt * F.G[f][n]);
    }
}

// ---- _pressure_correct_momentum! (/root/reference/src/AnelasticEquations/anelastic_time_stepping.jl:45-54)
__global__ __launch_bounds__(TX *TY) void k_pressure_correct(DevGrid g, double *__restrict__ ru,
                                                            double *__restrict__ rv,
                                                            double *__restrict__ rw,
                                                            const double *__restrict__ phi, double dt)
{
    int i = blockIdx.x * TX + threadIdx.x, j = blockIdx.y * TY + threadIdx.y, k = blockIdx.z;
    if (i >= g.Nx || j >= g.Ny) return;
    long long n = g.idx(i, j, k);
    double rc = g.rho[k], rf = g.rho_f[k];
    double p = phi[n];
    if (!(g.bounded_x && i == 0)) ru[n] -= rc * dt * ((p - phi[n - 1]) * g.rdx);      // a wall face is never corrected
    if (!g.flat_y && !(g.bounded_y && j == 0)) rv[n] -= rc * dt * ((p - phi[n - g.Sx]) * g.rdy);      // a wall face is never corrected
    rw[n] -= rf * dt * ((p - phi[n - g.Sxy]) * g.rdzf[k]);
}

__global__ __launch_bounds__(256) void k_copy5(long long count, double *d0, const double *s0, double *d1,
                                               const double *s1, double *d2, const double *s2,
                                               double *d3, const double *s3, long long count_w,
                                               double *dw, const double *sw)
{
    long long n = (long long)blockIdx.x * 256 + threadIdx.x;
    long long stride = (long long)gridDim.x * 256;
    for (; n < count_w; n += stride) {
        if (n < count) { d0[n] = s0[n]; d1[n] = s1[n]; d2[n] = s2[n]; d3[n] = s3[n]; }
        dw[n] = sw[n];
    }
}

__global__ __launch_bounds__(TX *TY) void k_max_abs_div(DevGrid g, const double *__restrict__ ru,
                                                       const double *__restrict__ rv,
                                                       const double *__restrict__ rw,
                                                       bz_bits_t *out)
{
    int i = blockIdx.x * TX + threadIdx.x, j = blockIdx.y * TY + threadIdx.y, k = blockIdx.z;
    double d = 0.0;
    if (i < g.Nx && j < g.Ny) {
        long long n = g.idx(i, j, k);
        double Ax = g.Ax[k], Ay = g.Ay[k], Az = g.Az;
        double a = Ax * ru[n + 1] - Ax * ru[n];
        double b = g.flat_y ? 0.0 : Ay * rv[n + g.Sx] - Ay * rv[n];
        double c = Az * rw[n + g.Sxy] - Az * rw[n];
        d = fabs(g.Vinv_c[k] * (a + b + c));
    }
    for (int o = 32; o > 0; o >>= 1) d = fmax(d, __shfl_down(d, o));
    if ((threadIdx.x & 63) == 0) atomicMax(out, bz_real_bits(d));
}

static inline dim3 cell_grid(const DevGrid &g, int nz) { return dim3((g.Nx + TX - 1) / TX, (g.Ny + TY - 1) / TY, nz); }

extern "C" int bz_compute_velocities(bz_ctx *ctx, const bz_state *s)
{
    if (!ctx || !s) return BZ_ERR_INVALID;
    const DevGrid &g = ctx->dg;
    // halos of momentum first (update_atmosphere_model_state.jl:135-136)
    double *mf[3] = {s->rho_u, s->rho_v, s->rho_w};
    int mk[3] = {BZ_HALO_XFACE, BZ_HALO_YFACE, 1};
    int rc = bzi_fill_halos_multi(ctx, mf, mk, 3);
    if (rc) return rc;
    {
        ProfileScope ps(ctx, "compute_velocities");
        hipLaunchKernelGGL(k_velocities, cell_grid(g, g.Nz + 1), dim3(TX, TY), 0, ctx->stream, g, s->u, s->v,
                           s->w, s->rho_u, s->rho_v, s->rho_w);
        BZ_LAUNCH_CHECK();
    }
    double *vf[3] = {s->u, s->v, s->w};
    int vk[3] = {2 | BZ_HALO_XFACE, 2 | BZ_HALO_YFACE, 3};
    return bzi_fill_halos_multi(ctx, vf, vk, 3);
}

extern "C" int bz_compute_auxiliary_thermodynamic_variables(bz_ctx *ctx, const bz_state *s)
{
    if (!ctx || !s) return BZ_ERR_INVALID;
    const DevGrid &g = ctx->dg;
    {
        ProfileScope ps(ctx, "compute_auxiliary_thermodynamic_variables");
        hipLaunchKernelGGL(k_thermo, cell_grid(g, g.Nz), dim3(TX, TY), 0, ctx->stream, g, s->theta, s->q, s->T,
                           s->rho_theta, s->rho_q);
        BZ_LAUNCH_CHECK();
    }
    double *f[6] = {s->T, s->q, s->theta, g.qv_field, g.microphysics == 2 ? g.qcl_field : g.ql_field, g.qr_field};
    int kd[6] = {0, 0, 0, 0, 0, 0};
    int rch = bzi_fill_halos_multi(ctx, f, kd, g.microphysics == 2 ? 6 : (g.microphysics ? 5 : 3));
    if (rch) return rch;
    return bzi_tracer_specific(ctx);       // tracer_density_to_specific! (update_atmosphere_model_state.jl:43)
}

extern "C" int bz_store_initial_state(bz_ctx *ctx, const bz_state *s, const bz_prognostic *U0)
{
    if (!ctx || !s || !U0) return BZ_ERR_INVALID;
    const DevGrid &g = ctx->dg;
    ProfileScope ps(ctx, "store_initial_state");
    long long nc = g.Sxy * (g.Nz + 2 * g.Hz), nw = g.Sxy * (g.Nz + 1 + 2 * g.Hz);
    hipLaunchKernelGGL(k_copy5, dim3(256 * 32), dim3(256), 0, ctx->stream, nc, U0->rho_u, s->rho_u, U0->rho_v,
                       s->rho_v, U0->rho_theta, s->rho_theta, U0->rho_q, s->rho_q, nw, U0->rho_w, s->rho_w);
    BZ_LAUNCH_CHECK();
    if (ctx->n_tracers) {
        int rct = bzi_tracer_store_initial_state(ctx);
        if (rct) return rct;
    }
    if (g.microphysics == 2) {
        BZ_HIP(hipMemcpyAsync(ctx->kessler.U0_cloud_liquid_density, ctx->kessler.cloud_liquid_density, nc * sizeof(double), hipMemcpyDeviceToDevice, ctx->stream));
        BZ_HIP(hipMemcpyAsync(ctx->kessler.U0_rain_density, ctx->kessler.rain_density, nc * sizeof(double), hipMemcpyDeviceToDevice, ctx->stream));
    }
    return BZ_OK;
}

extern "C" int bz_ssp_rk3_substep(bz_ctx *ctx, const bz_state *s, const bz_prognostic *U0,
                                  const bz_prognostic *G, double dt, double alpha)
{
    if (!ctx || !s || !U0 || !G) return BZ_ERR_INVALID;
    if (ctx->G_is_predictor) {      // a fused whole step left predictor momentum in G: rebuild the tendencies first
        int rc = bz_compute_tendencies(ctx, s, G);
        if (rc) return rc;
    }
    const DevGrid &g = ctx->dg;
    ProfileScope ps(ctx, "ssp_rk3_substep");
    RKFields F;
    F.u[0] = s->rho_u; F.u[1] = s->rho_v; F.u[2] = s->rho_w; F.u[3] = s->rho_theta; F.u[4] = s->rho_q;
    F.u0[0] = U0->rho_u; F.u0[1] = U0->rho_v; F.u0[2] = U0->rho_w; F.u0[3] = U0->rho_theta; F.u0[4] = U0->rho_q;
    F.G[0] = G->rho_u; F.G[1] = G->rho_v; F.G[2] = G->rho_w; F.G[3] = G->rho_theta; F.G[4] = G->rho_q;
    hipLaunchKernelGGL(k_rk3_substep, cell_grid(g, g.Nz), dim3(TX, TY), 0, ctx->stream, g, F, dt, alpha);
    BZ_LAUNCH_CHECK();
    if (ctx->n_tracers) {
        int rct = bzi_tracer_rk3(ctx, dt, alpha, false);
        if (rct) return rct;
    }
    if (g.microphysics == 2) return bzi_kessler_rk3(ctx, dt, alpha, false);
    return BZ_OK;
}

extern "C" int bz_make_pressure_correction(bz_ctx *ctx, const bz_state *s, double dt)
{
    if (!ctx || !s) return BZ_ERR_INVALID;
    const DevGrid &g = ctx->dg;
    ProfileScope ps(ctx, "make_pressure_correction");
    hipLaunchKernelGGL(k_pressure_correct, cell_grid(g, g.Nz), dim3(TX, TY), 0, ctx->stream, g, s->rho_u,
                       s->rho_v, s->rho_w, s->phi, dt);
    BZ_LAUNCH_CHECK();
    return BZ_OK;
}

extern "C" int bz_max_abs_divergence(bz_ctx *ctx, const bz_state *s, double *out)
{
    if (!ctx || !s || !out) return BZ_ERR_INVALID;
    const DevGrid &g = ctx->dg;
    double *mf[3] = {s->rho_u, s->rho_v, s->rho_w};
    int mk[3] = {BZ_HALO_XFACE, BZ_HALO_YFACE, 1};
    int rc = bzi_fill_halos_multi(ctx, mf, mk, 3);
    if (rc) return rc;
    BZ_HIP(hipMemsetAsync(ctx->d_scalar, 0, sizeof(double), ctx->stream));
    hipLaunchKernelGGL(k_max_abs_div, cell_grid(g, g.Nz), dim3(TX, TY), 0, ctx->stream, g, s->rho_u, s->rho_v,
                       s->rho_w, (bz_bits_t *)ctx->d_scalar);
    BZ_LAUNCH_CHECK();
    BZ_HIP(hipMemcpyAsync(out, ctx->d_scalar, sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    BZ_HIP(hipStreamSynchronize(ctx->stream));
    return BZ_OK;
}

// ---- the run!-loop reductions around the step ---------------------------------------------------
// cell_advection_timescale(model) = minimum over cells of 1 / (|u|/dx + |v|/dy + |w|/dz)
// (/root/reference/src/AtmosphereModels/cell_advection_timescale.jl:47-66 -> Oceananigans.Advection.cell_advection_timescale,
// recalled: spacings at the velocity locations, dz = centre spacing at the w face); the maximum inverse timescale is
// reduced with wave shuffles + one atomicMax per wave on the bit pattern of the non-negative double.
__global__ __launch_bounds__(TX *TY) void k_max_inverse_advection_timescale(DevGrid g, const double *__restrict__ u,
                                                                           const double *__restrict__ v,
                                                                           const double *__restrict__ w, int with_w,
                                                                           bz_bits_t *out)
{
    int i = blockIdx.x * TX + threadIdx.x, j = blockIdx.y * TY + threadIdx.y, k = blockIdx.z;
    double d = 0.0;
    if (i < g.Nx && j < g.Ny) {
        long long n = g.idx(i, j, k);
        const double ix = fabs(u[n]) / g.dx, iy = g.flat_y ? 0.0 : fabs(v[n]) / g.dy;
        const double iz = with_w ? fabs(w[n]) / g.dzf[k] : 0.0;
        d = ix + iy + iz;
        if (d != d) d = bz_real_inf();      // NaN velocity: timescale 0
    }
    for (int o = 32; o > 0; o >>= 1) d = fmax(d, __shfl_down(d, o));
    if ((threadIdx.x & 63) == 0) atomicMax(out, bz_real_bits(d));
}

extern "C" int bz_cell_advection_timescale(bz_ctx *ctx, const double *u, const double *v, const double *w, double *out)
{
    if (!ctx || !u || !v || !out) return BZ_ERR_INVALID;
    { const int rcs = bzi_refresh_diagnostics(ctx, nullptr, "bz_cell_advection_timescale"); if (rcs) return rcs; }      // the velocities are diagnostics
    const DevGrid &g = ctx->dg;
    BZ_HIP(hipMemsetAsync(ctx->d_scalar, 0, sizeof(double), ctx->stream));
    hipLaunchKernelGGL(k_max_inverse_advection_timescale, cell_grid(g, g.Nz), dim3(TX, TY), 0, ctx->stream, g, u, v,
                       w ? w : u, w ? 1 : 0, (bz_bits_t *)ctx->d_scalar);
    BZ_LAUNCH_CHECK();
    double inv = 0.0;
    BZ_HIP(hipMemcpyAsync(&inv, ctx->d_scalar, sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    BZ_HIP(hipStreamSynchronize(ctx->stream));
    *out = 1.0 / inv;
    return BZ_OK;
}

// NaNChecker on one field (default_nan_checker: the first prognostic field, atmosphere_model.jl:561-572): 1 if any
// interior value is NaN.
__global__ __launch_bounds__(256) void k_any_nan(DevGrid g, const double *__restrict__ f, int nlev, int *out)
{
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
    if (t >= (long long)g.Ny * g.Sx) return;
    const int k = blockIdx.y;
    if (k >= nlev) return;
    const long long n = g.Sxy * (k + g.Hz) + (long long)g.Hy * g.Sx + t;
    const int i = (int)(t % g.Sx) - g.Hx;
    const double x = f[n];
    if (i >= 0 && i < g.Nx && x != x) *out = 1;
}

extern "C" int bz_any_nan(bz_ctx *ctx, const double *field, int z_face, int32_t *out)
{
    if (!ctx || !field || !out) return BZ_ERR_INVALID;
    const DevGrid &g = ctx->dg;
    int *flag = (int *)(ctx->d_scalar + 8);
    BZ_HIP(hipMemsetAsync(flag, 0, sizeof(int), ctx->stream));
    const int nlev = g.Nz + (z_face ? 1 : 0);
    const long long per_level = (long long)g.Ny * g.Sx;
    hipLaunchKernelGGL(k_any_nan, dim3((unsigned)((per_level + 255) / 256), nlev), dim3(256), 0, ctx->stream, g, field, nlev, flag);
    BZ_LAUNCH_CHECK();
    int h = 0;
    BZ_HIP(hipMemcpyAsync(&h, flag, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
    BZ_HIP(hipStreamSynchronize(ctx->stream));
    *out = h;
    return BZ_OK;
}
