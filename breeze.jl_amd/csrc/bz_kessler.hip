// bz_kessler.hip — DCMIP2016 Kessler warm-rain microphysics: the operator-split column kernel of
// microphysics_model_update!(::DCMIP2016KesslerMicrophysics, model)
//   /root/reference/src/Microphysics/dcmip2016_kessler.jl:449-486 (launcher), :618-858 (kernel), helpers :397-430,:496-612;
//   Tetens saturation vapour pressure /root/reference/src/Thermodynamics/tetens_formula.jl:112-119.
// One thread per column (x fastest => every level access of a wave is one contiguous 512-byte run); the column marches
// upward with rain-sedimentation subcycling whose count is data dependent per column, exactly as the reference's `:xy`
// kernel.  The microphysical fields q^v, q^cl, q^r, W^r double as the column workspace (mixing ratios while the kernel
// runs, mass fractions on exit), as in the reference.
#include <cmath>

#include "bz_internal.h"

struct KesslerFields {
    const double *density, *pressure;          // 3-D (compressible) or nullptr (anelastic reference columns)
    double *theta, *rho_theta, *rho_qv, *rho_qcl, *rho_qr;
    double *qv, *qcl, *qr, *W;
    double *precipitation_rate;                // (Sx, Sy) horizontal parent
};

__device__ __forceinline__ double ks_terminal_velocity(const bz_kessler_microphysics &M, double rr, double rho, double rho1)
{
    return M.terminal_velocity_coefficient * pow(rr * M.density_scale * rho, M.terminal_velocity_exponent) * sqrt(rho1 / rho);
}
__device__ __forceinline__ double ks_psat(const bz_kessler_microphysics &M, double T)
{
    return M.tetens_reference_saturation_vapor_pressure *
           exp(M.tetens_liquid_coefficient * (T - M.tetens_reference_temperature) / (T - M.tetens_liquid_temperature_offset));
}
__device__ __forceinline__ void ks_fractions(double rv, double rl, double &qv, double &ql)
{
    const double inv = 1.0 / (1.0 + (rv + rl));
    qv = rv * inv;
    ql = rl * inv;
}
__device__ __forceinline__ void ks_mixture(const DevGrid &g, const bz_kessler_microphysics &M, double qv, double ql, double &Rm, double &cpm)
{
    const double qd = 1.0 - (qv + ql);
    Rm = qd * g.Rd + qv * g.Rv;
    cpm = qd * g.cpd + qv * g.cpv + ql * M.liquid_heat_capacity;
}

// step_kessler_microphysics (:519-562)
__device__ __forceinline__ void ks_step(const DevGrid &g, const bz_kessler_microphysics &M, double &rv, double &rcl, double &rr,
                                        double drW, double T, double rho, double p, double dt, double f5, double &drl)
{
    const double A = fmax(0.0, M.autoconversion_rate * (rcl - M.autoconversion_threshold));
    const double denom = 1.0 + dt * M.accretion_rate * pow(rr, M.accretion_exponent);
    const double dP = rcl - (rcl - dt * A) / denom;
    rcl = fmax(0.0, rcl - dP);
    rr = fmax(0.0, rr + dP + drW);
    const double qs = ks_psat(M, T) / (rho * g.Rv * T);
    const double rs = qs / (1.0 - qs);
    const double dTo = T - M.tetens_liquid_temperature_offset;
    const double dsat = (rv - rs) / (1.0 + rs * f5 / (dTo * dTo));
    const double rhok = M.density_scale * rho;
    const double rhorr = rhok * rr;
    const double Vev = (M.evaporation_ventilation_coefficient_1 +
                        M.evaporation_ventilation_coefficient_2 * pow(rhorr, M.evaporation_ventilation_exponent_1)) *
                       pow(rhorr, M.evaporation_ventilation_exponent_2);
    const double Dth = M.diffusivity_coefficient / (p * rs) + M.thermal_conductivity_coefficient;
    const double drs = fmax(0.0, rs - rv);
    const double E = Vev / Dth * drs / (rhok * rs + 1e-20);
    const double dEmax = fmax(0.0, -dsat - rcl);
    const double dE = fmin(fmin(dt * E, dEmax), rr);
    const double dC = fmax(dsat, -rcl);
    rv = fmax(0.0, rv - dC + dE);
    rcl = rcl + dC;
    rr = rr - dE;
    drl = dC - dE;
}

__global__ __launch_bounds__(64) void k_kessler_column(DevGrid g, bz_kessler_microphysics M, KesslerFields F, double dt, double pst)
{
    const int i = blockIdx.x * 64 + threadIdx.x, j = blockIdx.y;
    if (i >= g.Nx) return;
    const long long sz = g.Sxy;
    const long long n0 = g.idx(i, j, 0);
    const int Nz = g.Nz;
    const double f5 = M.tetens_liquid_coefficient * M.dcmip_temperature_scale * M.liquid_latent_heat / g.cpd;
    const double Ll = M.liquid_latent_heat;
#define KS_RHO(k, n) (F.density ? F.density[n] : g.rho[k])
#define KS_P(k, n) (F.pressure ? F.pressure[n] : g.p_r[k])
    const double rho1 = KS_RHO(0, n0);
    double max_dt = dt;
    for (int k = 0; k < Nz; ++k) {
        const long long n = n0 + sz * k;
        const double rho = KS_RHO(k, n);
        double qv = F.rho_qv[n] / rho;
        const double qcl = fmax(0.0, F.rho_qcl[n] / rho), qr = fmax(0.0, F.rho_qr[n] / rho);
        qv = fmax(0.0, qv);
        const double ql = qcl + qr;
        const double inv_qd = 1.0 / (1.0 - (qv + ql));
        const double rv = qv * inv_qd;
        const double rt = rv + ql * inv_qd;
        const double rcl = qcl * (1.0 + rt), rr = qr * (1.0 + rt);
        const double W = ks_terminal_velocity(M, rr, rho, rho1);
        F.W[n] = W;
        F.qv[n] = rv;
        F.qcl[n] = rcl;
        F.qr[n] = rr;
        if (k < Nz - 1) max_dt = fmin(max_dt, M.substep_cfl * (g.zc[k + 1] - g.zc[k]) / W);
    }
    const int Ns = max(1, (int)ceil(dt / max_dt));
    const double inv_Ns = 1.0 / (double)Ns;
    const double dts = dt * inv_Ns;
    double Psurf = 0.0;
    for (int m = 1; m <= Ns; ++m) {
        {
            const double rv1 = F.qv[n0], rcl1 = F.qcl[n0], rr1 = F.qr[n0];
            Psurf += rr1 / (1.0 + (rv1 + rcl1 + rr1)) * F.W[n0];
        }
        for (int k = 0; k < Nz; ++k) {
            const long long n = n0 + sz * k;
            const double rho = KS_RHO(k, n), p = KS_P(k, n);
            const double th = F.theta[n];
            double rv = F.qv[n], rcl = F.qcl[n], rr = F.qr[n];
            double qv, ql, Rm, cpm;
            ks_fractions(rv, rcl + rr, qv, ql);
            ks_mixture(g, M, qv, ql, Rm, cpm);
            const double Tk = pow(p / pst, Rm / cpm) * th + Ll * ql / cpm;
            const double rhok = M.density_scale * rho;
            double drW;
            if (k < Nz - 1) {
                const double dz = g.zc[k + 1] - g.zc[k];
                const double rhok1 = M.density_scale * KS_RHO(k + 1, n + sz);
                drW = dts * (rhok1 * F.qr[n + sz] * F.W[n + sz] - rhok * rr * F.W[n]) / (rhok * dz);
            } else {
                const double dz_half = (g.zc[k] - g.zc[k - 1]) / 2.0;
                drW = -dts * rr * F.W[n] / dz_half;
            }
            double drl;
            ks_step(g, M, rv, rcl, rr, drW, Tk, rho, p, dts, f5, drl);
            F.qv[n] = rv;
            F.qcl[n] = rcl;
            F.qr[n] = rr;
            const double T = Tk + Ll / g.cpd * drl;
            ks_fractions(rv, rcl + rr, qv, ql);
            ks_mixture(g, M, qv, ql, Rm, cpm);
            const double thn = (T - Ll * ql / cpm) / pow(p / pst, Rm / cpm);
            F.theta[n] = thn;
            F.rho_theta[n] = rho * thn;
        }
        if (m < Ns)
            for (int k = 0; k < Nz; ++k) {
                const long long n = n0 + sz * k;
                F.W[n] = ks_terminal_velocity(M, F.qr[n], KS_RHO(k, n), rho1);
            }
    }
    F.precipitation_rate[(long long)(i + g.Hx) + (long long)g.Sx * (j + g.Hy)] = Psurf * inv_Ns;
    for (int k = 0; k < Nz; ++k) {
        const long long n = n0 + sz * k;
        const double rho = KS_RHO(k, n);
        const double rv = F.qv[n], rcl = F.qcl[n], rr = F.qr[n];
        double qv, ql;
        ks_fractions(rv, rcl + rr, qv, ql);
        const double rt = rv + (rcl + rr);
        const double qcl = rcl / (1.0 + rt), qr = rr / (1.0 + rt);
        F.rho_qv[n] = rho * qv;
        F.rho_qcl[n] = rho * qcl;
        F.rho_qr[n] = rho * qr;
        F.qv[n] = qv;
        F.qcl[n] = qcl;
        F.qr[n] = qr;
    }
#undef KS_RHO
#undef KS_P
}

extern "C" int bz_kessler_microphysics_update(bz_ctx *ctx, const bz_kessler_microphysics *params, const bz_kessler_fields *f,
                                              double dt, double standard_pressure)
{
    if (!ctx || !params || !f) return BZ_ERR_INVALID;
    if (!f->potential_temperature || !f->potential_temperature_density || !f->moisture_density || !f->cloud_liquid_density ||
        !f->rain_density || !f->vapor_mass_fraction || !f->cloud_liquid_mass_fraction || !f->rain_mass_fraction ||
        !f->rain_terminal_velocity || !f->precipitation_rate)
        return BZ_ERR_INVALID;
    if (std::isnan(dt) || std::isinf(dt) || dt <= 0.0) return BZ_OK;        // (:455)
    const DevGrid &g = ctx->dg;
    KesslerFields F;
    F.density = f->density; F.pressure = f->pressure;
    F.theta = f->potential_temperature; F.rho_theta = f->potential_temperature_density;
    F.rho_qv = f->moisture_density; F.rho_qcl = f->cloud_liquid_density; F.rho_qr = f->rain_density;
    F.qv = f->vapor_mass_fraction; F.qcl = f->cloud_liquid_mass_fraction; F.qr = f->rain_mass_fraction;
    F.W = f->rain_terminal_velocity; F.precipitation_rate = f->precipitation_rate;
    ProfileScope ps(ctx, "kessler_microphysics_update");
    hipLaunchKernelGGL(k_kessler_column, dim3((g.Nx + 63) / 64, g.Ny), dim3(64), 0, ctx->stream, g, *params, F, dt, standard_pressure);
    BZ_LAUNCH_CHECK();
    return BZ_OK;
}

// ---------------------------------------------------------------------------------------------------------------------
// DCMIP2016KesslerMicrophysics attached to the anelastic AtmosphereModel: the two condensate species become prognostic
// (prognostic_field_names = (:rho q^cl, :rho q^r), dcmip2016_kessler.jl:216), the temperature diagnosis and the buoyancy use
// the moisture fractions (q^v, q^cl + q^r) (:298-303), and the column update closes every time step (:449-486).
// ---------------------------------------------------------------------------------------------------------------------
extern "C" int bz_set_kessler_microphysics(bz_ctx *ctx, const bz_kessler_microphysics *params, const bz_kessler_model_fields *f,
                                           double standard_pressure)
{
    if (ctx) ++ctx->config_epoch;      // captured steps (bz_graph.hip) belong to one configuration
    if (!ctx) return BZ_ERR_INVALID;
    DevGrid &g = ctx->dg;
    if (!params) {
        if (g.microphysics == 2) g.microphysics = 0;
        return BZ_OK;
    }
    if (!f || !f->cloud_liquid_density || !f->rain_density || !f->U0_cloud_liquid_density || !f->U0_rain_density ||
        !f->G_cloud_liquid_density || !f->G_rain_density || !f->vapor_mass_fraction || !f->cloud_liquid_mass_fraction ||
        !f->rain_mass_fraction || !f->rain_terminal_velocity || !f->precipitation_rate)
        return BZ_ERR_INVALID;
    if (g.formulation != 0) {
        ctx->last_error = "Kessler microphysics is attached to potential-temperature models";
        return BZ_ERR_UNSUPPORTED;
    }
    ctx->kessler_params = *params;
    ctx->kessler = *f;
    ctx->kessler_pst = standard_pressure;
    g.microphysics = 2;
    g.sa_Ll = params->liquid_latent_heat;
    g.sa_cl = params->liquid_heat_capacity;
    g.qv_field = f->vapor_mass_fraction;
    g.ql_field = nullptr;
    g.rqcl_field = f->cloud_liquid_density;
    g.rqr_field = f->rain_density;
    g.qcl_field = f->cloud_liquid_mass_fraction;
    g.qr_field = f->rain_mass_fraction;
    return BZ_OK;
}

// ssp_rk3_substep! of the two condensate species [+ store_initial_state! when first] (ssp_runge_kutta_3.jl:114-186)
template <bool FIRST>
__global__ __launch_bounds__(256) void k_rk3_two(DevGrid g, double *__restrict__ a, double *__restrict__ a0, const double *__restrict__ Ga,
                                                 double *__restrict__ b, double *__restrict__ b0, const double *__restrict__ Gb,
                                                 double dt, double alpha)
{
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
    if (t >= (long long)g.Ny * g.Sx) return;
    const long long n = g.Sxy * ((long long)blockIdx.y + g.Hz) + (long long)g.Hy * g.Sx + t;
    const double oma = 1.0 - alpha;
    const double ua = a[n], ub = b[n];
    if (FIRST) {
        a0[n] = ua;
        b0[n] = ub;
        a[n] = oma * ua + alpha * (ua + dt * Ga[n]);
        b[n] = oma * ub + alpha * (ub + dt * Gb[n]);
    } else {
        a[n] = oma * a0[n] + alpha * (ua + dt * Ga[n]);
        b[n] = oma * b0[n] + alpha * (ub + dt * Gb[n]);
    }
}

int bzi_kessler_rk3(bz_ctx *ctx, double dt, double alpha, bool first)
{
    const DevGrid &g = ctx->dg;
    const bz_kessler_model_fields &K = ctx->kessler;
    ProfileScope ps(ctx, "kessler_species_rk3");
    const long long per_level = (long long)g.Ny * g.Sx;
    dim3 grid((unsigned)((per_level + 255) / 256), g.Nz), block(256);
    if (first)
        hipLaunchKernelGGL(k_rk3_two<true>, grid, block, 0, ctx->stream, g, K.cloud_liquid_density, K.U0_cloud_liquid_density,
                           K.G_cloud_liquid_density, K.rain_density, K.U0_rain_density, K.G_rain_density, dt, alpha);
    else
        hipLaunchKernelGGL(k_rk3_two<false>, grid, block, 0, ctx->stream, g, K.cloud_liquid_density, K.U0_cloud_liquid_density,
                           K.G_cloud_liquid_density, K.rain_density, K.U0_rain_density, K.G_rain_density, dt, alpha);
    BZ_LAUNCH_CHECK();
    return BZ_OK;
}

// microphysics_model_update!: the column kernel, then update_state!(model) (dcmip2016_kessler.jl:480-485)
int bzi_kessler_update(bz_ctx *ctx, const bz_state *s, const bz_prognostic *G, double dt)
{
    const bz_kessler_model_fields &K = ctx->kessler;
    bz_kessler_fields F;
    F.density = nullptr; F.pressure = nullptr;
    F.potential_temperature = s->theta; F.potential_temperature_density = s->rho_theta;
    F.moisture_density = s->rho_q; F.cloud_liquid_density = K.cloud_liquid_density; F.rain_density = K.rain_density;
    F.vapor_mass_fraction = K.vapor_mass_fraction; F.cloud_liquid_mass_fraction = K.cloud_liquid_mass_fraction;
    F.rain_mass_fraction = K.rain_mass_fraction; F.rain_terminal_velocity = K.rain_terminal_velocity;
    F.precipitation_rate = K.precipitation_rate;
    int rc = bz_kessler_microphysics_update(ctx, &ctx->kessler_params, &F, dt, ctx->kessler_pst);
    if (rc) return rc;
    return bz_update_state(ctx, s, G, 1);
}

extern "C" int bz_kessler_model_update(bz_ctx *ctx, const bz_state *s, const bz_prognostic *G, double dt)
{
    if (!ctx || !s || !G) return BZ_ERR_INVALID;
    if (ctx->dg.microphysics != 2) { ctx->last_error = "bz_kessler_model_update: no Kessler microphysics attached"; return BZ_ERR_INVALID; }
    { const int rcs = bzi_refresh_diagnostics(ctx, s, "bz_kessler_model_update"); if (rcs) return rcs; }
    return bzi_kessler_update(ctx, s, G, dt);
}
