// bz_forcing.hip — forcing, f-plane Coriolis and bottom-flux terms of the BOMEX configuration (BASELINE configs[2],
// /root/reference/examples/bomex.jl:80-207) for the anelastic potential-temperature model.
//   SubsidenceForcing             /root/reference/src/Forcings/subsidence_forcing.jl:75-91,104-126
//   geostrophic_forcings          /root/reference/src/Forcings/geostrophic_forcings.jl  (F_u = -f v_g, F_v = +f u_g: the caller
//                                 hands over the finished specific profiles)
//   SpecificForcing (rho x F)     /root/reference/src/Forcings/specific_forcing.jl:61-74
//   energy forcing in G_rho_theta /root/reference/src/PotentialTemperatureFormulations/potential_temperature_tendency.jl:86-104
//   -x_f_cross_U, -y_f_cross_U    /root/reference/src/AtmosphereModels/dynamics_kernel_functions.jl:79,99 (Oceananigans FPlane)
//   compute_forcings!             /root/reference/src/AtmosphereModels/update_atmosphere_model_state.jl:52,81-86
//   compute_flux_bc_tendencies!   /root/reference/src/AtmosphereModels/update_atmosphere_model_state.jl:418-434
// Every forcing of that configuration is a column profile times the reference density, so the whole stack is one
// streaming pass over the four tendencies it touches plus a per-level reduction for the subsidence averages.
#include "bz_internal.h"

#define FSLICES 8

// per-level partial sums of u, v, theta, q over a slice of rows: partial[(f * Nz + k) * FSLICES + s]
__global__ __launch_bounds__(256) void k_level_sums(DevGrid g, const double *__restrict__ u, const double *__restrict__ v,
                                                    const double *__restrict__ th, const double *__restrict__ q,
                                                    double *__restrict__ partial)
{
    const int k = blockIdx.x, s = blockIdx.y;
    const int j0 = (int)((long long)g.Ny * s / FSLICES), j1 = (int)((long long)g.Ny * (s + 1) / FSLICES);
    double a[4] = {0.0, 0.0, 0.0, 0.0};
    const long long cells = (long long)(j1 - j0) * g.Nx;
    for (long long c = threadIdx.x; c < cells; c += 256) {
        const int j = j0 + (int)(c / g.Nx), i = (int)(c % g.Nx);
        const long long n = g.idx(i, j, k);
        a[0] += u[n]; a[1] += v[n]; a[2] += th[n]; a[3] += q[n];
    }
    __shared__ double red[4][256];
    for (int f = 0; f < 4; ++f) red[f][threadIdx.x] = a[f];
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
        if ((int)threadIdx.x < w)
            for (int f = 0; f < 4; ++f) red[f][threadIdx.x] += red[f][threadIdx.x + w];
        __syncthreads();
    }
    if (threadIdx.x < 4) partial[((long long)threadIdx.x * g.Nz + k) * FSLICES + s] = red[threadIdx.x][0];
}

// per-wavefront sums in the layout and ORDER of the projection + diagnosis kernel (bz_fused.hip: PDFields::lsum: 64 consecutive cells of a row
// added by the same shuffle tree): where the sums of later stages ride on that kernel, the stage that cannot (first stage of a call, host
// calls) sums here, so that both add in one order and n single-step calls leave the bits of one n-step call
__global__ __launch_bounds__(256) void k_level_wave_sums(DevGrid g, const double *__restrict__ u, const double *__restrict__ v,
                                                         const double *__restrict__ th, const double *__restrict__ q,
                                                         double *__restrict__ lsum, long long P)
{
    const int bx = blockIdx.x, j = blockIdx.y, k = blockIdx.z;
    const int i = bx * 256 + threadIdx.x;
    if (i >= g.Nx) return;      // rows of a multiple of 64 cells: whole wavefronts leave
    const long long n = g.idx(i, j, k);
    double a0 = u[n], a1 = v[n], a2 = th[n], a3 = q[n];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        a0 += __shfl_down(a0, off); a1 += __shfl_down(a1, off); a2 += __shfl_down(a2, off); a3 += __shfl_down(a3, off);
    }
    if ((threadIdx.x & 63) == 0) {
        const long long slot = ((long long)j * gridDim.x + bx) * 4 + (threadIdx.x >> 6), fs = (long long)g.Nz * P;
        double *o = lsum + (long long)k * P + slot;
        o[0] = a0; o[fs] = a1; o[2 * fs] = a2; o[3 * fs] = a3;
    }
}

// the same sums from the per-wave values the projection + diagnosis kernel left (bz_fused.hip: PDFields::lsum): block t = f Nz + k adds its P
// values in a fixed order (thread-strided, then an LDS tree) into partial[t]
__global__ __launch_bounds__(256) void k_level_reduce(const double *__restrict__ rows, long long P, double *__restrict__ partial)
{
    const double *r = rows + (long long)blockIdx.x * P;
    double a = 0.0;
    for (long long c = threadIdx.x; c < P; c += 256) a += r[c];
    __shared__ double red[256];
    red[threadIdx.x] = a;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
        if ((int)threadIdx.x < w) red[threadIdx.x] += red[threadIdx.x + w];
        __syncthreads();
    }
    if (threadIdx.x == 0) partial[blockIdx.x] = red[0];
}

// Average(specific field, dims=(1,2)) then F = -zb-average(w_s dz(avg)) (subsidence_forcing.jl:75-91); one block
// (nsl partial sums per field and level: FSLICES of k_level_sums, 1 of k_level_reduce)
__global__ void k_subsidence_profiles(DevGrid g, const double *__restrict__ partial, const double *__restrict__ ws,
                                      double *__restrict__ avg, double *__restrict__ sub, int mask, double count, int nsl)
{
    const int Nz = g.Nz;
    for (int t = threadIdx.x; t < 4 * Nz; t += blockDim.x) {
        double sum = 0.0;
        for (int s = 0; s < nsl; ++s) sum += partial[(long long)t * nsl + s];
        avg[t] = sum / count;
    }
    __syncthreads();
    for (int t = threadIdx.x; t < 4 * Nz; t += blockDim.x) {
        const int f = t / Nz, k = t % Nz;
        double out = 0.0;
        if ((mask >> f) & 1) {
            const double *a = avg + (long long)f * Nz;
            // w_dz(k) = w_s[k] * (a[k] - a[k-1]) / dzf[k] on interior faces k = 1 .. Nz-1
            const double up = (k + 1 <= Nz - 1) ? ws[k + 1] * ((a[k + 1] - a[k]) / g.dzf[k + 1]) : 0.0;
            const double dn = (k >= 1) ? ws[k] * ((a[k] - a[k - 1]) / g.dzf[k]) : 0.0;
            const double mid = (up + dn) / 2;
            out = -((k == Nz - 1) ? dn : ((k == 0) ? up : mid));
        }
        sub[t] = out;
    }
}

struct ForcingCols {
    const double *Fu, *Fv, *Fth, *Fq, *Fe;     // static specific profiles (nullptr: none)
    const double *sub;                         // 4 * Nz subsidence profiles (zeros where inactive), nullptr: none
    double f;
};

// SCALARS = false: the stack holds momentum terms only (Coriolis, geostrophic / u, v profiles — the CBL benchmark case): the
// theta / moisture arrays are not touched (the general form read-modify-writes all four: 8 words per cell instead of 4 + 2)
// MOMENTUM = false (round 5): the momentum terms went into the RK epilogues of the stored-velocity momentum kernels
// (bz_tendency5.hip: bzi_k6_stored with bz_ctx::fold_momentum_forcing); only rho theta / rho q are read-modify-written
template <bool SCALARS, bool MOMENTUM = true>
__global__ __launch_bounds__(256) void k_apply_forcings(DevGrid g, ForcingCols F, double *__restrict__ Gu,
                                                        double *__restrict__ Gv, double *__restrict__ Gth,
                                                        double *__restrict__ Gq, const double *__restrict__ ru,
                                                        const double *__restrict__ rv, const double *__restrict__ q,
                                                        double scale)
{
    const int i = blockIdx.x * 256 + threadIdx.x, j = blockIdx.y, k = blockIdx.z;
    if (i >= g.Nx) return;
    const long long n = g.idx(i, j, k);
    const double rho = g.rho[k];
    const int Nz = g.Nz;
    const long long sx = g.Hx ? 1 : 0, sy = g.Hy ? g.Sx : 0;      // Flat directions have no neighbours
    auto column = [&](const double *stat, int f, bool &any) {
        double tot = 0.0;
        any = false;
        if (F.sub) { tot = rho * F.sub[f * Nz + k]; any = true; }
        if (stat) { tot = any ? tot + rho * stat[k] : rho * stat[k]; any = true; }
        return tot;
    };
    bool any;
    // wall faces (v at j = 0 of a Bounded y, u at i = 0 of a Bounded x) carry no tendency: the wall-aware tendency kernels never write
    // G there, so anything added here would pile up from evaluation to evaluation; as k_apply_relaxation
    if (MOMENTUM && !(g.bounded_x && i == 0)) {
        double G = Gu[n];
        if (F.f != 0.0) {
            const double a = (rv[n - sx] + rv[n - sx + sy]) / 2, b = (rv[n] + rv[n + sy]) / 2;
            G -= scale * (-F.f * ((a + b) / 2));
        }
        const double t = column(F.Fu, 0, any);
        if (any) G += scale * t;
        Gu[n] = G;
    }
    if (MOMENTUM && !(g.bounded_y && j == 0)) {
        double G = Gv[n];
        if (F.f != 0.0) {
            const double a = (ru[n - sy] + ru[n - sy + sx]) / 2, b = (ru[n] + ru[n + sx]) / 2;
            G -= scale * (F.f * ((a + b) / 2));
        }
        const double t = column(F.Fv, 1, any);
        if (any) G += scale * t;
        Gv[n] = G;
    }
    if constexpr (!SCALARS) return;
    {
        double G = Gth[n];
        const double t = column(F.Fth, 2, any);
        if (any) G += scale * t;
        if (F.Fe) {
            double qv, ql = 0.0;
            if (g.microphysics == 1) { qv = g.qv_field[n]; ql = g.ql_field[n]; }
            else qv = q[n];
            const double qd = 1.0 - (qv + ql);
            const double Rm = qd * g.Rd + qv * g.Rv;
            const double cpm = qd * g.cpd + qv * g.cpv + ql * g.sa_cl;
            const double Pi = pow(g.p_r[k] / g.pst, Rm / cpm);
            G += scale * ((rho * F.Fe[k]) / (cpm * Pi));
        }
        Gth[n] = G;
    }
    {
        const double t = column(F.Fq, 3, any);
        if (any) Gq[n] += scale * t;
    }
}

// bottom FluxBoundaryConditions: G[i,j,1] += J / dz_1 (Oceananigans apply_z_bcs!); the drag flux of examples/bomex.jl:95-101
__global__ __launch_bounds__(256) void k_bottom_flux(DevGrid g, double Jth, double Jq, double Je, const double *__restrict__ qfield, double drag, double drag_eps, double *__restrict__ Gu,
                                                     double *__restrict__ Gv, double *__restrict__ Gth,
                                                     double *__restrict__ Gq, const double *__restrict__ ru,
                                                     const double *__restrict__ rv, double scale)
{
    const int i = blockIdx.x * 256 + threadIdx.x, j = blockIdx.y;
    if (i >= g.Nx) return;
    const long long n = g.idx(i, j, 0);
    const double dz = g.dzc[0];
    const long long sx = g.Hx ? 1 : 0, sy = g.Hy ? g.Sx : 0;
    if (Jth != 0.0) Gth[n] += scale * (Jth / dz);
    if (Jq != 0.0) Gq[n] += scale * (Jq / dz);
    if (Je != 0.0) {      // energy flux -> theta flux: Q / c_pm of the lowest cell (moisture fractions as the microphysics holds them)
        double qv, ql = 0.0;
        if (g.microphysics == 1) { qv = g.qv_field[n]; ql = g.ql_field[n]; }
        else qv = qfield[n];
        const double qd = 1.0 - (qv + ql);
        const double cpm = qd * g.cpd + qv * g.cpv + ql * g.sa_cl;
        Gth[n] += scale * ((Je / cpm) / dz);
    }
    if (drag != 0.0) {
        const double u = ru[n], v = rv[n];
        const double va = (rv[n - sx] + rv[n - sx + sy]) / 2, vb = (rv[n] + rv[n + sy]) / 2;
        const double v_fc = (va + vb) / 2;
        const double ua = (ru[n - sy] + ru[n - sy + sx]) / 2, ub = (ru[n] + ru[n + sx]) / 2;
        const double u_cf = (ua + ub) / 2;
        if (!(g.bounded_x && i == 0)) Gu[n] += scale * ((-drag * u / sqrt(u * u + v_fc * v_fc + drag_eps)) / dz);      // wall faces carry no tendency
        if (!(g.bounded_y && j == 0)) Gv[n] += scale * ((-drag * v / sqrt(u_cf * u_cf + v * v + drag_eps)) / dz);
    }
}

// BulkDrag / BulkSensibleHeatFlux / BulkVaporFlux with constant coefficients on unfiltered fields
// (/root/reference/src/BoundaryConditions/bulk_drag.jl:114-135, bulk_scalar_fluxes.jl:82-90,123-137,206-232,
//  BoundaryConditions.jl:64-85; surface density reference_states.jl:73-76): J^u = -rho0 C^D U~ u at the x face with
// U~^2 = u^2 + xy-average(v^2) + gustiness^2, scalar fluxes at centres with U~^2 = x-average(u^2) + y-average(v^2) + gustiness^2.
struct BulkParams {
    double drag_c, drag_g2, drag_rho0;        // rho0 = p0 / (Rd T0) of each condition's own surface temperature
    double heat_c, heat_g2, heat_rho0, heat_theta0;
    double vap_c, vap_g2, vap_rho0, vap_q0;
    int drag, heat, vapor;
};

__global__ __launch_bounds__(256) void k_bulk_bottom_flux(DevGrid g, BulkParams B, double *__restrict__ Gu, double *__restrict__ Gv,
                                                          double *__restrict__ Gth, double *__restrict__ Gq,
                                                          const double *__restrict__ u, const double *__restrict__ v,
                                                          const double *__restrict__ th, const double *__restrict__ qv, double scale)
{
    const int i = blockIdx.x * 256 + threadIdx.x, j = blockIdx.y;
    if (i >= g.Nx) return;
    const long long n = g.idx(i, j, 0);
    const double dz = g.dzc[0];
    const long long sx = g.Hx ? 1 : 0, sy = g.Hy ? g.Sx : 0;
    auto sq = [](double a) { return a * a; };
    if (B.drag) {
        const double v2 = ((sq(v[n - sx]) + sq(v[n - sx + sy])) / 2 + (sq(v[n]) + sq(v[n + sy])) / 2) / 2;
        const double u2 = ((sq(u[n - sy]) + sq(u[n - sy + sx])) / 2 + (sq(u[n]) + sq(u[n + sx])) / 2) / 2;
        const double Ju = -B.drag_rho0 * B.drag_c * sqrt(sq(u[n]) + v2 + B.drag_g2) * u[n];
        const double Jv = -B.drag_rho0 * B.drag_c * sqrt(u2 + sq(v[n]) + B.drag_g2) * v[n];
        if (!(g.bounded_x && i == 0)) Gu[n] += scale * (Ju / dz);
        if (!(g.bounded_y && j == 0)) Gv[n] += scale * (Jv / dz);
    }
    if (B.heat || B.vapor) {
        const double U2 = (sq(u[n]) + sq(u[n + sx])) / 2 + (sq(v[n]) + sq(v[n + sy])) / 2;
        if (B.heat) Gth[n] += scale * ((-B.heat_rho0 * B.heat_c * sqrt(U2 + B.heat_g2) * (th[n] - B.heat_theta0)) / dz);
        if (B.vapor) Gq[n] += scale * ((-B.vap_rho0 * B.vap_c * sqrt(U2 + B.vap_g2) * (qv[n] - B.vap_q0)) / dz);
    }
}

static void free_forcings(bz_ctx *ctx)
{
    if (ctx->d_forcing) hipFree(ctx->d_forcing);
    ctx->d_forcing = nullptr;
    if (ctx->d_lsum_rows) hipFree(ctx->d_lsum_rows);
    ctx->d_lsum_rows = nullptr;
    ctx->lsum_P = 0;
    ctx->lsum_fresh = false;
    ctx->has_forcings = false;
}

// ---- Relaxation(rate, mask(z), target(z)) sponges (include/breeze_hip.h: bz_column_relaxation) ----
struct RelaxCols {
    const double *rate[5], *target[5];      // u v w theta q; nullptr = absent
    int specific;
};

__global__ __launch_bounds__(256) void k_apply_relaxation(DevGrid g, RelaxCols R, const double *__restrict__ ru, const double *__restrict__ rv,
                                                          const double *__restrict__ rw, const double *__restrict__ rth, const double *__restrict__ rq,
                                                          const double *__restrict__ u, const double *__restrict__ v, const double *__restrict__ w,
                                                          double *__restrict__ Gu, double *__restrict__ Gv, double *__restrict__ Gw,
                                                          double *__restrict__ Gth, double *__restrict__ Gq, const double *__restrict__ Fth,
                                                          const double *__restrict__ rho3d, int fth_specific)
{
    const int i = blockIdx.x * 256 + threadIdx.x, j = blockIdx.y, k = blockIdx.z;
    if (i >= g.Nx) return;
    const long long n = g.idx(i, j, k);
    // F = (rate mask) (target - field) (Oceananigans Relaxation); the specific-keyed form is rho_r F(specific field) (specific_forcing.jl:61-74)
    if (R.rate[0] && !(g.bounded_x && i == 0)) {
        const double F = R.rate[0][k] * (R.target[0][k] - ((R.specific & 1) ? u[n] : ru[n]));
        Gu[n] += (R.specific & 1) ? g.rho[k] * F : F;
    }
    if (R.rate[1] && !(g.bounded_y && j == 0)) {
        const double F = R.rate[1][k] * (R.target[1][k] - ((R.specific & 2) ? v[n] : rv[n]));
        Gv[n] += (R.specific & 2) ? g.rho[k] * F : F;
    }
    if (R.rate[2] && k >= 1) {      // interior faces: the wall faces carry no tendency
        const double F = R.rate[2][k] * (R.target[2][k] - ((R.specific & 4) ? w[n] : rw[n]));
        Gw[n] += (R.specific & 4) ? g.rho_f[k] * F : F;
    }
    if (R.rate[3]) Gth[n] += R.rate[3][k] * (R.target[3][k] - rth[n]);
    if (R.rate[4]) Gq[n] += R.rate[4][k] * (R.target[4][k] - rq[n]);
    // Forcing(field) on the thermodynamic variable (bz_set_field_forcing): rho F for the specific key, with the coupling density
    if (Fth) Gth[n] += fth_specific ? (rho3d ? rho3d[n] : g.rho[k]) * Fth[n] : Fth[n];
}

static void free_relaxation(bz_ctx *ctx)
{
    if (ctx->d_relax) hipFree(ctx->d_relax);
    ctx->d_relax = nullptr;
    ctx->relax_mask = ctx->relax_specific = 0;
    ctx->has_relaxation = ctx->field_forcing != nullptr;
}

extern "C" int bz_set_field_forcing(bz_ctx *ctx, const double *F, int specific)
{
    if (!ctx) return BZ_ERR_INVALID;
    ++ctx->config_epoch;
    if (F && ctx->slab_mode && ctx->compressible) { ctx->last_error = "bz_set_field_forcing: compressible y-slabs are not built"; return BZ_ERR_UNSUPPORTED; }
    ctx->field_forcing = F;
    ctx->field_forcing_specific = specific ? 1 : 0;
    ctx->has_relaxation = ctx->relax_mask != 0 || F != nullptr;      // one pass adds the sponges and this term (bzi_apply_relaxation)
    return BZ_OK;
}

extern "C" int bz_set_relaxation(bz_ctx *ctx, const bz_column_relaxation *r)
{
    if (!ctx) return BZ_ERR_INVALID;
    // validate first: a rejected request leaves the sponge that is attached untouched
    if (r && ctx->compressible && (ctx->slab_mode || r->specific_mask || r->rate_moisture)) {
        ctx->last_error = "bz_set_relaxation: CompressibleDynamics takes the density-keyed sponges of rho u, rho v, rho w, rho theta on single-device contexts";
        return BZ_ERR_UNSUPPORTED;
    }
    if (r && (r->specific_mask & ~7)) { ctx->last_error = "bz_set_relaxation: specific_mask names u (1), v (2), w (4)"; return BZ_ERR_INVALID; }
    ++ctx->config_epoch;
    free_relaxation(ctx);
    if (!r) return BZ_OK;
    const int Nz = ctx->dg.Nz, L = Nz + 1;
    const double *rate[5] = {r->rate_u, r->rate_v, r->rate_w, r->rate_theta, r->rate_moisture};
    const double *target[5] = {r->target_u, r->target_v, r->target_w, r->target_theta, r->target_moisture};
    // a HIP failure on the way detaches cleanly: the context is either unchanged (above) or without a sponge
    auto fail = [&](hipError_t e) {
        ctx->last_error = std::string("bz_set_relaxation: ") + hipGetErrorString(e);
        free_relaxation(ctx);
        return -(int)e;
    };
    hipError_t e;
    if ((e = hipMalloc(&ctx->d_relax, (size_t)10 * L * sizeof(double))) != hipSuccess) { ctx->d_relax = nullptr; return fail(e); }
    if ((e = hipMemsetAsync(ctx->d_relax, 0, (size_t)10 * L * sizeof(double), ctx->stream)) != hipSuccess) return fail(e);
    int mask = 0;
    for (int c = 0; c < 5; ++c) {
        if (!rate[c]) continue;
        const size_t len = (size_t)(c == 2 ? Nz + 1 : Nz) * sizeof(double);
        if ((e = hipMemcpyAsync(ctx->d_relax + (size_t)(2 * c) * L, rate[c], len, hipMemcpyHostToDevice, ctx->stream)) != hipSuccess) return fail(e);
        if (target[c] && (e = hipMemcpyAsync(ctx->d_relax + (size_t)(2 * c + 1) * L, target[c], len, hipMemcpyHostToDevice, ctx->stream)) != hipSuccess) return fail(e);
        mask |= 1 << c;
    }
    if ((e = hipStreamSynchronize(ctx->stream)) != hipSuccess) return fail(e);      // the host columns may go away after the call
    ctx->relax_mask = mask;
    ctx->relax_specific = r->specific_mask;
    ctx->has_relaxation = ctx->relax_mask != 0 || ctx->field_forcing != nullptr;
    return BZ_OK;
}

int bzi_apply_relaxation(bz_ctx *ctx, const bz_state *s, const bz_prognostic *G, const double *rho3d)
{
    if (!ctx->has_relaxation) return BZ_OK;
    const DevGrid &g = ctx->dg;
    ProfileScope ps(ctx, "relaxation_forcings");
    RelaxCols R;
    const int L = g.Nz + 1;
    for (int c = 0; c < 5; ++c) {
        const bool on = ctx->relax_mask & (1 << c);
        R.rate[c] = on ? ctx->d_relax + (size_t)(2 * c) * L : nullptr;
        R.target[c] = on ? ctx->d_relax + (size_t)(2 * c + 1) * L : nullptr;
    }
    R.specific = ctx->relax_specific;
    hipLaunchKernelGGL(k_apply_relaxation, dim3((g.Nx + 255) / 256, g.Ny, g.Nz), dim3(256), 0, ctx->stream, g, R, s->rho_u, s->rho_v, s->rho_w,
                       s->rho_theta, s->rho_q, s->u, s->v, s->w, G->rho_u, G->rho_v, G->rho_w, G->rho_theta, G->rho_q, ctx->field_forcing, rho3d,
                       ctx->field_forcing_specific);
    BZ_LAUNCH_CHECK();
    return BZ_OK;
}

void bzi_forcing_teardown(bz_ctx *ctx) { free_forcings(ctx); free_relaxation(ctx); }

extern "C" int bz_set_forcings(bz_ctx *ctx, const bz_column_forcings *f)
{
    if (ctx) ++ctx->config_epoch;      // captured steps (bz_graph.hip) belong to one configuration
    if (!ctx) return BZ_ERR_INVALID;
    free_forcings(ctx);
    if (!f) return BZ_OK;
    // CompressibleDynamics: the f-plane term of the slow momentum tendencies alone (examples/tropical_cyclone_with_rainband.jl:508-514);
    // the column profiles are specific forcings of the anelastic model (times rho_r), the bottom fluxes belong to its flux-BC pass
    const bool coriolis_only = !f->u_forcing && !f->v_forcing && !f->theta_forcing && !f->moisture_forcing && !f->energy_forcing &&
                               !f->subsidence_vertical_velocity && f->bottom_theta_flux == 0.0 && f->bottom_moisture_flux == 0.0 &&
                               f->bottom_drag_rho0_ustar2 == 0.0 && f->bottom_energy_flux == 0.0;
    if ((ctx->compressible && (!coriolis_only || ctx->slab_mode)) || (!ctx->compressible && (ctx->dg.formulation != 0 || ctx->dg.microphysics == 2))) {      // y-slab contexts: through the library-owned distributed step (bz_comm.hip)
        ctx->last_error = "bz_set_forcings: the forcing stack is implemented for the anelastic "
                          "potential-temperature model (microphysics nothing or SaturationAdjustment); single-device compressible contexts "
                          "take coriolis_f alone";
        return BZ_ERR_UNSUPPORTED;
    }
    const int Nz = ctx->dg.Nz;
    // layout: [Fu Fv Fth Fq Fe](5 Nz) [ws](Nz+1) [avg](4 Nz) [sub](4 Nz) [partial](4 Nz FSLICES)
    const size_t total = (size_t)5 * Nz + (Nz + 1) + 8 * (size_t)Nz + (size_t)4 * Nz * FSLICES;
    BZ_HIP(hipMalloc(&ctx->d_forcing, total * sizeof(double)));
    BZ_HIP(hipMemsetAsync(ctx->d_forcing, 0, total * sizeof(double), ctx->stream));
    const double *stat[5] = {f->u_forcing, f->v_forcing, f->theta_forcing, f->moisture_forcing, f->energy_forcing};
    ctx->forcing_static_mask = 0;
    for (int c = 0; c < 5; ++c)
        if (stat[c]) {
            BZ_HIP(hipMemcpyAsync(ctx->d_forcing + (size_t)c * Nz, stat[c], Nz * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
            ctx->forcing_static_mask |= 1 << c;
        }
    ctx->forcing_subsidence_mask = 0;
    if (f->subsidence_vertical_velocity) {
        BZ_HIP(hipMemcpyAsync(ctx->d_forcing + (size_t)5 * Nz, f->subsidence_vertical_velocity, (Nz + 1) * sizeof(double),
                              hipMemcpyHostToDevice, ctx->stream));
        ctx->forcing_subsidence_mask = (f->subsidence_u ? 1 : 0) | (f->subsidence_v ? 2 : 0) | (f->subsidence_theta ? 4 : 0) |
                                       (f->subsidence_moisture ? 8 : 0);
    }
    BZ_HIP(hipStreamSynchronize(ctx->stream));      // the host profiles may go away after the call
    ctx->forcing_f = f->coriolis_f;
    ctx->forcing_flux_theta = f->bottom_theta_flux;
    ctx->forcing_flux_q = f->bottom_moisture_flux;
    ctx->forcing_flux_energy = f->bottom_energy_flux;
    ctx->forcing_drag = f->bottom_drag_rho0_ustar2;
    ctx->forcing_drag_eps = f->bottom_drag_epsilon;
    ctx->has_forcings = true;
    return BZ_OK;
}

extern "C" int bz_set_bulk_surface_fluxes(bz_ctx *ctx, const bz_bulk_surface_fluxes *b)
{
    if (ctx) ++ctx->config_epoch;      // captured steps (bz_graph.hip) belong to one configuration
    if (!ctx) return BZ_ERR_INVALID;
    if (!b) { ctx->has_bulk = false; return BZ_OK; }
    if (ctx->compressible || ctx->dg.formulation != 0 || ctx->dg.microphysics == 2) {      // y-slab contexts: through the library-owned distributed step (bz_comm.hip)
        ctx->last_error = "bz_set_bulk_surface_fluxes: implemented for the anelastic potential-temperature model";
        return BZ_ERR_UNSUPPORTED;
    }
    ctx->bulk = *b;
    ctx->has_bulk = (b->drag_coefficient > 0.0) || (b->heat_coefficient > 0.0) || (b->vapor_coefficient > 0.0);
    return BZ_OK;
}

static int bulk_flux(bz_ctx *ctx, const bz_state *s, double *Gu, double *Gv, double *Gth, double *Gq, double scale)
{
    const DevGrid &g = ctx->dg;
    const bz_bulk_surface_fluxes &b = ctx->bulk;
    const double Rd = g.Rd;
    BulkParams B;
    B.drag = b.drag_coefficient > 0.0; B.heat = b.heat_coefficient > 0.0; B.vapor = b.vapor_coefficient > 0.0;
    B.drag_c = b.drag_coefficient; B.drag_g2 = b.drag_gustiness * b.drag_gustiness;
    B.drag_rho0 = B.drag ? b.surface_pressure / (Rd * b.drag_surface_temperature) : 0.0;
    B.heat_c = b.heat_coefficient; B.heat_g2 = b.heat_gustiness * b.heat_gustiness;
    B.heat_rho0 = B.heat ? b.surface_pressure / (Rd * b.heat_surface_temperature) : 0.0;
    B.heat_theta0 = B.heat ? b.heat_surface_temperature / pow(b.surface_pressure / b.standard_pressure, Rd / g.cpd) : 0.0;
    B.vap_c = b.vapor_coefficient; B.vap_g2 = b.vapor_gustiness * b.vapor_gustiness;
    B.vap_rho0 = 0.0; B.vap_q0 = 0.0;
    if (B.vapor) {      // saturation_specific_humidity(T0, rho0, constants, PlanarLiquidSurface()) (Clausius-Clapeyron)
        const double T0 = b.vapor_surface_temperature;
        B.vap_rho0 = b.surface_pressure / (Rd * T0);
        const double dc = ctx->constants.vapor_heat_capacity - b.liquid_heat_capacity;
        const double L0 = b.liquid_latent_heat - dc * b.energy_reference_temperature;
        const double ps = b.triple_point_pressure * pow(T0 / b.triple_point_temperature, dc / g.Rv) *
                          exp((1.0 / b.triple_point_temperature - 1.0 / T0) * L0 / g.Rv);
        B.vap_q0 = ps / (B.vap_rho0 * g.Rv * T0);
    }
    ProfileScope ps(ctx, "bulk_surface_fluxes");
    const double *qv = (g.microphysics == 1) ? g.qv_field : s->q;
    hipLaunchKernelGGL(k_bulk_bottom_flux, dim3((g.Nx + 255) / 256, g.Ny), dim3(256), 0, ctx->stream, g, B, Gu, Gv, Gth, Gq,
                       s->u, s->v, s->theta, qv, scale);
    BZ_LAUNCH_CHECK();
    return BZ_OK;
}

// The horizontal sums can ride on the projection + diagnosis kernel of the fused-RK tier (single device, rows of a multiple of 64 cells:
// whole wavefronts).  Summation order differs from k_level_sums (waves of a row, then rows; there: strided threads over a slice of rows).
bool bzi_level_sums_ride(const bz_ctx *ctx)
{
    return ctx->has_forcings && ctx->forcing_subsidence_mask && !ctx->slab_mode && !ctx->compressible && ctx->dg.Nx % 64 == 0 &&
           !ctx->tune.no_fuse_level_sums;
}
double *bzi_level_sum_rows(bz_ctx *ctx, long long *P)
{
    if (!bzi_level_sums_ride(ctx)) return nullptr;
    const DevGrid &g = ctx->dg;
    const long long p = (long long)g.Ny * ((g.Nx + 255) / 256) * 4;
    if (!ctx->d_lsum_rows) {
        const size_t bytes = (size_t)4 * g.Nz * (size_t)p * sizeof(double);
        if (hipMalloc(&ctx->d_lsum_rows, bytes) != hipSuccess) { (void)hipGetLastError(); ctx->d_lsum_rows = nullptr; return nullptr; }
        // (the slots of wavefronts beyond the end of a row are never written: they stay zero)
        if (hipMemsetAsync(ctx->d_lsum_rows, 0, bytes, ctx->stream) != hipSuccess) return nullptr;
        ctx->lsum_P = p;
    }
    *P = ctx->lsum_P;
    return ctx->d_lsum_rows;
}

// compute_forcings!(model) (update_atmosphere_model_state.jl:81-86): horizontal averages -> subsidence profiles
extern "C" int bz_compute_forcings(bz_ctx *ctx, const bz_state *s)
{
    if (!ctx || !s) return BZ_ERR_INVALID;
    ctx->lsum_fresh = false;      // a host call: the fields may have been written since the last step
    return bzi_compute_forcings(ctx, s);
}

int bzi_compute_forcings(bz_ctx *ctx, const bz_state *s)
{
    if (!ctx->has_forcings || !ctx->forcing_subsidence_mask) return BZ_OK;
    { const int rcs = bzi_refresh_diagnostics(ctx, s, "bz_compute_forcings"); if (rcs) return rcs; }
    const DevGrid &g = ctx->dg;
    const int Nz = g.Nz;
    const bool ride = ctx->lsum_fresh && ctx->d_lsum_rows;
    // a context whose later stages ride sums its other stages in the riding order too (k_level_wave_sums): one order per context
    long long Pw = 0;
    double *rows = ride ? ctx->d_lsum_rows : bzi_level_sum_rows(ctx, &Pw);
    ProfileScope ps(ctx, ride ? "subsidence_averages_from_wave_sums" : "subsidence_averages");
    double *ws = ctx->d_forcing + (size_t)5 * Nz, *avg = ws + (Nz + 1), *sub = avg + (size_t)4 * Nz, *partial = sub + (size_t)4 * Nz;
    if (!ride && rows)
        hipLaunchKernelGGL(k_level_wave_sums, dim3((g.Nx + 255) / 256, g.Ny, Nz), dim3(256), 0, ctx->stream, g, s->u, s->v, s->theta, s->q, rows, Pw);
    if (rows) {
        hipLaunchKernelGGL(k_level_reduce, dim3(4 * Nz), dim3(256), 0, ctx->stream, ctx->d_lsum_rows, ctx->lsum_P, partial);
        hipLaunchKernelGGL(k_subsidence_profiles, dim3(1), dim3(256), 0, ctx->stream, g, partial, ws, avg, sub,
                           ctx->forcing_subsidence_mask, (double)g.Nx * (double)ctx->Ny_global, 1);
        BZ_LAUNCH_CHECK();
        return BZ_OK;
    }
    hipLaunchKernelGGL(k_level_sums, dim3(Nz, FSLICES), dim3(256), 0, ctx->stream, g, s->u, s->v, s->theta, s->q, partial);
    if (ctx->slab_mode) {      // horizontal averages run over the whole domain: add the other ranks' partial sums (rank order: same bits everywhere)
        if (!ctx->comm) { ctx->last_error = "bz_compute_forcings: a y-slab context needs a communicator for the horizontal averages"; return BZ_ERR_UNSUPPORTED; }
        int rc = bzi_comm_allreduce_sum(ctx, partial, 4 * Nz * FSLICES);
        if (rc) return rc;
    }
    hipLaunchKernelGGL(k_subsidence_profiles, dim3(1), dim3(256), 0, ctx->stream, g, partial, ws, avg, sub,
                       ctx->forcing_subsidence_mask, (double)g.Nx * (double)ctx->Ny_global, FSLICES);
    BZ_LAUNCH_CHECK();
    return BZ_OK;
}

// the forcing + Coriolis terms of compute_tendencies!, added to the advective tendencies already in G
// scale = 1 adds to tendencies; the whole-step seam passes scale = alpha dt and the arrays the fused RK update just wrote
// (predictor momentum in G, rho_theta / rho_q in place): u_new = (1-alpha) u0 + alpha (u + dt (G + F)) either way.
int bzi_apply_forcings(bz_ctx *ctx, const bz_state *s, double *Gu, double *Gv, double *Gth, double *Gq, double scale, bool momentum_done)
{
    const DevGrid &g = ctx->dg;
    const int Nz = g.Nz;
    // momentum_done: the caller computed the stage's subsidence profiles (bz_compute_forcings) before its tendency launches and the
    // momentum kernels carried Coriolis / u, v profiles / u, v subsidence in their epilogues
    int rc = momentum_done ? BZ_OK : bzi_compute_forcings(ctx, s);
    if (rc) return rc;
    ProfileScope ps(ctx, "forcing_tendencies");
    ForcingCols F;
    const double *base = ctx->d_forcing;
    const int m = ctx->forcing_static_mask;
    F.Fu = (m & 1) ? base : nullptr;
    F.Fv = (m & 2) ? base + (size_t)Nz : nullptr;
    F.Fth = (m & 4) ? base + (size_t)2 * Nz : nullptr;
    F.Fq = (m & 8) ? base + (size_t)3 * Nz : nullptr;
    F.Fe = (m & 16) ? base + (size_t)4 * Nz : nullptr;
    F.sub = ctx->forcing_subsidence_mask ? base + (size_t)5 * Nz + (Nz + 1) + (size_t)4 * Nz : nullptr;
    F.f = ctx->forcing_f;
    const bool scalars = F.Fth || F.Fq || F.Fe || (ctx->forcing_subsidence_mask & 12);
    if (momentum_done) {
        if (scalars)
            hipLaunchKernelGGL((k_apply_forcings<true, false>), dim3((g.Nx + 255) / 256, g.Ny, g.Nz), dim3(256), 0, ctx->stream, g, F, Gu, Gv, Gth, Gq,
                               s->rho_u, s->rho_v, s->q, scale);
    } else if (scalars)
        hipLaunchKernelGGL(k_apply_forcings<true>, dim3((g.Nx + 255) / 256, g.Ny, g.Nz), dim3(256), 0, ctx->stream, g, F, Gu, Gv, Gth, Gq,
                           s->rho_u, s->rho_v, s->q, scale);
    else
        hipLaunchKernelGGL(k_apply_forcings<false>, dim3((g.Nx + 255) / 256, g.Ny, g.Nz), dim3(256), 0, ctx->stream, g, F, Gu, Gv, Gth, Gq,
                           s->rho_u, s->rho_v, s->q, scale);
    BZ_LAUNCH_CHECK();
    return BZ_OK;
}

// Forcing stacks the lean whole-step seam carries (bz_step.hip): the kernels above read rho u, rho v of the previous stage only — which
// the lean seam keeps intact while the predictor sits in the G slots — when there is no subsidence (its horizontal averages need
// the stored u, v, theta, q diagnostics) and no bulk condition (it reads u, v, theta); theta / moisture / energy profiles would change
// rho theta everywhere after the scalar kernel has written T, so those stacks stay on the fused-RK tier.  What remains is the
// reference's own benchmark case: FPlane + geostrophic forcing + bottom fluxes (benchmarking/src/convective_boundary_layer.jl).
bool bzi_lean_forcings_ok(const bz_ctx *ctx)
{
    // (an energy flux divides by the mixture heat capacity of the stored q^v, which the lean seam does not keep inside a step)
    return ctx->has_forcings && !ctx->has_bulk && ctx->forcing_subsidence_mask == 0 && (ctx->forcing_static_mask & (4 | 8 | 16)) == 0 &&
           ctx->forcing_flux_energy == 0.0;
}

extern "C" int bz_compute_flux_bc_tendencies(bz_ctx *ctx, const bz_state *s, const bz_prognostic *G)
{
    if (!ctx || !s || !G) return BZ_ERR_INVALID;
    { const int rcs = bzi_refresh_diagnostics(ctx, s, "bz_compute_flux_bc_tendencies"); if (rcs) return rcs; }
    if (ctx->G_is_predictor) {      // a fused whole step left predictor momentum in G: the fluxes must land on tendencies
        int rc = bz_compute_tendencies(ctx, s, G);
        if (rc) return rc;
    }
    return bzi_flux_bc(ctx, s, G->rho_u, G->rho_v, G->rho_theta, G->rho_q, 1.0);
}

int bzi_flux_bc(bz_ctx *ctx, const bz_state *s, double *Gu, double *Gv, double *Gth, double *Gq, double scale)
{
    if (ctx->has_bulk) {
        int rcb = bulk_flux(ctx, s, Gu, Gv, Gth, Gq, scale);
        if (rcb) return rcb;
    }
    if (!ctx->has_forcings) return BZ_OK;
    if (ctx->forcing_flux_theta == 0.0 && ctx->forcing_flux_q == 0.0 && ctx->forcing_drag == 0.0 && ctx->forcing_flux_energy == 0.0) return BZ_OK;
    const DevGrid &g = ctx->dg;
    ProfileScope ps(ctx, "flux_bc_tendencies");
    hipLaunchKernelGGL(k_bottom_flux, dim3((g.Nx + 255) / 256, g.Ny), dim3(256), 0, ctx->stream, g, ctx->forcing_flux_theta,
                       ctx->forcing_flux_q, ctx->forcing_flux_energy, s->q, ctx->forcing_drag, ctx->forcing_drag_eps, Gu, Gv, Gth, Gq, s->rho_u,
                       s->rho_v, scale);
    BZ_LAUNCH_CHECK();
    return BZ_OK;
}
