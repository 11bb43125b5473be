// bz_tracers.hip — user tracers of the anelastic AtmosphereModel (`tracers = (:a, :b)`; SURVEY §8 row a6: scalar_tendency for
// moisture *and each tracer*).
//   prognostic rho c in model.tracers, G / U0 mirrors       /root/reference/src/AtmosphereModels/atmosphere_model.jl:224-227,380-387
//   tracer_density_to_specific! / specific_to_density!       /root/reference/src/AtmosphereModels/update_atmosphere_model_state.jl:43,65,88-112
//   scalar_tendency = -div_rhoUc(c) per tracer               /root/reference/src/AtmosphereModels/update_atmosphere_model_state.jl:352-372,
//                                                            /root/reference/src/AtmosphereModels/dynamics_kernel_functions.jl:132-159
//   ssp_rk3_substep! over prognostic_fields(model)           /root/reference/src/TimeSteppers/ssp_runge_kutta_3.jl:114-186
// The reference converts rho c -> c in place before the halo fill and back after the tendencies; here rho c stays untouched and
// the specific field lives in its own array (no ulp-level drift from the round trip).
#include "bz_internal.h"

__global__ __launch_bounds__(256) void k_tracer_specific(DevGrid g, const double *__restrict__ rc, double *__restrict__ c)
{
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
    if (t >= (long long)g.Ny * g.Sx) return;
    const int k = blockIdx.y;
    const long long n = g.Sxy * ((long long)k + g.Hz) + (long long)g.Hy * g.Sx + t;
    c[n] = rc[n] / g.rho[k];
}

template <bool FIRST>
__global__ __launch_bounds__(256) void k_rk3_one(DevGrid g, double *__restrict__ a, double *__restrict__ a0,
                                                 const double *__restrict__ Ga, double dt, double alpha)
{
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
    if (t >= (long long)g.Ny * g.Sx) return;
    const long long n = g.Sxy * ((long long)blockIdx.y + g.Hz) + (long long)g.Hy * g.Sx + t;
    const double oma = 1.0 - alpha;
    const double ua = a[n];
    if (FIRST) {
        a0[n] = ua;
        a[n] = oma * ua + alpha * (ua + dt * Ga[n]);
    } else {
        a[n] = oma * a0[n] + alpha * (ua + dt * Ga[n]);
    }
}

extern "C" int bz_set_tracers(bz_ctx *ctx, int32_t n, const bz_tracer_fields *tracers)
{
    if (ctx) ++ctx->config_epoch;      // captured steps (bz_graph.hip) belong to one configuration
    if (!ctx || n < 0 || (n > 0 && !tracers)) return BZ_ERR_INVALID;
    if (n > BZ_MAX_TRACERS) { ctx->last_error = "bz_set_tracers: more than BZ_MAX_TRACERS tracers"; return BZ_ERR_UNSUPPORTED; }
    if (n > 0 && ctx->compressible) {
        ctx->last_error = "bz_set_tracers: user tracers are implemented for the anelastic model";
        return BZ_ERR_UNSUPPORTED;
    }
    for (int t = 0; t < n; ++t)
        if (!tracers[t].density || !tracers[t].specific || !tracers[t].U0 || !tracers[t].G) return BZ_ERR_INVALID;
    ctx->n_tracers = n;
    for (int t = 0; t < n; ++t) ctx->tracers[t] = tracers[t];
    return BZ_OK;
}

// tracer_density_to_specific! + the halo fill of the specific field
int bzi_tracer_specific(bz_ctx *ctx)
{
    if (!ctx->n_tracers) return BZ_OK;
    const DevGrid &g = ctx->dg;
    ProfileScope ps(ctx, "tracer_specific");
    const long long per_level = (long long)g.Ny * g.Sx;
    dim3 grid((unsigned)((per_level + 255) / 256), g.Nz), block(256);
    double *f[BZ_MAX_TRACERS];
    int kd[BZ_MAX_TRACERS];
    for (int t = 0; t < ctx->n_tracers; ++t) {
        hipLaunchKernelGGL(k_tracer_specific, grid, block, 0, ctx->stream, g, ctx->tracers[t].density, ctx->tracers[t].specific);
        f[t] = ctx->tracers[t].specific;
        kd[t] = 0;
    }
    BZ_LAUNCH_CHECK();
    return bzi_fill_halos_multi(ctx, f, kd, ctx->n_tracers);
}

int bzi_tracer_rk3(bz_ctx *ctx, double dt, double alpha, bool first)
{
    if (!ctx->n_tracers) return BZ_OK;
    const DevGrid &g = ctx->dg;
    ProfileScope ps(ctx, "tracer_rk3");
    const long long per_level = (long long)g.Ny * g.Sx;
    dim3 grid((unsigned)((per_level + 255) / 256), g.Nz), block(256);
    for (int t = 0; t < ctx->n_tracers; ++t) {
        const bz_tracer_fields &T = ctx->tracers[t];
        if (first) hipLaunchKernelGGL(k_rk3_one<true>, grid, block, 0, ctx->stream, g, T.density, T.U0, T.G, dt, alpha);
        else hipLaunchKernelGGL(k_rk3_one<false>, grid, block, 0, ctx->stream, g, T.density, T.U0, T.G, dt, alpha);
    }
    BZ_LAUNCH_CHECK();
    return BZ_OK;
}

int bzi_tracer_store_initial_state(bz_ctx *ctx)
{
    const DevGrid &g = ctx->dg;
    const size_t nc = (size_t)g.Sxy * (size_t)(g.Nz + 2 * g.Hz) * sizeof(double);
    for (int t = 0; t < ctx->n_tracers; ++t)
        BZ_HIP(hipMemcpyAsync(ctx->tracers[t].U0, ctx->tracers[t].density, nc, hipMemcpyDeviceToDevice, ctx->stream));
    return BZ_OK;
}
