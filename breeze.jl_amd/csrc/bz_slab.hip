// bz_slab.hip — entry points for the y-slab (one process per GPU) decomposition.  The collective steps —
// y-halo exchange and the transposes of the distributed Fourier transform — are done by the host
// (breeze.jl_amd/distributed.py, torch.distributed over RCCL/xGMI); these calls are the rank-local kernels
// in between.  They are the same kernels the single-GPU whole-step seam uses (bz_fused.hip, bz_poisson.hip).
#include "bz_internal.h"

__global__ void k_tridiag_solve(int NXH, int Ny, int Nz, const double *__restrict__ lower,
                                const double *__restrict__ ibeta, const double *__restrict__ tfac,
                                double2 *__restrict__ hat, double scale, int mean_column);

extern "C" int bz_slab_info(bz_ctx *ctx, int32_t *y_nranks, int32_t *y_rank, int32_t *nkx, int32_t *kx0, int32_t *ny_global)
{
    if (!ctx) return BZ_ERR_INVALID;
    if (y_nranks) *y_nranks = ctx->y_nranks;
    if (y_rank) *y_rank = ctx->y_rank;
    if (nkx) *nkx = ctx->nkx;
    if (kx0) *kx0 = ctx->kx0;
    if (ny_global) *ny_global = ctx->Ny_global;
    return BZ_OK;
}

extern "C" int bz_ssp_rk3_substep_fused(bz_ctx *ctx, const bz_state *s, const bz_prognostic *U0,
                                        const bz_prognostic *G, double dt, double alpha, int first)
{
    if (!ctx || !s || !U0 || !G) return BZ_ERR_INVALID;
    if (ctx->G_is_predictor) {
        int rc = bz_compute_tendencies(ctx, s, G);
        if (rc) return rc;
    }
    return bzi_rk3_fused(ctx, s, U0, G, dt, alpha, first != 0);
}

extern "C" int bz_poisson_source_term(bz_ctx *ctx, const bz_state *s, double dt, double *rhs)
{
    if (!ctx || !s || !rhs) return BZ_ERR_INVALID;
    return bzi_poisson_source_fused(ctx, s, dt, rhs);
}

extern "C" int bz_spectral_tridiagonal_solve(bz_ctx *ctx, double *hat, double scale)
{
    if (!ctx || !hat) return BZ_ERR_INVALID;
    ProfileScope ps(ctx, "poisson_tridiagonal");
    const int Ny = ctx->slab_mode ? ctx->Ny_global : ctx->dg.Ny;
    long long plane = (long long)ctx->NXH * Ny;
    hipLaunchKernelGGL(k_tridiag_solve, dim3((unsigned)((plane + 63) / 64)), dim3(64), 0, ctx->stream, ctx->NXH, Ny,
                       ctx->dg.Nz, ctx->d_lower, ctx->d_ibeta, ctx->d_tfac, (double2 *)hat, scale,
                       ctx->kx0 == 0 ? 1 : 0);
    BZ_LAUNCH_CHECK();
    return BZ_OK;
}

extern "C" int bz_project_and_diagnose(bz_ctx *ctx, const bz_state *s, const double *phi_c, const double *phi_below,
                                       double dt)
{
    if (!ctx || !s || !phi_c) return BZ_ERR_INVALID;
    if (!ctx->dg.wrap_y && !phi_below) return BZ_ERR_INVALID;
    return bzi_project_diagnose(ctx, s, dt, phi_c, phi_below);
}

// compute_tendencies! with the following ssp_rk3_substep! folded in (see bz_time_step_anelastic): predictor
// momentum -> G->rho_u/v/w, rho_theta and rho_q advance in place, U0 is filled when first != 0.
extern "C" int bz_tendencies_fused_rk(bz_ctx *ctx, const bz_state *s, const bz_prognostic *U0, const bz_prognostic *G,
                                      double dt, double alpha, int first)
{
    if (!ctx || !s || !U0 || !G) return BZ_ERR_INVALID;
    const DevGrid &g = ctx->dg;
    if (first) {      // wall faces of the predictor stay 0
        BZ_HIP(hipMemsetAsync(G->rho_w + g.Sxy * g.Hz, 0, g.Sxy * sizeof(double), ctx->stream));
        BZ_HIP(hipMemsetAsync(G->rho_w + g.Sxy * (g.Hz + g.Nz), 0, g.Sxy * sizeof(double), ctx->stream));
    }
    int rc = bzi_tendencies_fused_rk(ctx, s, U0, G, dt, alpha, first != 0);
    ctx->G_is_predictor = true;
    return rc;
}

extern "C" int bz_poisson_source_term_from(bz_ctx *ctx, const bz_state *s, const bz_prognostic *predictor, double dt,
                                           double *rhs)
{
    if (!ctx || !s || !predictor || !rhs) return BZ_ERR_INVALID;
    return bzi_poisson_source_fused(ctx, s, dt, rhs, predictor);
}

extern "C" int bz_project_and_diagnose_from(bz_ctx *ctx, const bz_state *s, const bz_prognostic *predictor,
                                            const double *phi_c, const double *phi_below, double dt)
{
    if (!ctx || !s || !predictor || !phi_c) return BZ_ERR_INVALID;
    if (!ctx->dg.wrap_y && !phi_below) return BZ_ERR_INVALID;
    return bzi_project_diagnose(ctx, s, dt, phi_c, phi_below, predictor);
}
