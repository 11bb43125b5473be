// bz_bounded.hip — bounds-preserving WENO for moisture-like scalars: advection = (; rho_q = WENO(order = 5, bounds = (lo, hi)))
//   Breeze side     /root/reference/src/Advection.jl:42-47 (div_rhoUc(..., ::BoundsPreservingWENO, rho, U, c) =
//                   V^-1 (bounded_tracer_flux_divergence_x + _y + _z)), used by /root/reference/examples/rico.jl:184-190 and
//                   /root/reference/examples/tropical_cyclone_world.jl:169
//   Oceananigans    bounded_tracer_flux_divergence_{x,y,z} (0.110.x, not vendored: PARITY UNPINNED, restated from the published
//                   positivity-preserving limiter; the CPU restatement used by the tests carries the same formulas): the two reconstructions
//                   that start in a cell (left-biased at its upper face, right-biased at its lower face) are pulled towards the cell
//                   mean by theta in [0, 1] so that the cell's Gauss-Lobatto point values stay inside [lo, hi]; fluxes use
//                   upwind_biased_product.  theta is a property of the cell, so face values are NOT shared between neighbours:
//                   four reconstructions per direction and cell (the plain scheme needs one per face).
#include <cstdlib>

#include "bz_internal.h"
#include "bz_weno.h"

#define TYB 4

__device__ __forceinline__ double bz_ubp(double u, double cl, double cr) { return ((u + fabs(u)) * cl + (u - fabs(u)) * cr) / 2.0; }

// divergence of one direction from the seven values c[-3..3] around the cell (v[3] = the cell), buffers Blo / Bhi of its two faces
__device__ __forceinline__ double bounded_div(const double *v, int Blo, int Bhi, double f_lo, double f_hi, double lo, double hi)
{
    const double w1 = 5.0 / 18.0, eps2 = 1e-20;
    const double cij = v[3];
    double cpL = bz_upB(v[1], v[2], v[3], v[4], v[5], v[6], true, Bhi);
    const double cpR = bz_upB(v[1], v[2], v[3], v[4], v[5], v[6], false, Bhi);
    const double cmL = bz_upB(v[0], v[1], v[2], v[3], v[4], v[5], true, Blo);
    double cmR = bz_upB(v[0], v[1], v[2], v[3], v[4], v[5], false, Blo);
    const double pt = (cij - w1 * cmR - w1 * cpL) / (1.0 - 2.0 * w1);
    const double M = fmax(pt, fmax(cpL, cmR));
    const double m = fmin(pt, fmin(cpL, cmR));
    const double th = fmin(fmin(fabs((hi - cij) / (M - cij + eps2)), fabs((lo - cij) / (m - cij + eps2))), 1.0);
    cpL = th * (cpL - cij) + cij;
    cmR = th * (cmR - cij) + cij;
    return bz_ubp(f_hi, cpL, cpR) - bz_ubp(f_lo, cmL, cmR);
}

__global__ __launch_bounds__(64 * TYB) void k_scalar_tendency_bounded(DevGrid g, double *__restrict__ Gc, const double *__restrict__ u,
                                                                     const double *__restrict__ v, const double *__restrict__ w,
                                                                     const double *__restrict__ c, int kchunk, double lo, double hi)
{
    const int i = blockIdx.x * 64 + threadIdx.x;
    const int j = blockIdx.y * TYB + threadIdx.y;
    if (i >= g.Nx || j >= g.Ny) return;
    const int k0 = blockIdx.z * kchunk;
    const int k1 = min(k0 + kchunk, g.Nz);
    const long long sy = g.Sx, sz = g.Sxy;
    long long n = g.idx(i, j, k0);
    double z[7];
#pragma unroll
    for (int s = 0; s < 6; ++s) z[s + 1] = c[n + (s - 3) * sz];
    for (int k = k0; k < k1; ++k, n += sz) {
#pragma unroll
        for (int s = 0; s < 6; ++s) z[s] = z[s + 1];
        z[6] = c[n + 3 * sz];
        const double rho = g.rho[k], Ax = g.Ax[k], Ay = g.Ay[k];
        double x[7];
#pragma unroll
        for (int s = 0; s < 7; ++s) x[s] = (s == 3) ? z[3] : c[n + (s - 3)];
        const double dxF = bounded_div(x, 3, 3, rho * (Ax * u[n]), rho * (Ax * u[n + 1]), lo, hi);
        double dyF = 0.0;
        if (!g.flat_y) {      // a Flat y direction has no faces (Ny = 1, Hy = 0: rows +-1 would be other z levels)
            double y[7];
#pragma unroll
            for (int s = 0; s < 7; ++s) y[s] = (s == 3) ? z[3] : c[n + (s - 3) * sy];
            dyF = bounded_div(y, 3, 3, rho * (Ay * v[n]), rho * (Ay * v[n + sy]), lo, hi);
        }
        const double dzF = bounded_div(z, bz_buffer_face(k, g.Nz), bz_buffer_face(k + 1, g.Nz), g.rho_f[k] * (g.Az * w[n]),
                                       g.rho_f[k + 1] * (g.Az * w[n + sz]), lo, hi);
        Gc[n] = -(g.Vinv_c[k] * (dxF + dyF + dzF));
    }
}

extern "C" int bz_set_bounds_preserving_advection(bz_ctx *ctx, const bz_bounds_preserving_advection *b)
{
    BZ_REJECT_BOUNDED_Y(ctx, b != nullptr, "bz_set_bounds_preserving_advection");
    if (ctx) ++ctx->config_epoch;      // captured steps (bz_graph.hip) belong to one configuration
    if (!ctx) return BZ_ERR_INVALID;
    if (!b) { ctx->bounded_mask = 0; return BZ_OK; }
    if (!(b->upper > b->lower)) return BZ_ERR_INVALID;
    if (ctx->compressible) {
        ctx->last_error = "bz_set_bounds_preserving_advection: implemented for the anelastic model";
        return BZ_ERR_UNSUPPORTED;
    }
#ifdef BZ_CENTERED2
    ctx->last_error = "bounds-preserving advection needs the WENO build";
    return BZ_ERR_UNSUPPORTED;
#endif
    ctx->bounded_mask = (b->moisture ? 1 : 0) | (b->microphysical_species ? 2 : 0) | (b->tracers ? 4 : 0);
    ctx->bounded_lo = b->lower;
    ctx->bounded_hi = b->upper;
    return BZ_OK;
}

// overwrite the tendencies of the flagged scalars with the bounds-preserving divergence (called at the end of the advective part
// of bz_compute_tendencies, before closure / forcing terms are added)
int bzi_bounded_tendencies(bz_ctx *ctx, const bz_state *s, const bz_prognostic *G)
{
    if (!ctx->bounded_mask) return BZ_OK;
    const DevGrid &g = ctx->dg;
    ProfileScope ps(ctx, "bounded_scalar_tendencies");
    long long tiles = (long long)((g.Nx + 63) / 64) * ((g.Ny + TYB - 1) / TYB);
    long long want = (4096 + tiles - 1) / tiles, maxchunks = g.Nz / 8 > 0 ? g.Nz / 8 : 1;
    if (want > maxchunks) want = maxchunks;
    if (want < 1) want = 1;
    const int kc = (int)((g.Nz + want - 1) / want);
    dim3 block(64, TYB), grid((g.Nx + 63) / 64, (g.Ny + TYB - 1) / TYB, (g.Nz + kc - 1) / kc);
    auto run = [&](double *Gc, const double *c) {
        hipLaunchKernelGGL(k_scalar_tendency_bounded, grid, block, 0, ctx->stream, g, Gc, s->u, s->v, s->w, c, kc, ctx->bounded_lo,
                           ctx->bounded_hi);
    };
    if (ctx->bounded_mask & 1) run(G->rho_q, s->q);
    if ((ctx->bounded_mask & 2) && g.microphysics == 2) {
        run(ctx->kessler.G_cloud_liquid_density, ctx->kessler.cloud_liquid_mass_fraction);
        run(ctx->kessler.G_rain_density, ctx->kessler.rain_mass_fraction);
    }
    if (ctx->bounded_mask & 4)
        for (int t = 0; t < ctx->n_tracers; ++t) run(ctx->tracers[t].G, ctx->tracers[t].specific);
    BZ_LAUNCH_CHECK();
    return BZ_OK;
}
