// bz_tendency3.hip — launcher of the third-generation tendency kernels (bz_tendency3_kernels.h).
// tools/tendbench (512x512x256, random data, MI355X): scalar 19.3 -> 16.6 ps/cell, u 19.4 -> 17.4,
// v 19.7 -> 19.9 (R=2), w 25.5 -> 27.1: the scalar and u/v kernels ship in this form, w stays gen-1.
#include <cstdlib>

#include "bz_tendency3_kernels.h"
#include "bz_tendency4_kernels.h"

#define T3_TYW 4

static int pick_chunk3(const DevGrid &g, int nlev, int rows_per_block)
{
    long long tiles = (long long)((g.Nx + 63) / 64) * ((g.Ny + rows_per_block - 1) / rows_per_block);
    long long want = (4096 + tiles - 1) / tiles;
    long long maxchunks = nlev / 64 > 0 ? nlev / 64 : 1;     // >= 64 levels per chunk amortises the batched edge flux
    if (want > maxchunks) want = maxchunks;
    if (want < 1) want = 1;
    return (int)((nlev + want - 1) / want);
}

// LDS-tiled kernels amortise a per-block prologue (tile + ring fill): prefer chunks of >= 128 levels while keeping
// >= 2 blocks per CU in flight
static int pick_chunk_lds(const DevGrid &g, int nlev, int rows_per_block)
{
    long long tiles = (long long)((g.Nx + 63) / 64) * ((g.Ny + rows_per_block - 1) / rows_per_block);
    long long want = (1024 + tiles - 1) / tiles;
    long long maxchunks = nlev / 128 > 0 ? nlev / 128 : 1;
    if (want > maxchunks) want = maxchunks;
    if (want < 1) want = 1;
    // small grids (BOMEX 256 x 256 x 128: 128 tiles; the 168 x 168 x 40 supercell box: 63): filling the 256 CUs matters more
    // than the prologue, go down to 8-level chunks until there are two blocks per CU (measured: supercell box 3.3 -> 2.5 ms/step,
    // BOMEX 128 x 128 x 96 2.6 -> 1.7 ms/step against a 32-level floor; 512^3 is not affected)
    if (tiles * want < 512) {
        long long fill = (512 + tiles - 1) / tiles;
        const int minlev = 8;
        long long cap = nlev / minlev > 0 ? nlev / minlev : 1;
        if (fill > cap) fill = cap;
        if (fill > want) want = fill;
    }
    return (int)((nlev + want - 1) / want);
}

template <int KIND, int R>
static void launch3(bz_ctx *ctx, const Tend3Fields &F)
{
    const DevGrid &g = ctx->dg;
    const int nlev = (KIND == T3_W) ? g.Nz - 1 : g.Nz;
    const int kc = pick_chunk3(g, nlev, R * T3_TYW);
    dim3 block(64, T3_TYW), grid((g.Nx + 63) / 64, (g.Ny + R * T3_TYW - 1) / (R * T3_TYW), (nlev + kc - 1) / kc);
    hipLaunchKernelGGL((k_tend3<KIND, R, T3_TYW>), grid, block, 0, ctx->stream, g, F, kc);
}


int bzi_compute_tendencies3(bz_ctx *ctx, const bz_state *s, const bz_prognostic *G, bool include_w)
{
    Tend3Fields F;
    F.ru = s->rho_u; F.rv = s->rho_v; F.rw = s->rho_w;
    F.u = s->u; F.v = s->v; F.w = s->w; F.T = s->T; F.q = s->q;
    {
        ProfileScope ps(ctx, "x_momentum_tendency");
        F.c = s->u; F.G = G->rho_u;
        launch3<T3_U, 1>(ctx, F);
    }
    {
        ProfileScope ps(ctx, "y_momentum_tendency");
        F.c = s->v; F.G = G->rho_v;
        launch3<T3_V, 1>(ctx, F);
    }
    if (include_w) {
        int rc = bzi_w_tendency_ring(ctx, s, G);
        if (rc) return rc;
    }
    {
        int rc = bzi_scalar_pair_tendency(ctx, s, G);
        if (rc) return rc;
    }
    BZ_LAUNCH_CHECK();
    return BZ_OK;
}

// potential temperature + moisture in one pass (k_scalar_pair)
// With an RK epilogue (E != nullptr) rho_theta and rho_q are advanced in place and, in the first stage, U0 is filled.
int bzi_scalar_pair_tendency(bz_ctx *ctx, const bz_state *s, const bz_prognostic *G, const bz_prognostic *U0,
                             const RKEpilogue *Ein)
{
    const DevGrid &g = ctx->dg;
    ProfileScope ps(ctx, Ein ? "scalar_tendencies+rk3" : "scalar_tendencies");
    const bool gen3 = false;      // the generation-3 pair kernel stays for reference; the LDS-tiled one is what runs
    const int TYP = gen3 ? T3_TYW : 8;
    const int kc = gen3 ? pick_chunk3(g, g.Nz, T3_TYW) : pick_chunk_lds(g, g.Nz, 8);
    dim3 block(64, TYP), grid((g.Nx + 63) / 64, (g.Ny + TYP - 1) / TYP, (g.Nz + kc - 1) / kc);
    RKEpilogue E;
    double *outa = G->rho_theta, *outb = G->rho_q;
    if (Ein) {
        E = *Ein;
        E.u0 = U0->rho_theta; E.u0_out = U0->rho_theta; E.u0b = U0->rho_q; E.u0b_out = U0->rho_q;
        outa = s->rho_theta; outb = s->rho_q;
    }
    if (gen3)
        hipLaunchKernelGGL((k_scalar_pair<T3_TYW>), grid, block, 0, ctx->stream, g, s->u, s->v, s->w, s->theta, s->q,
                           outa, outb, kc);
    else
        hipLaunchKernelGGL((k_scalar_pair_lds<8>), grid, block, 0, ctx->stream, g, s->u, s->v, s->w, s->theta, s->q,
                           outa, outb, kc, E, s->rho_theta, s->rho_q);
    BZ_LAUNCH_CHECK();
    return BZ_OK;
}

// z-momentum tendency with register rings for every vertical stencil (k_w_tend_ring)
int bzi_w_tendency_ring(bz_ctx *ctx, const bz_state *s, const bz_prognostic *G, const bz_prognostic *U0,
                        const RKEpilogue *Ein, int buoyancy_mode)
{
    const DevGrid &g = ctx->dg;
    if (bzi_k6_stored_ok(ctx)) return bzi_k6_stored(ctx, 2, s, G, U0, Ein, (buoyancy_mode == 0 && g.microphysics) ? 3 : buoyancy_mode);
    ProfileScope ps(ctx, Ein ? "z_momentum_tendency+rk3" : "z_momentum_tendency");
    RKEpilogue E;
    if (Ein) { E = *Ein; E.u0 = U0->rho_w; E.u0_out = U0->rho_w; }
    Tend3Fields F;
    F.ru = s->rho_u; F.rv = s->rho_v; F.rw = s->rho_w;
    F.u = s->u; F.v = s->v; F.w = s->w; F.T = s->T; F.q = s->q;
    F.c = s->w; F.G = G->rho_w;
    const int nlev = g.Nz - 1;
    if (buoyancy_mode == 0 && g.microphysics) {
        const int kc = pick_chunk_lds(g, nlev, 8);
        dim3 block(64, 8), grid((g.Nx + 63) / 64, (g.Ny + 7) / 8, (nlev + kc - 1) / kc);
        hipLaunchKernelGGL((k_w_tend_lds<8, 3>), grid, block, 0, ctx->stream, g, F, kc, E);
    } else if (buoyancy_mode != 0) {
        const int kc = pick_chunk_lds(g, nlev, 8);
        dim3 block(64, 8), grid((g.Nx + 63) / 64, (g.Ny + 7) / 8, (nlev + kc - 1) / kc);
        if (buoyancy_mode == 1) hipLaunchKernelGGL((k_w_tend_lds<8, 1>), grid, block, 0, ctx->stream, g, F, kc, E);
        else hipLaunchKernelGGL((k_w_tend_lds<8, 2>), grid, block, 0, ctx->stream, g, F, kc, E);
    } else if (ctx->tend_lds) {
        // in situ (512^3 bubble): ring 3.3 ms, LDS tile with 8 rows 2.9 ms, with 4 rows 3.9 ms per launch
        const int kc = pick_chunk_lds(g, nlev, 8);
        dim3 block(64, 8), grid((g.Nx + 63) / 64, (g.Ny + 7) / 8, (nlev + kc - 1) / kc);
        hipLaunchKernelGGL((k_w_tend_lds<8>), grid, block, 0, ctx->stream, g, F, kc, E);
    } else {
        const int kc = pick_chunk3(g, nlev, T3_TYW);
        dim3 block(64, T3_TYW), grid((g.Nx + 63) / 64, (g.Ny + T3_TYW - 1) / T3_TYW, (nlev + kc - 1) / kc);
        hipLaunchKernelGGL((k_w_tend_ring<T3_TYW>), grid, block, 0, ctx->stream, g, F, kc, E);
    }
    BZ_LAUNCH_CHECK();
    return BZ_OK;
}

// x-momentum tendency with the u y-stencil in an LDS tile (k_u_tend_lds)
int bzi_u_tendency_lds(bz_ctx *ctx, const bz_state *s, const bz_prognostic *G, const bz_prognostic *U0, const RKEpilogue *Ein)
{
    const DevGrid &g = ctx->dg;
    if (bzi_k6_stored_ok(ctx)) return bzi_k6_stored(ctx, 0, s, G, U0, Ein, 0);
    ProfileScope ps(ctx, Ein ? "x_momentum_tendency+rk3" : "x_momentum_tendency");
    RKEpilogue E;
    if (Ein) { E = *Ein; E.u0 = U0->rho_u; E.u0_out = U0->rho_u; }
    Tend3Fields F;
    F.ru = s->rho_u; F.rv = s->rho_v; F.rw = s->rho_w;
    F.u = s->u; F.v = s->v; F.w = s->w; F.T = s->T; F.q = s->q;
    F.c = s->u; F.G = G->rho_u;
    const int kc = pick_chunk_lds(g, g.Nz, 8);
    dim3 block(64, 8), grid((g.Nx + 63) / 64, (g.Ny + 7) / 8, (g.Nz + kc - 1) / kc);
    hipLaunchKernelGGL((k_u_tend_lds<8>), grid, block, 0, ctx->stream, g, F, kc, E);
    BZ_LAUNCH_CHECK();
    return BZ_OK;
}

// y-momentum tendency with every y-stencil in LDS tiles (k_v_tend_lds)
int bzi_v_tendency_lds(bz_ctx *ctx, const bz_state *s, const bz_prognostic *G, const bz_prognostic *U0, const RKEpilogue *Ein)
{
    const DevGrid &g = ctx->dg;
    if (bzi_k6_stored_ok(ctx)) return bzi_k6_stored(ctx, 1, s, G, U0, Ein, 0);
    ProfileScope ps(ctx, Ein ? "y_momentum_tendency+rk3" : "y_momentum_tendency");
    RKEpilogue E;
    if (Ein) { E = *Ein; E.u0 = U0->rho_v; E.u0_out = U0->rho_v; }
    Tend3Fields F;
    F.ru = s->rho_u; F.rv = s->rho_v; F.rw = s->rho_w;
    F.u = s->u; F.v = s->v; F.w = s->w; F.T = s->T; F.q = s->q;
    F.c = s->v; F.G = G->rho_v;
    const int kc = pick_chunk_lds(g, g.Nz, 8);
    dim3 block(64, 8), grid((g.Nx + 63) / 64, (g.Ny + 7) / 8, (g.Nz + kc - 1) / kc);
    hipLaunchKernelGGL((k_v_tend_lds<8>), grid, block, 0, ctx->stream, g, F, kc, E);
    BZ_LAUNCH_CHECK();
    return BZ_OK;
}
