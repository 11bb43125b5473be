// bz_tendency5.hip — launcher of the lean (prognostic-only) tendency kernels of the whole-step seam
// (bz_tendency5_kernels.h) and of their dry Exner table.
#include <cstdlib>

#include "bz_tendency5_kernels.h"

__global__ void k_pi_dry(DevGrid g, double *__restrict__ pi, int n)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    const int k = t - g.Hz;
    // the expression of k_project_diagnose<0> / buoyancy5 for q = 0: qd = 1, Rm = Rd, cpm = cpd exactly
    const double q = 0.0, qd = 1.0 - q;
    const double Rm = qd * g.Rd + q * g.Rv;
    const double cpm = qd * g.cpd + q * g.cpv;
    pi[t] = pow(g.p_r[k] / g.pst, Rm / cpm);
}

// the packed per-level rows of the lean kernels (LevRow5), copied from the column tables and the dry Exner table
__global__ void k_lev_rows(DevGrid g, const double *__restrict__ pi, LevRow5 *__restrict__ rows, int n)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    const int k = t - g.Hz;
    LevRow5 r;
    r.rho = g.rho[k]; r.rrho = g.rrho[k]; r.rho_f = g.rho_f[k]; r.rrho_f = g.rrho_f[k];
    r.Ax = g.Ax[k]; r.Ay = g.Ay[k]; r.Vinv_c = g.Vinv_c[k]; r.pi = pi[t];
    r.Vinv_f = g.Vinv_f[k]; r.T_r = g.T_r[k];
    for (int q = 0; q < 6; ++q) r.pad[q] = 0.0;
    rows[t] = r;
}

int bzi_lean_setup(bz_ctx *ctx)
{
    const DevGrid &g = ctx->dg;
    const int n = g.Nz + 2 * g.Hz;
    BZ_HIP(hipMalloc(&ctx->d_pi_dry, n * sizeof(double)));
    hipLaunchKernelGGL(k_pi_dry, dim3((n + 63) / 64), dim3(64), 0, 0, g, ctx->d_pi_dry, n);
    ctx->dg.pi_dry = ColPtr(ctx->d_pi_dry + g.Hz);      // bz_exner_factor (bz_internal.h) of every later launch
    BZ_HIP(hipMalloc(&ctx->d_lev_rows, n * sizeof(LevRow5)));
    hipLaunchKernelGGL(k_lev_rows, dim3((n + 63) / 64), dim3(64), 0, 0, g, ctx->d_pi_dry, (LevRow5 *)ctx->d_lev_rows, n);
    BZ_HIP(hipGetLastError());
    BZ_HIP(hipDeviceSynchronize());
    if (ctx->tune.side_cus > 0 && ctx->tune.side_cus < ctx->num_cus) {
        // a side stream that owns a share of the CUs: the scalar-pair kernel (issue-bound) then runs BESIDE the pressure solve (bandwidth-
        // bound) instead of in front of it — two unmasked streams simply take turns (measured: every kernel stretches by the other's time)
        uint32_t mask[32] = {0};
        const int total = ctx->num_cus, want = ctx->tune.side_cus;
        for (int b = 0; b < want; ++b) {
            const int bit = ctx->tune.side_cu_layout == 1 ? (int)((long long)b * total / want) : b;
            mask[bit >> 5] |= 1u << (bit & 31);
        }
        BZ_HIP(hipExtStreamCreateWithCUMask(&ctx->side_stream, (uint32_t)((total + 31) / 32), mask));
    } else
    BZ_HIP(hipStreamCreateWithFlags(&ctx->side_stream, hipStreamNonBlocking));
    BZ_HIP(hipEventCreateWithFlags(&ctx->ev_fork, hipEventDisableTiming));
    BZ_HIP(hipEventCreateWithFlags(&ctx->ev_join, hipEventDisableTiming));
    // single GPU: measured 47.4 -> 46.4 ms/step at 512^3 when on, but concurrent kernels make per-kernel durations (and the roofline
    // bookkeeping built on them) meaningless, so it is opt-in there; the distributed step turns it on whenever messages are in flight
    ctx->side_scalar = ctx->tune.side_scalar;
    ctx->lean = !ctx->tune.no_lean;
    ctx->lean_xcd = !ctx->tune.no_xcd;
    return BZ_OK;
}

void bzi_lean_teardown(bz_ctx *ctx)
{
    if (ctx->d_pi_dry) hipFree(ctx->d_pi_dry);
    ctx->d_pi_dry = nullptr;
    if (ctx->d_lev_rows) hipFree(ctx->d_lev_rows);
    ctx->d_lev_rows = nullptr;
    if (ctx->d_qstate) hipFree(ctx->d_qstate);
    ctx->d_qstate = nullptr;
    if (ctx->h_qstate) hipHostFree(ctx->h_qstate);
    ctx->h_qstate = nullptr;
    if (ctx->ev_q) hipEventDestroy(ctx->ev_q);
    ctx->ev_q = nullptr;
    if (ctx->side_stream) { hipStreamSynchronize(ctx->side_stream); hipStreamDestroy(ctx->side_stream); ctx->side_stream = nullptr; }
    if (ctx->ev_fork) hipEventDestroy(ctx->ev_fork);
    if (ctx->ev_join) hipEventDestroy(ctx->ev_join);
    ctx->ev_fork = ctx->ev_join = nullptr;
}

// z-chunking of the LDS-tiled kernels (same rule as pick_chunk_lds of bz_tendency3.hip)
// fine: chunks of >= 64 levels and >= 4096 blocks instead of >= 128 levels and >= 1024 blocks.  Measured at 512^3 in Float64 (chunk length
// 512 / 256 / 128 / 64 / 32 / 16): scalar pair 3.10 / 3.05 / 2.99 / 2.97 / 2.98 / 3.04 ms, x momentum 1.61 / 1.60 / 1.55 / 1.54 / 1.57 / 1.65,
// z momentum 2.03 / 2.00 / 1.99 / 1.98 / 1.99 / 2.09 — eight blocks per tile column keep every XCD's resident blocks closer together in z than
// two do; the 64 x 16 y-momentum kernel (1.69 / 1.70 / 1.70 / 1.73 / 1.76 / 1.81) and the Float32 build (equal to within noise) keep the coarse rule
static int pick_chunk5(const DevGrid &g, int nlev, int rows_per_block, bool fine = false)
{
    long long tiles = (long long)((g.Nx + 63) / 64) * ((g.Ny + rows_per_block - 1) / rows_per_block);
    const int floor_levels = fine ? 64 : 128;
    long long want = ((fine ? 4096 : 1024) + tiles - 1) / tiles;
    long long maxchunks = nlev / floor_levels > 0 ? nlev / floor_levels : 1;
    if (want > maxchunks) want = maxchunks;
    if (want < 1) want = 1;
    if (tiles * want < 512) {
        long long fill = (512 + tiles - 1) / tiles;
        long long cap = nlev / 8 > 0 ? nlev / 8 : 1;
        // latency-bound boxes (64^3: 64 blocks of 8 levels): the march is a chain of dependent level loads, so shorter chunks
        // finish sooner even though each re-reads the stencil levels below it
        if (tiles * cap < 256) cap = nlev / 2 > 0 ? nlev / 2 : 1;
        if (fill > cap) fill = cap;
        if (fill > want) want = fill;
    }
    return (int)((nlev + want - 1) / want);
}

// which: 1 the three momentum kernels, 2 the scalar-pair kernel, 3 all (the scalar kernel feeds nothing of the pressure solve, so the
// drivers may run it on a second stream beside the solve).
// rows: 0 all tile rows; 1 interior tile rows only (1 .. nty-2); 2 the two edge tile rows (0 and nty-1).  The slab driver runs
// the interior rows while the y-halo exchange of the stage-start state is in flight and the edge rows once it has landed.
template <int TY>
static int lean_launch(bz_ctx *ctx, const bz_state *s, const bz_prognostic *U0, const bz_prognostic *G, const double *pa,
                       const double *pb, double *oa, double *ob, double dt, double alpha, bool first, int rows, int which)
{
    const DevGrid &g = ctx->dg;
    RKEpilogue E;
    E.mode = first ? 3 : 2; E.dt = dt; E.alpha = alpha; E.oma = 1.0 - alpha;      // 3: first stage, U0 is the stage-start array itself: nothing stored
    Lean5 L;
    L.ru = s->rho_u; L.rv = s->rho_v; L.rw = s->rho_w; L.pa = pa; L.pb = pb; L.oa = oa; L.ob = ob; L.out = nullptr;
    L.T = s->T;
    L.pi_dry = ColPtr(ctx->d_pi_dry + g.Hz);
    L.lev = (const LevRow5 *)ctx->d_lev_rows + g.Hz;
    L.qstate = bzi_moisture_state(ctx);
    const int qh = L.qstate ? ctx->q_host : 2;      // the host's view of the scan's verdict (bz_step.hip); no word: the general bodies only
    L.mforce = 0; L.cor_f = 0.0; L.Fu = ColPtr(nullptr); L.Fv = ColPtr(nullptr); L.Su = ColPtr(nullptr); L.Sv = ColPtr(nullptr);
    if (ctx->has_forcings && bzi_lean_forcings_ok(ctx)) {      // the stack's momentum terms ride the RK epilogues of k6_u / k6_v
        const int m = ctx->forcing_static_mask;
        L.cor_f = ctx->forcing_f;
        if (m & 1) L.Fu = ColPtr(ctx->d_forcing);
        if (m & 2) L.Fv = ColPtr(ctx->d_forcing + (size_t)g.Nz);
        L.mforce = (ctx->forcing_f != 0.0 ? 1 : 0) | ((m & 1) ? 2 : 0) | ((m & 2) ? 4 : 0);
    }
    const dim3 block(64, TY);
    const int tx = (g.Nx + 63) / 64, nty = (g.Ny + TY - 1) / TY;
    if (rows && nty < 3) return rows == 1 ? BZ_OK : lean_launch<TY>(ctx, s, U0, G, pa, pb, oa, ob, dt, alpha, first, 0, which);
    const int ty = rows == 1 ? nty - 2 : rows == 2 ? 2 : nty;
    L.by0 = rows == 1 ? 1 : 0;
    L.bys = rows == 2 ? nty - 1 : 1;
    auto shape = [&](int nlev, int &kc) {
        kc = pick_chunk5(g, nlev, TY, sizeof(double) == 8);
        dim3 grid(tx, ty, (nlev + kc - 1) / kc);
        L.xcd = (ctx->lean_xcd && ((long long)grid.x * grid.y * grid.z) % 8 == 0) ? 1 : 0;
        return grid;
    };
    int kc;
    if (which & 1) {
        ProfileScope ps(ctx, "x_momentum_tendency+rk3+velocity");
        E.u0 = U0->rho_u; E.u0_out = U0->rho_u;
        L.out = G->rho_u;
        const dim3 grid = shape(g.Nz, kc);
        if (g.bounded_y) {      // walls in y: the WY instantiations (row-wise buffers)
            if (L.mforce) hipLaunchKernelGGL((k6_u<TY, true, true>), grid, block, 0, ctx->stream, g, L, kc, E);
            else hipLaunchKernelGGL((k6_u<TY, false, true>), grid, block, 0, ctx->stream, g, L, kc, E);
        } else if (L.mforce) hipLaunchKernelGGL((k6_u<TY, true>), grid, block, 0, ctx->stream, g, L, kc, E);
        else hipLaunchKernelGGL((k6_u<TY, false>), grid, block, 0, ctx->stream, g, L, kc, E);
    }
    if (which & 1) {
        ProfileScope ps(ctx, "y_momentum_tendency+rk3+velocity");
        E.u0 = U0->rho_v; E.u0_out = U0->rho_v;
        L.out = G->rho_v;
        // 64 x 16 tiles for the y-momentum kernel in Float64 (one 1024-thread workgroup per CU): its y reconstructions read five frame rows
        // of v per interior row, the tallest frame of the four kernels — measured 1.77 -> 1.68 ms per launch at 512^3; the other three
        // kernels are equal to within noise (x 1.59 -> 1.58, z 2.04 -> 2.05, scalars 3.09 -> 3.08) and stay on 64 x 8, as does Float32
        // (1.06 -> 1.05).  Single-device contexts with whole tile rows (the slab driver's row split counts 8-row tiles; walls keep the WY instantiations).
        if (sizeof(double) == 8 && TY == 8 && rows == 0 && !ctx->slab_mode && !g.bounded_y && g.Ny % 16 == 0 && g.Ny >= 64) {
            const int kc16 = pick_chunk5(g, g.Nz, 16);
            const dim3 grid16(tx, g.Ny / 16, (g.Nz + kc16 - 1) / kc16), block16(64, 16);
            L.xcd = (ctx->lean_xcd && ((long long)grid16.x * grid16.y * grid16.z) % 8 == 0) ? 1 : 0;
            if (L.mforce) hipLaunchKernelGGL((k6_v<16, true>), grid16, block16, 0, ctx->stream, g, L, kc16, E);
            else hipLaunchKernelGGL((k6_v<16, false>), grid16, block16, 0, ctx->stream, g, L, kc16, E);
        } else {
        const dim3 grid = shape(g.Nz, kc);
        if (g.bounded_y) {
            if (L.mforce) hipLaunchKernelGGL((k6_v<TY, true, true>), grid, block, 0, ctx->stream, g, L, kc, E);
            else hipLaunchKernelGGL((k6_v<TY, false, true>), grid, block, 0, ctx->stream, g, L, kc, E);
        } else if (L.mforce) hipLaunchKernelGGL((k6_v<TY, true>), grid, block, 0, ctx->stream, g, L, kc, E);
        else hipLaunchKernelGGL((k6_v<TY, false>), grid, block, 0, ctx->stream, g, L, kc, E);
        }
    }
    if (which & 1) {
        ProfileScope ps(ctx, "z_momentum_tendency+rk3+velocity");
        E.u0 = U0->rho_w; E.u0_out = U0->rho_w;
        L.out = G->rho_w;
        const dim3 grid = shape(g.Nz - 1, kc);
        // dry and general body are separate kernels (see bz_lean_skip): general alone once the host knows the model is moist (or has no word)
#define BZ_LAUNCH_DRY_GENERAL(KERNEL, WYV)                                                                                                  \
        do {                                                                                                                                \
            if (qh != 2) {                                                                                                                  \
                hipLaunchKernelGGL((KERNEL<TY, WYV, true, true>), grid, block, 0, ctx->stream, g, L, kc, E);                                \
                hipLaunchKernelGGL((KERNEL<TY, WYV, false, true>), grid, block, 0, ctx->stream, g, L, kc, E);                               \
            } else hipLaunchKernelGGL((KERNEL<TY, WYV, false, false>), grid, block, 0, ctx->stream, g, L, kc, E);                           \
        } while (0)
        if (g.bounded_y) BZ_LAUNCH_DRY_GENERAL(k6_w, true);
        else BZ_LAUNCH_DRY_GENERAL(k6_w, false);
    }
    if (which & 2) {
        ProfileScope ps(ctx, "scalar_tendencies+rk3+thermo");
        E.u0 = U0->rho_theta; E.u0_out = U0->rho_theta; E.u0b = U0->rho_q; E.u0b_out = U0->rho_q;
        L.out = nullptr;
        const dim3 grid = shape(g.Nz, kc);
        if (g.bounded_y) BZ_LAUNCH_DRY_GENERAL(k5_scalar_pair, true);
        else BZ_LAUNCH_DRY_GENERAL(k5_scalar_pair, false);
    }
    BZ_LAUNCH_CHECK();
    return BZ_OK;
}

// ---- stored-velocity instantiations of the sixth-generation momentum kernels (round 5) -------------------------------------------------
// The tiers that keep u, v, w as stored fields — the fused-RK tier (saturation adjustment, closures, tracers, forcing stacks: BASELINE
// configs[2]), the per-operator entry points and the slow tendencies of the compressible model (configs[4]) — ran the fourth-generation
// tiles (k_{u,v,w}_tend_lds: x stencils from global memory, loads at their point of use, hardware block order; 0.29 of the roof on the
// compressible slow tendencies).  They now launch k6_u / k6_v / k6_w<ST>: the same kernels as the lean seam with
// the velocity read instead of derived.  comp: 0 u, 1 v, 2 w;  bm: buoyancy mode of the z-momentum kernel (k6_w).
bool bzi_k6_stored_ok(const bz_ctx *ctx)
{
    const DevGrid &g = ctx->dg;
    return !ctx->tune.no_k6_stored && !g.flat_y && !g.bounded_x && !g.bounded_y && g.Nz > 1 &&
           (long long)g.Sxy * (g.Nz + 2 * g.Hz + 1) < (1LL << 32);
}

int bzi_k6_stored(bz_ctx *ctx, int comp, const bz_state *s, const bz_prognostic *G, const bz_prognostic *U0, const RKEpilogue *Ein, int bm)
{
    constexpr int TY = 8;
    const DevGrid &g = ctx->dg;
    RKEpilogue E;
    if (Ein) E = *Ein;
    Lean5 L;
    L.ru = s->rho_u; L.rv = s->rho_v; L.rw = s->rho_w; L.pa = L.pb = nullptr; L.oa = L.ob = nullptr; L.T = nullptr;
    L.pi_dry = ColPtr(nullptr);
    L.lev = nullptr;
    L.qstate = nullptr;
    L.mforce = 0; L.cor_f = 0.0; L.Fu = ColPtr(nullptr); L.Fv = ColPtr(nullptr); L.Su = ColPtr(nullptr); L.Sv = ColPtr(nullptr);
    if (ctx->fold_momentum_forcing && Ein && comp < 2) {
        // fused-RK tier (bz_step.hip): Coriolis, the static u / v profiles and the subsidence profiles of the stage ride the RK epilogues
        // of the x / y momentum kernels (terms and order of k_apply_forcings); the forcing pass that follows touches the scalars only
        const int m = ctx->forcing_static_mask, sm = ctx->forcing_subsidence_mask;
        const double *sub = ctx->d_forcing + (size_t)5 * g.Nz + (g.Nz + 1) + (size_t)4 * g.Nz;      // 4 x Nz subsidence profiles (bz_forcing.hip)
        L.cor_f = ctx->forcing_f;
        if (m & 1) L.Fu = ColPtr(ctx->d_forcing);
        if (m & 2) L.Fv = ColPtr(ctx->d_forcing + (size_t)g.Nz);
        if (sm) { L.Su = ColPtr(sub); L.Sv = ColPtr(sub + (size_t)g.Nz); }      // (the columns of inactive fields hold zeros; k_apply_forcings adds them too)
        L.mforce = (ctx->forcing_f != 0.0 ? 1 : 0) | ((m & 1) ? 2 : 0) | ((m & 2) ? 4 : 0) | (sm ? 8 | 16 : 0);
    }
    L.bT = s->T; L.bq = s->q;
    L.by0 = 0; L.bys = 1;
    const dim3 block(64, TY);
    const int tx = (g.Nx + 63) / 64, nty = (g.Ny + TY - 1) / TY;
    const int nlev = comp == 2 ? g.Nz - 1 : g.Nz;
    const int kc = pick_chunk5(g, nlev, TY, sizeof(double) == 8);
    const dim3 grid(tx, nty, (nlev + kc - 1) / kc);
    L.xcd = (!ctx->tune.no_xcd && ((long long)grid.x * grid.y * grid.z) % 8 == 0) ? 1 : 0;
    if (comp == 0) {
        ProfileScope ps(ctx, Ein ? "x_momentum_tendency+rk3" : "x_momentum_tendency");
        if (Ein) { E.u0 = U0->rho_u; E.u0_out = U0->rho_u; }
        L.vel = s->u; L.out = G->rho_u;
        if (L.mforce) hipLaunchKernelGGL((k6_u<TY, true, false, true>), grid, block, 0, ctx->stream, g, L, kc, E);
        else hipLaunchKernelGGL((k6_u<TY, false, false, true>), grid, block, 0, ctx->stream, g, L, kc, E);
    } else if (comp == 1) {
        ProfileScope ps(ctx, Ein ? "y_momentum_tendency+rk3" : "y_momentum_tendency");
        if (Ein) { E.u0 = U0->rho_v; E.u0_out = U0->rho_v; }
        L.vel = s->v; L.out = G->rho_v;
        if (L.mforce) hipLaunchKernelGGL((k6_v<TY, true, false, true>), grid, block, 0, ctx->stream, g, L, kc, E);
        else hipLaunchKernelGGL((k6_v<TY, false, false, true>), grid, block, 0, ctx->stream, g, L, kc, E);
    } else {
        ProfileScope ps(ctx, Ein ? "z_momentum_tendency+rk3" : "z_momentum_tendency");
        if (Ein) { E.u0 = U0->rho_w; E.u0_out = U0->rho_w; }
        L.vel = s->w; L.out = G->rho_w;
        if (bm == 0) hipLaunchKernelGGL((k6_w<TY, false, false, false, 0>), grid, block, 0, ctx->stream, g, L, kc, E);
        else if (bm == 1) hipLaunchKernelGGL((k6_w<TY, false, false, false, 1>), grid, block, 0, ctx->stream, g, L, kc, E);
        else if (bm == 2) hipLaunchKernelGGL((k6_w<TY, false, false, false, 2>), grid, block, 0, ctx->stream, g, L, kc, E);
        else hipLaunchKernelGGL((k6_w<TY, false, false, false, 3>), grid, block, 0, ctx->stream, g, L, kc, E);
    }
    BZ_LAUNCH_CHECK();
    return BZ_OK;
}

int bzi_tendencies_lean(bz_ctx *ctx, const bz_state *s, const bz_prognostic *U0, const bz_prognostic *G, const double *pa,
                        const double *pb, double *oa, double *ob, double dt, double alpha, bool first, int rows, int which)
{
    // 64 x 8 tiles (64 x 4 measured +8 % on the momentum kernels: more frame cells per interior cell; 64 x 16 pays for k6_v alone, see there)
    return lean_launch<8>(ctx, s, U0, G, pa, pb, oa, ob, dt, alpha, first, rows, which);
}
