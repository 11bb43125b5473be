// bz_context.hip — context lifetime, column tables, stream and instrumentation.
#include <cmath>
#include <cstdlib>
#include <cstring>

#include "bz_internal.h"

ProfileScope::ProfileScope(bz_ctx *c, const char *name) : ctx(c)
{
    if (!ctx->profiling || ctx->profile_mute) return;
    for (size_t s = 0; s < ctx->slots.size(); ++s)
        if (ctx->slots[s].name == name || std::strcmp(ctx->slots[s].name, name) == 0) slot = (int)s;
    if (slot < 0) {
        ctx->slots.emplace_back();
        ctx->slots.back().name = name;
        slot = (int)ctx->slots.size() - 1;
    }
    if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) { slot = -1; return; }
    hipEventRecord(e0, ctx->stream);
}
ProfileScope::~ProfileScope()
{
    if (slot < 0) return;
    hipEventRecord(e1, ctx->stream);
    ctx->slots[slot].pending.emplace_back(e0, e1);
}

static void profile_drain(bz_ctx *ctx)
{
    for (auto &s : ctx->slots) {
        for (auto &p : s.pending) {
            hipEventSynchronize(p.second);
            float ms = 0.f;
            if (hipEventElapsedTime(&ms, p.first, p.second) == hipSuccess) {
                s.total_ms += ms;
                s.launches += 1;
            }
            hipEventDestroy(p.first);
            hipEventDestroy(p.second);
        }
        s.pending.clear();
    }
}

void bzi_read_tuning(bz_tuning &t)
{
    auto on = [](const char *n) { return getenv(n) != nullptr; };
    auto num = [](const char *n, int d) { const char *e = getenv(n); return e ? atoi(e) : d; };
    t.no_fused = on("BZ_NO_FUSED");
    t.no_fuse_rk = on("BZ_NO_FUSE_RK");
    t.no_tend_lds = on("BZ_NO_TEND_LDS");
    t.tend_gen = num("BZ_TEND_GEN", 0);
    t.no_lean = on("BZ_NO_LEAN");
    t.no_xcd = on("BZ_NO_XCD");
    t.no_k6_stored = on("BZ_NO_K6_STORED");
    t.no_closure_march = on("BZ_NO_CLOSURE_MARCH");
    t.no_dry_shortcut = on("BZ_NO_DRY_SHORTCUT");
    t.side_scalar = on("BZ_SIDE_SCALAR");
    t.side_cus = num("BZ_SIDE_CUS", 0);
    t.side_cu_layout = num("BZ_SIDE_CU_LAYOUT", 0);
    t.no_fuse_forcing = on("BZ_NO_FUSE_FORCING");
    t.no_fold_forcing = on("BZ_NO_FOLD_FORCING");
    t.no_fuse_level_sums = on("BZ_NO_FUSE_LEVEL_SUMS");
    t.ac_xcd = num("BZ_AC_XCD", 1);
    t.ac_forward2 = num("BZ_AC_FWD2", 1);
    t.ac_pfold = num("BZ_AC_PFOLD", 1);
    t.ac_init_fold = num("BZ_AC_INIT_FOLD", 1);
    t.ac_pair_avg = num("BZ_AC_PAIR_AVG", 1);
    t.ac_rotate = num("BZ_AC_ROTATE", 1);
    t.ac_cfg = num("BZ_AC_CFG", 29);
    t.ac_bx = num("BZ_AC_BX", 128);
    t.no_tridiag_coop = on("BZ_NO_TRIDIAG_COOP");
    t.no_xfft = on("BZ_NO_XFFT");
    t.poisson_kxmajor = num("BZ_POISSON_KXMAJOR", 1);
    t.poisson_kx_chunk_mb = num("BZ_POISSON_KX_CHUNK_MB", 0);
    t.poisson_kx_pad = num("BZ_POISSON_KX_PAD", 1);
    t.poisson_kx_chunk_kb = num("BZ_POISSON_KX_CHUNK_KB", 0);
    t.poisson_chunk = num("BZ_POISSON_CHUNK", 0);
    t.xf_kchunk_f = num("BZ_XF_KCHUNK_F", 0);
    t.xf_kchunk_i = num("BZ_XF_KCHUNK_I", 0);
    t.generic_onepass = on("BZ_GENERIC_ONEPASS");
    t.no_generic_march = on("BZ_NO_GENERIC_MARCH");
    t.no_rho3d_exchange = on("BZ_NO_RHO3D_EXCHANGE");
    t.scalar_lds = num("BZ_SCALAR_LDS", 1);
    t.no_ac_fuse = on("BZ_NO_AC_FUSE");
    t.no_ac_end_fuse = on("BZ_NO_AC_END_FUSE");
    t.comm_no_overlap = on("BZ_COMM_NO_OVERLAP");
    t.comm_self_messages = on("BZ_COMM_SELF_MESSAGES");
    t.comm_no_side_scalar = on("BZ_COMM_NO_SIDE_SCALAR");
    t.graph = num("BZ_GRAPH", -1);
    t.graph_debug = on("BZ_GRAPH_DEBUG");
}

extern "C" int bz_profile_enable(bz_ctx *ctx, int on)
{
    if (!ctx) return BZ_ERR_INVALID;
    if (!on) profile_drain(ctx);
    ctx->profiling = on != 0;
    return BZ_OK;
}
extern "C" int bz_profile_reset(bz_ctx *ctx)
{
    if (!ctx) return BZ_ERR_INVALID;
    profile_drain(ctx);
    for (auto &s : ctx->slots) { s.total_ms = 0.0; s.launches = 0; }
    return BZ_OK;
}
extern "C" int bz_profile_count(bz_ctx *ctx) { return ctx ? (int)ctx->slots.size() : 0; }
extern "C" int bz_profile_get(bz_ctx *ctx, int idx, const char **name, double *total_ms, int64_t *launches)
{
    if (!ctx || idx < 0 || idx >= (int)ctx->slots.size()) return BZ_ERR_INVALID;
    profile_drain(ctx);
    if (name) *name = ctx->slots[idx].name;
    if (total_ms) *total_ms = ctx->slots[idx].total_ms;
    if (launches) *launches = ctx->slots[idx].launches;
    return BZ_OK;
}

extern "C" const char *bz_last_error(const bz_ctx *ctx) { return ctx ? ctx->last_error.c_str() : "null ctx"; }

// every launch of the library goes to ctx->stream; rocFFT executes on the stream its plan carries
int bzi_apply_stream(bz_ctx *ctx, hipStream_t stream)
{
    ctx->stream = stream;
    if (ctx->plans_ok) {
        BZ_FFT(hipfftSetStream(ctx->plan_fwd, ctx->stream));
        BZ_FFT(hipfftSetStream(ctx->plan_inv, ctx->stream));
    }
    if (ctx->xf && ctx->plan_y) BZ_FFT(hipfftSetStream(ctx->plan_y, ctx->stream));      // the y transforms of the LDS x-transform pipeline
    if (ctx->plan_yc) BZ_FFT(hipfftSetStream(ctx->plan_yc, ctx->stream));
    if (ctx->plan_yc_last) BZ_FFT(hipfftSetStream(ctx->plan_yc_last, ctx->stream));
    if (ctx->pchunk) {
        BZ_FFT(hipfftSetStream(ctx->plan_fwd_c, ctx->stream));
        BZ_FFT(hipfftSetStream(ctx->plan_inv_c, ctx->stream));
    }
    if (ctx->slab_plans_ok) {
        BZ_FFT(hipfftSetStream(ctx->slab_plan_x_fwd, ctx->stream));
        BZ_FFT(hipfftSetStream(ctx->slab_plan_x_inv, ctx->stream));
        BZ_FFT(hipfftSetStream(ctx->slab_plan_y, ctx->stream));
    }
    return BZ_OK;
}

extern "C" int bz_set_stream(bz_ctx *ctx, void *hip_stream)
{
    if (!ctx) return BZ_ERR_INVALID;
    ++ctx->config_epoch;
    return bzi_apply_stream(ctx, (hipStream_t)hip_stream);
}

// thermodynamic formulation of the anelastic model: 0 = :LiquidIcePotentialTemperature, 1 = :StaticEnergy
// (src/StaticEnergyFormulations/static_energy_formulation.jl); with 1 the rho_theta / theta slots carry rho_e / e.
extern "C" int bz_set_formulation(bz_ctx *ctx, int formulation)
{
    if (ctx) ++ctx->config_epoch;      // captured steps (bz_graph.hip) belong to one configuration
    if (!ctx || formulation < 0 || formulation > 1) return BZ_ERR_INVALID;
    if (ctx->compressible && formulation != 0) return BZ_ERR_UNSUPPORTED;
    ctx->dg.formulation = formulation;
    return BZ_OK;
}

// microphysics = SaturationAdjustment(equilibrium = WarmPhaseEquilibrium(), solver = SecantSolver(abstol, maxiter))
// (src/Microphysics/saturation_adjustment.jl:20-60); params == NULL switches back to microphysics = nothing.
extern "C" int bz_set_saturation_adjustment(bz_ctx *ctx, const bz_saturation_adjustment *params, double *q_vapor,
                                            double *q_liquid)
{
    if (ctx) ++ctx->config_epoch;      // captured steps (bz_graph.hip) belong to one configuration
    if (!ctx) return BZ_ERR_INVALID;
    DevGrid &g = ctx->dg;
    if (!params) { g.microphysics = 0; g.qv_field = g.ql_field = nullptr; return BZ_OK; }
    if (!q_vapor || !q_liquid || params->maxiter < 0) return BZ_ERR_INVALID;
    if (g.formulation != 0) {
        ctx->last_error = "saturation adjustment is implemented for the potential-temperature formulation";
        return BZ_ERR_UNSUPPORTED;
    }
    g.microphysics = 1;
    g.sa_Ll = params->liquid_latent_heat;
    g.sa_cl = params->liquid_heat_capacity;
    g.sa_dc = ctx->constants.vapor_heat_capacity - params->liquid_heat_capacity;
    g.sa_L0 = params->liquid_latent_heat - g.sa_dc * params->energy_reference_temperature;
    g.sa_Ttr = params->triple_point_temperature;
    g.sa_ptr = params->triple_point_pressure;
    g.sa_abstol = params->abstol;
    g.sa_maxiter = params->maxiter;
    g.qv_field = q_vapor;
    g.ql_field = q_liquid;
    return BZ_OK;
}

extern "C" int bz_sync(bz_ctx *ctx)
{
    if (!ctx) return BZ_ERR_INVALID;
    // an undiagnosed last stage leaves its halo exchange on the side stream: a host that reads `s` after bz_sync must see it landed
    if (ctx->comm) { const int rc = bzi_comm_join_pending(ctx); if (rc) return rc; }
    BZ_HIP(hipStreamSynchronize(ctx->stream));
    return BZ_OK;
}

extern "C" int bz_create(bz_ctx **out, const bz_grid *grid, const bz_constants *constants,
                         const bz_reference_state *ref, int weno_order)
{
    return bzi_create(out, grid, constants, ref, weno_order, 1, 0, false);
}

extern "C" int bz_create_slab(bz_ctx **out, const bz_grid *local_grid, const bz_constants *constants,
                              const bz_reference_state *ref, int weno_order, int y_nranks, int y_rank)
{
    if (y_nranks < 1 || y_rank < 0 || y_rank >= y_nranks) return BZ_ERR_INVALID;
    return bzi_create(out, local_grid, constants, ref, weno_order, y_nranks, y_rank, true);
}

int bzi_create(bz_ctx **out, const bz_grid *grid, const bz_constants *constants, const bz_reference_state *ref,
               int weno_order, int y_nranks, int y_rank, bool slab_mode, bool compressible)
{
    if (!out || !grid || !constants || !ref || !grid->zf || !ref->density || !ref->pressure || !ref->temperature)
        return BZ_ERR_INVALID;
    *out = nullptr;
#ifdef BZ_CENTERED2
    if (weno_order != 2) return BZ_ERR_UNSUPPORTED;      // this build is Centered(order = 2)
#else
    // 5: the tuned path; 7, 9: generic kernels (bz_tendency_generic.hip), anelastic single-GPU contexts, halos >= (order + 1) / 2
    if (weno_order != 5 && weno_order != 7 && weno_order != 9) return BZ_ERR_UNSUPPORTED;
    
    const int weno_R = (weno_order + 1) / 2;
    if (grid->Hx < weno_R || (grid->topo[1] != BZ_FLAT && grid->Hy < weno_R) || grid->Hz < weno_R) return BZ_ERR_UNSUPPORTED;
#endif
    if (grid->ftype != 8) return BZ_ERR_UNSUPPORTED;
    // (Periodic, Periodic, Bounded), or (Periodic, Flat, Bounded) — the reference's 2-D x-z cases (README.md:67-75, examples/
    // dry_thermal_bubble.jl, acoustic_wave.jl, inertia_gravity_wave.jl): Ny = 1, Hy = 0, single-GPU WENO-5 contexts
    const bool flat_y = grid->topo[1] == BZ_FLAT;
    // (Periodic, Bounded, Bounded) — the reference benchmark driver's PBB option (benchmarking/run_benchmarks.jl:130): single-GPU
    // anelastic WENO-5 contexts, stepped operator by operator
    const bool bounded_y = grid->topo[1] == BZ_BOUNDED;
    // (Bounded, Flat, Bounded) — walls in x of a 2-D model (examples/cloudy_thermal_bubble.jl:20-24, tropical_cyclone_with_rainband.jl:294)
    const bool bounded_x = grid->topo[0] == BZ_BOUNDED;
    if ((grid->topo[0] != BZ_PERIODIC && !bounded_x) || (grid->topo[1] != BZ_PERIODIC && !flat_y && !bounded_y) || grid->topo[2] != BZ_BOUNDED)
        return BZ_ERR_UNSUPPORTED;
    // compressible contexts with lateral walls ((Bounded | Periodic, Bounded | Periodic, Bounded), 3-D, single device): the acoustic substep
    // loop with its wall / open-boundary enforcement (bz_compressible.hip: BZ_REJECT_WALLS lists what they do not run)
    if (compressible && (bounded_x || bounded_y)) {
        if (flat_y || slab_mode || weno_order == 2 || grid->Nx < 2 * grid->Hx || grid->Ny < 2 * grid->Hy) return BZ_ERR_UNSUPPORTED;
    } else {
    if (bounded_x && (!flat_y || compressible || weno_order == 2 || (grid->Nx & 1) || grid->Nx < 2 * grid->Hx || grid->Nx > 4096)) return BZ_ERR_UNSUPPORTED;
    if (bounded_y && (slab_mode || compressible || weno_order == 2 || grid->Ny < 2 * grid->Hy)) return BZ_ERR_UNSUPPORTED;
    }
    if (flat_y && (grid->Ny != 1 || grid->Hy != 0 || slab_mode)) return BZ_ERR_UNSUPPORTED;
    if (grid->Hx < 3 || (!flat_y && grid->Hy < 3) || grid->Hz < 3) return BZ_ERR_UNSUPPORTED;
    // Oceananigans: N >= H in every direction (k_halo_y's wrap copy would otherwise read a halo row that is not filled yet)
    if (grid->Nx < grid->Hx || grid->Ny < grid->Hy || grid->Nz < grid->Hz) return BZ_ERR_UNSUPPORTED;

    bz_ctx *ctx = new (std::nothrow) bz_ctx();
    if (ctx) bzi_read_tuning(ctx->tune);      // the only place the environment is read
    if (!ctx) return BZ_ERR_ALLOC;
    ctx->grid = *grid;
    ctx->grid.zf = nullptr;
    ctx->constants = *constants;
    ctx->y_nranks = y_nranks;
    ctx->y_rank = y_rank;
    ctx->slab_mode = slab_mode;
    ctx->Ny_global = grid->Ny * y_nranks;
    {
        int dev = 0, cus = 0;
        if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && cus > 0) ctx->num_cus = cus;
    }

    const int Nx = grid->Nx, Ny = grid->Ny, Nz = grid->Nz, Hz = grid->Hz;
    const int nc = Nz + 2 * Hz, nf = Nz + 1 + 2 * Hz;

    // ---- z metrics with halo extension (first/last spacing mirrored outward) ----
    std::vector<double> zf_ext(nf), zc_ext(nc), dzc(nc), dzf(nf, 0.0);
    for (int k = 0; k <= Nz; ++k) zf_ext[Hz + k] = grid->zf[k];
    for (int h = 1; h <= Hz; ++h) {
        zf_ext[Hz - h] = zf_ext[Hz - h + 1] - (grid->zf[1] - grid->zf[0]);
        zf_ext[Hz + Nz + h] = zf_ext[Hz + Nz + h - 1] + (grid->zf[Nz] - grid->zf[Nz - 1]);
    }
    for (int k = 0; k < nc; ++k) {
        zc_ext[k] = 0.5 * (zf_ext[k] + zf_ext[k + 1]);
        dzc[k] = zf_ext[k + 1] - zf_ext[k];
    }
    for (int k = 1; k < nc; ++k) dzf[k] = zc_ext[k] - zc_ext[k - 1];
    dzf[0] = dzf[1];
    dzf[nf - 1] = dzf[nf - 2];
    if (grid->regular_z) {
        double dz = (grid->zf[Nz] - grid->zf[0]) / Nz;
        for (auto &v : dzc) v = dz;
        for (auto &v : dzf) v = dz;
        for (int k = 0; k < nc; ++k) zc_ext[k] = grid->zf[0] + dz * ((k - Hz) + 0.5);   // Oceananigans regular-grid nodes
    }

    // ---- column tables: 11 columns of nf entries each ----
    enum { C_DZC, C_DZF, C_RDZF, C_AX, C_AY, C_VIC, C_VIF, C_RHO, C_RHOF, C_PR, C_TR, C_RDZC, C_ZC, C_RRHO, C_RRHOF, C_LNPI, C_COUNT };
    std::vector<double> cols((size_t)C_COUNT * nf, 0.0);
    auto col = [&](int c) { return cols.data() + (size_t)c * nf; };
    const double dx = grid->dx, dy = grid->dy;
    for (int k = 0; k < nc; ++k) {
        col(C_DZC)[k] = dzc[k];
        col(C_RDZC)[k] = 1.0 / dzc[k];
        col(C_ZC)[k] = zc_ext[k];
        col(C_AX)[k] = dy * dzc[k];
        col(C_AY)[k] = dx * dzc[k];
        col(C_VIC)[k] = 1.0 / (dx * dy * dzc[k]);
        col(C_RHO)[k] = ref->density[k];
        col(C_RRHO)[k] = 1.0 / ref->density[k];
        col(C_PR)[k] = ref->pressure[k];
        col(C_TR)[k] = ref->temperature[k];
        col(C_LNPI)[k] = std::log(ref->pressure[k] / ref->standard_pressure);
    }
    for (int k = 0; k < nf; ++k) {
        col(C_DZF)[k] = dzf[k];
        col(C_RDZF)[k] = 1.0 / dzf[k];
        col(C_VIF)[k] = 1.0 / (dx * dy * dzf[k]);
        // Iz(rho) at face k: 0.5*(rho[k-1]+rho[k]); defined where both exist
        if (k >= 1 && k < nc) {
            col(C_RHOF)[k] = 0.5 * (ref->density[k - 1] + ref->density[k]);
            col(C_RRHOF)[k] = 1.0 / col(C_RHOF)[k];
        }
    }

    {   // a failed allocation must not leak the half-built context
        hipError_t e = hipMalloc(&ctx->d_columns, cols.size() * sizeof(double));
        if (e == hipSuccess) e = hipMemcpy(ctx->d_columns, cols.data(), cols.size() * sizeof(double), hipMemcpyHostToDevice);
        if (e == hipSuccess) e = hipMalloc(&ctx->d_scalar, 64 * sizeof(double));
        if (e != hipSuccess) {
            fprintf(stderr, "bz_create: column tables: %s\n", hipGetErrorString(e));
            bz_destroy(ctx);
            return -(int)e;
        }
    }

    DevGrid &g = ctx->dg;
    g.Nx = Nx; g.Ny = Ny; g.Nz = Nz;
    g.Hx = grid->Hx; g.Hy = grid->Hy; g.Hz = Hz;
    g.Sx = Nx + 2 * grid->Hx;
    g.Sy = Ny + 2 * grid->Hy;
    g.Sxy = (long long)g.Sx * g.Sy;
    g.dx = dx; g.dy = dy; g.rdx = 1.0 / dx; g.Az = dx * dy;
    // a Flat direction has zero derivatives: with 1/dy = 0 every y difference quotient of the wrap-indexed kernels (whose y neighbour
    // is the cell itself) is an exact zero, also where hipcc contracts c0 - C*rt into an FMA that would return c0's rounding error
    g.rdy = flat_y ? 0.0 : 1.0 / dy;
    auto dcol = [&](int c) { return ctx->d_columns + (size_t)c * nf + Hz; };
    g.dzc = dcol(C_DZC); g.dzf = dcol(C_DZF); g.rdzf = dcol(C_RDZF); g.rdzc = dcol(C_RDZC); g.zc = dcol(C_ZC);
    g.formulation = 0;
    g.microphysics = 0; g.sa_maxiter = 0; g.qv_field = nullptr; g.ql_field = nullptr;
    g.sa_Ll = g.sa_cl = g.sa_dc = g.sa_L0 = g.sa_Ttr = g.sa_ptr = g.sa_abstol = 0.0;
    g.Ax = dcol(C_AX); g.Ay = dcol(C_AY);
    g.Vinv_c = dcol(C_VIC); g.Vinv_f = dcol(C_VIF);
    g.rho = dcol(C_RHO); g.rho_f = dcol(C_RHOF);
    g.rrho = dcol(C_RRHO); g.rrho_f = dcol(C_RRHOF);
    g.p_r = dcol(C_PR); g.T_r = dcol(C_TR);
    g.lnpi = dcol(C_LNPI);
    g.pi_dry = ColPtr(nullptr);      // set by bzi_lean_setup (anelastic contexts)
    g.g = constants->gravitational_acceleration;
    g.Rd = constants->dry_air_gas_constant;
    g.Rv = constants->vapor_gas_constant;
    g.cpd = constants->dry_air_heat_capacity;
    g.cpv = constants->vapor_heat_capacity;
    g.kap_num = g.Rv * g.cpd - g.Rd * g.cpv;
    g.pst = ref->standard_pressure;
    g.wrap_y = (slab_mode || bounded_y) ? 0 : 1;
    g.flat_y = flat_y ? 1 : 0;
    g.bounded_y = bounded_y ? 1 : 0;
    g.bounded_x = bounded_x ? 1 : 0;

    ctx->fused_ok = (Nx >= 2 * grid->Hx) && (Ny >= 2 * grid->Hy || slab_mode) && !ctx->tune.no_fused;
#ifndef BZ_CENTERED2
    ctx->weno_R = weno_R;
    ctx->scalar_R = weno_R;
    // the lean and fused-RK tiers of the anelastic model are order-5 kernels (bz_step.hip tests weno_R); orders 7, 9 take the
    // fused-streaming tier: generic tendency kernels + RK update, projection / diagnosis / halo fills in fused passes (the compressible
    // sequence takes its slow tendencies from the generic kernels and is otherwise independent of the advection order)
#endif
    if (slab_mode && (Ny < grid->Hy || Nx < 2 * grid->Hx)) { delete ctx; return BZ_ERR_UNSUPPORTED; }
    if (ctx->tune.tend_gen) ctx->tend_gen = ctx->tune.tend_gen;
    ctx->fuse_rk = !ctx->tune.no_fuse_rk;
    ctx->tend_lds = !ctx->tune.no_tend_lds;
    // Flat y: the anelastic model steps with one kernel per reference kernel (bz_tendency.hip); the compressible kernels reach their
    // y neighbours through wrap offsets, which are zero when Ny = 1 (bz_compressible.hip: wrap_of), and keep their fused sequence
    if (flat_y) { if (!compressible) ctx->fused_ok = false; ctx->tend_gen = 1; ctx->tend_lds = false; }
    if (bounded_y && !compressible) {      // per-operator entry points: one kernel per reference kernel, row-wise buffers (bz_tendency.hip); whole steps of the dry
        ctx->walls_lean_ok = ctx->fused_ok;      // model: the lean seam with its WY kernels (bz_step.hip)
        ctx->fused_ok = false; ctx->tend_gen = 1; ctx->tend_lds = false;
    }
    ctx->compressible = compressible;
    ctx->dz_min = dzc[Hz];
    for (int k = 0; k < Nz; ++k) ctx->dz_min = std::fmin(ctx->dz_min, dzc[Hz + k]);
    int rc = compressible ? BZ_OK : bzi_poisson_setup(ctx, ref->density);
    if (rc == BZ_OK && !compressible) rc = bzi_lean_setup(ctx);
    if (rc != BZ_OK) {
        fprintf(stderr, "bz_create: Poisson setup failed (%d): %s\n", rc, ctx->last_error.c_str());
        bz_destroy(ctx);
        return rc;
    }
    bzi_graph_configure(ctx);
    *out = ctx;
    return BZ_OK;
}

extern "C" void bz_destroy(bz_ctx *ctx)
{
    if (!ctx) return;
    profile_drain(ctx);
    bzi_graph_destroy(ctx);
    bzi_comm_teardown(ctx);
    bzi_poisson_teardown(ctx);
    bzi_lean_teardown(ctx);
    bzi_compressible_teardown(ctx);
    if (ctx->d_columns) hipFree(ctx->d_columns);
    bzi_forcing_teardown(ctx);
    bzi_closure_teardown(ctx);
    if (ctx->d_scalar) hipFree(ctx->d_scalar);
    if (ctx->d_gflux) hipFree(ctx->d_gflux);
    delete ctx;
}
