// bz_step.hip — host-side orchestration of the device-resident anelastic SSP-RK3 step.
//   update_state!  /root/reference/src/AtmosphereModels/update_atmosphere_model_state.jl:41-68
//   time_step!     /root/reference/src/TimeSteppers/ssp_runge_kutta_3.jl:209-278
// Everything is enqueued on ctx->stream; there is no host synchronisation inside a step.
#include <cstdlib>

#include "bz_internal.h"

extern "C" int bz_update_state(bz_ctx *ctx, const bz_state *s, const bz_prognostic *G, int compute_tendencies)
{
    if (!ctx || !s || (compute_tendencies && !G)) return BZ_ERR_INVALID;
    int rc;
    if (ctx->comm && (rc = bzi_comm_join_pending(ctx))) return rc;      // an undiagnosed last stage leaves its halo exchange on the side stream
    ctx->diagnostics_stale = false;
    if (ctx->d_qstate) BZ_HIP(hipMemsetAsync(ctx->d_qstate, 0, sizeof(int), ctx->stream));      // the moisture may have been set!: unknown (= moist) until the next scan
    bzi_moisture_unknown(ctx);
    // fill_halo_regions!(prognostic_fields(model))  (:48) — momentum halos are filled inside
    // bz_compute_velocities (:135-136), the scalars here.
    double *sf[4] = {s->rho_theta, s->rho_q, ctx->dg.rqcl_field, ctx->dg.rqr_field};
    int sk[4] = {0, 0, 0, 0};
    if ((rc = bzi_fill_halos_multi(ctx, sf, sk, ctx->dg.microphysics == 2 ? 4 : 2))) return rc;
    // compute_auxiliary_variables!  (:207-223)
    if ((rc = bz_compute_velocities(ctx, s))) return rc;
    if ((rc = bz_compute_auxiliary_thermodynamic_variables(ctx, s))) return rc;
    // compute_tendencies!  (:63)
    if (compute_tendencies && (rc = bz_compute_tendencies(ctx, s, G))) return rc;
    return BZ_OK;
}

static int anelastic_step_body(bz_ctx *ctx, const bz_state *s, const bz_prognostic *U0, const bz_prognostic *G, double dt, bool diagnose);

// host-side bookkeeping at the end of a lean step (single-device and slab drivers): where the ping-pong pair sits now
void bzi_lean_step_done(bz_ctx *ctx, bool diagnosed)
{
    ctx->lean_step_last = true;
    ctx->G_is_predictor = true;
    ctx->diagnostics_stale = !diagnosed;
}

// Buffer rotation of the lean seam (round 4).  Three sets of prognostic arrays take part: A = the state `s`, B = the G slots,
// C = the U0 slots.  Nothing is ever copied into U0 (store_initial_state!, ssp_runge_kutta_3.jl:180-186, has no pass and no store
// of its own): the state arrays stay intact as "U0" until the last writer of the step.
//   momentum   tendency + RK kernels: stencils from A (stage 1) / C (stages 2, 3), u0 from A, predictor -> B;
//              projection: B -> C (stages 1, 2), B -> A (stage 3, when nothing reads A any more)
//   scalars    stage 1: A -> B;  stage 2: stencils from B, u0 from A -> C;  stage 3: stencils from C, u0 from A -> A in place
//              (a thread reads A only at its own cell, before it writes it)
// so every step ends with the whole prognostic state in `s`, diagnosed or not.  The U0 arrays are the time stepper's scratch
// (as in the reference, where nothing reads U0 outside time_step!).
void bzi_lean_stage(const bz_state *s, const bz_prognostic *U0, const bz_prognostic *G, int stage, LeanStage *L)
{
    L->sin = *s; L->sout = *s;
    if (stage > 0) { L->sin.rho_u = U0->rho_u; L->sin.rho_v = U0->rho_v; L->sin.rho_w = U0->rho_w; }
    if (stage < 2) { L->sout.rho_u = U0->rho_u; L->sout.rho_v = U0->rho_v; L->sout.rho_w = U0->rho_w; }
    L->u0.rho_u = s->rho_u; L->u0.rho_v = s->rho_v; L->u0.rho_w = s->rho_w; L->u0.rho_theta = s->rho_theta; L->u0.rho_q = s->rho_q;
    L->pa = stage == 0 ? s->rho_theta : stage == 1 ? G->rho_theta : U0->rho_theta;
    L->pb = stage == 0 ? s->rho_q : stage == 1 ? G->rho_q : U0->rho_q;
    L->oa = stage == 0 ? G->rho_theta : stage == 1 ? U0->rho_theta : s->rho_theta;
    L->ob = stage == 0 ? G->rho_q : stage == 1 ? U0->rho_q : s->rho_q;
}

// ---- moisture scan -------------------------------------------------------------------------------------------------------------------
// The reference advects rho q^v in every model, dry ones included (update_atmosphere_model_state.jl:333-343).  In a dry run the field
// is identically zero and stays so: every flux of it is an exact zero, its update is 0 -> 0.  Every step call opens with one pass over
// rho q (1 word per cell, once per CALL — bz_time_steps_anelastic amortises it over its n steps) that leaves *d_qstate = 1 if every
// element is zero (2 otherwise); the lean scalar-pair and z-momentum kernels read that word and, where it is 1, skip every load, store and
// flux of rho q (4.6 of the scalar kernel's 13 words per cell, a fifth of its instructions).  The result carries the same bits either
// way.  2 is sticky (a moist model is scanned once, later calls return at once) until bz_update_state — what set! ends with — resets it;
// a dry model is scanned at every call, so a moisture field written behind the library's back is seen.  Contexts with a moisture
// source (bottom moisture flux, moisture forcing, subsidence of q) or with microphysics never take the shortcut.
// state[0]: 0 unknown (treated as moist: nothing is skipped), 1 identically zero (verified by the scan that opened this call),
//           2 moist (sticky until bz_update_state puts 0 back);  state[1]: "some block found a non-zero element" of the scan in flight
__global__ __launch_bounds__(256) void k_scan_moisture(const double *__restrict__ rq, long long n, int *__restrict__ state)
{
    if (state[0] == 2) return;
    bool found = false;
    for (long long t = (long long)blockIdx.x * 256 + threadIdx.x; t < n; t += (long long)gridDim.x * 256) found = found || (rq[t] != 0.0);
    if (__any(found) && (threadIdx.x & 63) == 0) state[1] = 1;
}
__global__ void k_scan_moisture_close(int *__restrict__ state)
{
    if (state[0] != 2) state[0] = state[1] ? 2 : 1;
    state[1] = 0;
}
// y-slab contexts: the verdict must be the DOMAIN's, not the slab's (a rank whose slab is dry while its neighbour's is not would drop
// the moisture that arrives through its halo rows for the rest of the call): every rank contributes "found" (0 / 1, also a
// rank that already knows it is moist) to a sum over the communicator and closes the scan on the sum
__global__ void k_scan_moisture_flag(const int *__restrict__ state, double *__restrict__ flag) { flag[0] = (state[0] == 2 || state[1]) ? 1.0 : 0.0; }
__global__ void k_scan_moisture_close_global(int *__restrict__ state, const double *__restrict__ flag)
{
    state[0] = flag[0] != 0.0 ? 2 : 1;
    state[1] = 0;
}

static bool moisture_sources(const bz_ctx *ctx)
{
    return ctx->tune.no_dry_shortcut || ctx->dg.microphysics != 0 || ctx->has_bulk || ctx->has_relaxation || ctx->field_forcing != nullptr ||
           (ctx->has_forcings && (ctx->forcing_flux_q != 0.0 || (ctx->forcing_static_mask & 8) || (ctx->forcing_subsidence_mask & 8))) ||
           (ctx->slab_mode && !ctx->comm);      // slabs driven from the host: the library cannot ask the other ranks, so nothing is skipped
}

// the word the lean kernels read (nullptr: no shortcut on this context)
const int *bzi_moisture_state(const bz_ctx *ctx) { return (ctx->d_qstate && !moisture_sources(ctx)) ? ctx->d_qstate : nullptr; }

int bzi_scan_moisture(bz_ctx *ctx, const bz_state *s) { return ctx->compressible ? BZ_OK : bzi_scan_moisture_field(ctx, s->rho_q); }

void bzi_moisture_unknown(bz_ctx *ctx) { ctx->q_host = 0; ctx->q_pending = false; }

int bzi_scan_moisture_field(bz_ctx *ctx, const double *rho_q)
{
    if (moisture_sources(ctx)) return BZ_OK;
    const DevGrid &g = ctx->dg;
    int rc;
    if (!ctx->d_qstate) {
        BZ_HIP(hipMalloc(&ctx->d_qstate, 2 * sizeof(int) + 2 * sizeof(double)));      // [state, found] + the flag of the slab all-reduce
        BZ_HIP(hipMemsetAsync(ctx->d_qstate, 0, 2 * sizeof(int) + 2 * sizeof(double), ctx->stream));
        BZ_HIP(hipHostMalloc(&ctx->h_qstate, sizeof(int)));
        BZ_HIP(hipEventCreateWithFlags(&ctx->ev_q, hipEventDisableTiming));
        ctx->q_host = 0;
    }
    // the verdict of the previous call's scan, if it has landed (asynchronous copy, never waited for)
    if (ctx->q_pending && !ctx->comm && hipEventQuery(ctx->ev_q) == hipSuccess) { ctx->q_host = *ctx->h_qstate; ctx->q_pending = false; }
    if (ctx->q_host == 2) return BZ_OK;      // moist is sticky on the device (until bz_update_state) and on every rank of a slab communicator alike: nothing to scan
    // an undiagnosed last stage leaves the halo exchange of rho q on the side stream: the scan reads those rows
    if (ctx->comm && (rc = bzi_comm_join_pending(ctx))) return rc;
    const long long n = (long long)g.Sxy * (g.Nz + 2 * g.Hz);
    {
    ProfileScope ps(ctx, "moisture_scan");
    hipLaunchKernelGGL(k_scan_moisture, dim3(ctx->num_cus * 8), dim3(256), 0, ctx->stream, rho_q, n, ctx->d_qstate);
    if (ctx->comm) {
        double *flag = (double *)(ctx->d_qstate + 2);
        hipLaunchKernelGGL(k_scan_moisture_flag, dim3(1), dim3(1), 0, ctx->stream, ctx->d_qstate, flag);
        BZ_LAUNCH_CHECK();
        if ((rc = bzi_comm_allreduce_sum(ctx, flag, 1))) return rc;      // a collective: every rank of the communicator scans at every call
        hipLaunchKernelGGL(k_scan_moisture_close_global, dim3(1), dim3(1), 0, ctx->stream, ctx->d_qstate, flag);
    } else
        hipLaunchKernelGGL(k_scan_moisture_close, dim3(1), dim3(1), 0, ctx->stream, ctx->d_qstate);
    BZ_LAUNCH_CHECK();
    }
    if (ctx->graph_capturing) return BZ_OK;      // (never: the scan precedes the recorded region)
    BZ_HIP(hipMemcpyAsync(ctx->h_qstate, ctx->d_qstate, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
    if (ctx->q_host == 0 || ctx->comm) {
        // first scan since update_state!: read the verdict back before the first tendency launch — one host wait per set! / first step —
        // so that from here on the host launches the body that applies (separate kernels, separate rocprofv3 rows).
        // Slab communicators read it back at EVERY call: the asynchronous path below lands at a rank-dependent time, and a rank that has
        // learnt "moist" stops scanning — its peers would wait for it in the scan's all-reduce.  Read synchronously, the (all-reduced)
        // verdict is the same on every rank at the same call
        BZ_HIP(hipStreamSynchronize(ctx->stream));
        ctx->q_host = *ctx->h_qstate;
        ctx->q_pending = false;
    } else {
        BZ_HIP(hipEventRecord(ctx->ev_q, ctx->stream));
        ctx->q_pending = true;
    }
    return BZ_OK;
}

// the lean whole-step seam applies: dry / vapour WENO5 theta model without closure, bulk fluxes, relaxation, tracers, microphysics
// (walls in y ((Periodic, Bounded, Bounded)) ride it too: WY kernels, wall rows in the projection kernels)
static bool anelastic_lean_tier(const bz_ctx *ctx)
{
    const bool walls_lean = ctx->dg.bounded_y && ctx->walls_lean_ok;
    return (ctx->fused_ok || walls_lean) && ctx->fuse_rk && ctx->lean && ctx->weno_R == 3 && ctx->scalar_R == 3 && !ctx->compressible && ctx->dg.formulation == 0 && ctx->dg.microphysics == 0 &&
        (!ctx->has_forcings || bzi_lean_forcings_ok(ctx)) && !ctx->has_bulk && !ctx->has_closure && !ctx->has_relaxation && ctx->n_tracers == 0 && !ctx->bounded_mask &&
        (long long)ctx->dg.Sxy * (ctx->dg.Nz + 2 * ctx->dg.Hz + 1) < (1LL << 32);
}

static int one_anelastic_step(bz_ctx *ctx, const bz_state *s, const bz_prognostic *U0, const bz_prognostic *G, double dt, bool diagnose)
{
    if (ctx->slab_mode) {
        if (ctx->comm) return bzi_dist_time_step(ctx, s, U0, G, dt, diagnose);      // the library owns the exchanges (bz_comm.hip)
        ctx->last_error = "bz_time_step_anelastic: a y-slab context needs a communicator (bz_comm_init_rccl / bz_comm_init_local) "
                          "or a host-side distributed driver";
        return BZ_ERR_UNSUPPORTED;
    }
    // the tiers other than the lean seam start from the stored diagnostics: if undiagnosed lean steps came before (the configuration
    // changed in between), rebuild them — here, outside the region a graph records (a rebuild baked into a recorded step would run, and
    // reset the moisture scan's word, on every replay)
    if (ctx->diagnostics_stale && !anelastic_lean_tier(ctx)) { const int rcs = bz_update_state(ctx, s, G, 0); if (rcs) return rcs; }
    // launch-bound grids replay the recorded step (bz_graph.hip); a failed recording has executed nothing and falls through
    // (the host's view of the moisture scan selects the kernels a recording holds; so does the source of the first stage's level sums)
    const int kind = 1 + (diagnose ? 0 : 16) + 32 * ctx->q_host + (ctx->lsum_fresh ? 128 : 0);
    const uint64_t key = bzi_graph_key(ctx, kind, dt, s, sizeof(*s), U0, sizeof(*U0), G, sizeof(*G), nullptr, 0);
    bool capture = false;
    int rc;
    if (bzi_graph_begin(ctx, key, &capture) == 1) {
        if (ctx->lean_step_last) ctx->diagnostics_stale = !diagnose;      // the host-side bookkeeping of the recorded body
        ctx->lsum_fresh = ctx->lsum_step_last;
        return BZ_OK;
    }
    const bool lsum_in = ctx->lsum_fresh;
    if (capture) {
        rc = anelastic_step_body(ctx, s, U0, G, dt, diagnose);
        if ((rc = bzi_graph_end(ctx, key, rc)) != -1) return rc;
        ctx->lsum_fresh = lsum_in;      // nothing ran
    }
    return anelastic_step_body(ctx, s, U0, G, dt, diagnose);
}

// The level sums of SubsidenceForcing ride on the projection + diagnosis kernel (bz_fused.hip) from one stage to the next INSIDE a call;
// the first stage of a call takes them from its own pass (the host may have written the fields), and nothing is trusted after it.
static void level_sums_open(bz_ctx *ctx)
{
    ctx->lsum_fresh = false;
    long long P;
    (void)bzi_level_sum_rows(ctx, &P);      // allocated here: not inside a region a graph records
}

extern "C" int bz_time_step_anelastic(bz_ctx *ctx, const bz_state *s, const bz_prognostic *U0,
                                      const bz_prognostic *G, double dt)
{
    if (!ctx || !s || !U0 || !G) return BZ_ERR_INVALID;
    level_sums_open(ctx);
    int rc = bzi_scan_moisture(ctx, s);
    if (!rc) rc = one_anelastic_step(ctx, s, U0, G, dt, true);
    ctx->lsum_fresh = false;
    return rc;
}

// n steps of time_step!(model, dt) in one call — the loop of the reference's benchmark driver, many_time_steps!
// (/root/reference/benchmarking/src/timestepping.jl:11-16) and of run!(simulation) between two callback / output iterations: nothing
// reads model.velocities, theta, q^v, T or the pressure anomaly in between, and the lean tendency kernels derive what they need from the
// prognostic fields.  So every step but (optionally) the last ends its third stage with the momentum-only projection instead of the
// projection + diagnosis pass (13 words per cell written for host consumers only).  diagnose_last != 0: on return every field and halo
// of `s` carries the bits n calls of bz_time_step_anelastic leave.  diagnose_last == 0: the prognostic fields of `s` are current, the
// diagnostics are stale until bz_update_state(ctx, s, G, 0) — which bz_time_step(s)_anelastic do NOT need: stepping can simply go
// on.  Tiers other than the lean seam diagnose every step.
extern "C" int bz_time_steps_anelastic(bz_ctx *ctx, const bz_state *s, const bz_prognostic *U0, const bz_prognostic *G, double dt, int n,
                                       int diagnose_last)
{
    if (!ctx || !s || !U0 || !G || n < 0) return BZ_ERR_INVALID;
    level_sums_open(ctx);
    if (n > 0) { const int rc = bzi_scan_moisture(ctx, s); if (rc) return rc; }
    for (int it = 0; it < n; ++it) {
        const int rc = one_anelastic_step(ctx, s, U0, G, dt, it == n - 1 && diagnose_last);
        if (rc) { ctx->lsum_fresh = false; return rc; }
    }
    ctx->lsum_fresh = false;
    return BZ_OK;
}

int bzi_refresh_diagnostics(bz_ctx *ctx, const bz_state *s, const char *who)
{
    if (!ctx->diagnostics_stale) return BZ_OK;
    if (!s) {
        ctx->last_error = std::string(who) + ": the diagnostics are stale (bz_time_steps_anelastic ended without the last diagnosis); call bz_update_state first";
        return BZ_ERR_INVALID;
    }
    if (ctx->comm) return bz_comm_update_state_and_project(ctx, s, nullptr, 1.0, 0);
    return bz_update_state(ctx, s, nullptr, 0);
}

extern "C" int bz_diagnostics_stale(const bz_ctx *ctx) { return ctx ? (ctx->diagnostics_stale ? 1 : 0) : BZ_ERR_INVALID; }

static int anelastic_step_body(bz_ctx *ctx, const bz_state *s, const bz_prognostic *U0, const bz_prognostic *G, double dt, bool diagnose)
{
    ctx->lean_step_last = false;
    ctx->lsum_step_last = false;
    const bool lsum_in = ctx->lsum_fresh;      // every other tier leaves no sums behind
    ctx->lsum_fresh = false;
    int rc;
    const double alphas[3] = {1.0, 1.0 / 4.0, 2.0 / 3.0};                   // :99-101
    // walls in y ((Periodic, Bounded, Bounded)) ride the lean seam too (WY kernels, wall rows in the projection kernels); every other
    // configuration with walls steps operator by operator
    const bool walls_lean = ctx->dg.bounded_y && ctx->walls_lean_ok;
    const bool lean_tier = anelastic_lean_tier(ctx);
    // (the other tiers start from the stored diagnostics: one_anelastic_step rebuilt them if undiagnosed lean steps came before)
    if (lean_tier) {
        // Lean seam (bz_tendency5_kernels.h): the tendency kernels read the prognostic fields only and derive u, v, w, theta,
        // q^v, T on the fly (bit-identical to the stored diagnostics), rho theta / rho q ping-pong between their own arrays
        // and the G slots, the projection of stages 1-2 writes momentum only; stage 3 runs the full projection + diagnosis
        // kernel, so on return every field of `s` (halos included) is what the per-operator sequence leaves.
        const DevGrid &g = ctx->dg;
        BZ_HIP(hipMemsetAsync(G->rho_w + g.Sxy * g.Hz, 0, g.Sxy * sizeof(double), ctx->stream));                 // wall faces of the
        BZ_HIP(hipMemsetAsync(G->rho_w + g.Sxy * (g.Hz + g.Nz), 0, g.Sxy * sizeof(double), ctx->stream));        // predictor stay 0,
        BZ_HIP(hipMemsetAsync(U0->rho_w + g.Sxy * g.Hz, 0, g.Sxy * sizeof(double), ctx->stream));                // and so do those of the
        BZ_HIP(hipMemsetAsync(U0->rho_w + g.Sxy * (g.Hz + g.Nz), 0, g.Sxy * sizeof(double), ctx->stream));       // projected momentum of stages 1-2
        for (int stage = 0; stage < 3; ++stage) {
            const double alpha = alphas[stage];
            const bool full = diagnose && stage == 2;      // projection + diagnosis (else: momentum-only projection)
            LeanStage LS;
            bzi_lean_stage(s, U0, G, stage, &LS);
            const bz_state *sin = &LS.sin, *sout = &LS.sout;
            const bz_prognostic *u0 = &LS.u0;
            const double *pa = LS.pa, *pb = LS.pb;
            double *oa = LS.oa, *ob = LS.ob;
            if (ctx->side_scalar && !ctx->has_forcings) {
                // the scalar-pair kernel feeds nothing of the pressure solve: it runs on the side stream beside the source term,
                // the transforms and the Thomas solve (issue-bound stencil kernel next to bandwidth-bound streaming kernels)
                if ((rc = bzi_tendencies_lean(ctx, sin, u0, G, pa, pb, oa, ob, dt, alpha, stage == 0, 0, 1))) return rc;
                BZ_HIP(hipEventRecord(ctx->ev_fork, ctx->stream));
                BZ_HIP(hipStreamWaitEvent(ctx->side_stream, ctx->ev_fork, 0));
                hipStream_t keep = ctx->stream;
                ctx->stream = ctx->side_stream;
                rc = bzi_tendencies_lean(ctx, sin, u0, G, pa, pb, oa, ob, dt, alpha, stage == 0, 0, 2);
                ctx->stream = keep;
                if (rc) return rc;
                BZ_HIP(hipEventRecord(ctx->ev_join, ctx->side_stream));
            } else if ((rc = bzi_tendencies_lean(ctx, sin, u0, G, pa, pb, oa, ob, dt, alpha, stage == 0))) return rc;
            if (ctx->has_forcings) {
                // the bottom fluxes of the stage (bzi_lean_forcings_ok), evaluated from the intact previous-stage momentum and added,
                // weighted alpha dt, to what the fused RK updates just wrote: the predictor momentum in the G slots and rho theta /
                // rho q in the stage's output buffers — as the fused-RK tier below does
                // (the Coriolis / profile terms went into the RK epilogues of the momentum kernels: Lean5::mforce, bz_tendency5.hip)
                if ((rc = bzi_flux_bc(ctx, sin, G->rho_u, G->rho_v, oa, ob, alpha * dt))) return rc;
                // (T is no stored field inside the lean seam any more: the z-momentum kernel of the next stage derives it from oa, ob)
            }
            if (ctx->pchunk) {
                // chunked pipeline: each level range goes source term -> x transform -> y transform (and, after the vertical solves,
                // y -> x -> projection) back to back, so the intermediate passes find the range in the Infinity Cache
                const int ch = ctx->pchunk;
                rc = 0;
                {
                    ProfileScope ps(ctx, "poisson_source_term+fft_forward");
                    ctx->profile_mute++;
                    for (int k0 = 0; k0 < g.Nz && !rc; k0 += ch) {
                        ctx->kr0 = k0; ctx->krn = ch;
                        rc = bzi_poisson_source_fused(ctx, s, alpha * dt, nullptr, G);
                        if (!rc) rc = bzi_fft_chunk(ctx, k0, true);
                    }
                    ctx->krn = 0;
                    ctx->profile_mute--;
                }
                if (rc) return rc;
                {
                    ProfileScope ps(ctx, "poisson_tridiagonal");
                    if ((rc = bzi_tridiag_launch(ctx, (double *)ctx->d_hat, 1.0 / ((double)g.Nx * (double)g.Ny), g.Ny, 1))) return rc;
                }
                if (ctx->side_scalar && !ctx->has_forcings) BZ_HIP(hipStreamWaitEvent(ctx->stream, ctx->ev_join, 0));
                {
                    ProfileScope ps(ctx, !full ? "poisson_fft_inverse+project_momentum" : "poisson_fft_inverse+project_and_diagnose");
                    ctx->profile_mute++;
                    for (int k0 = 0; k0 < g.Nz && !rc; k0 += ch) {
                        rc = bzi_fft_chunk(ctx, k0, false);
                        ctx->kr0 = k0; ctx->krn = ch;
                        if (rc) break;
                        if (!full) rc = bzi_project_lean(ctx, sout, alpha * dt, nullptr, nullptr, G, oa, ob);
                        else rc = bzi_project_diagnose(ctx, s, alpha * dt, nullptr, nullptr, G, true, oa, ob);
                    }
                    ctx->krn = 0;
                    ctx->profile_mute--;
                }
                if (rc) return rc;
                continue;
            }
            if (ctx->xf && !ctx->pchunk) {
                // hand-written x transforms (bz_xfft_kernels.h): the source term is evaluated inside the forward x pass; y transforms
                // (contiguous) and vertical solves on the transposed spectrum
                {
                    ProfileScope ps(ctx, "poisson_source_term+fft_x");
                    if ((rc = bzi_xf_forward(ctx, s, alpha * dt, G))) return rc;
                }
                if ((rc = bzi_xf_middle(ctx))) return rc;
                {
                    ProfileScope ps(ctx, "poisson_fft_x_inverse");
                    if ((rc = bzi_xf_inverse(ctx))) return rc;
                }
                // the scalar-pair kernel of the side stream joins here: the projection kernels write the z-halo images of rho theta / rho q
                if (ctx->side_scalar && !ctx->has_forcings) BZ_HIP(hipStreamWaitEvent(ctx->stream, ctx->ev_join, 0));
                if (!full) rc = bzi_project_lean(ctx, sout, alpha * dt, nullptr, nullptr, G, oa, ob);
                else rc = bzi_project_diagnose(ctx, s, alpha * dt, nullptr, nullptr, G, true, oa, ob);
                if (rc) return rc;
                continue;
            }
            if ((rc = bzi_poisson_source_fused(ctx, s, alpha * dt, nullptr, G))) return rc;
            if ((rc = bzi_poisson_spectral(ctx))) return rc;
            if (ctx->side_scalar && !ctx->has_forcings) BZ_HIP(hipStreamWaitEvent(ctx->stream, ctx->ev_join, 0));
            if (!full) rc = bzi_project_lean(ctx, sout, alpha * dt, nullptr, nullptr, G, oa, ob);
            else rc = bzi_project_diagnose(ctx, s, alpha * dt, nullptr, nullptr, G, true, oa, ob);
            if (rc) return rc;
        }
        bzi_lean_step_done(ctx, diagnose);
        return BZ_OK;
    }
    // (walls in y: the generic order-7 / 9 kernels are wall-aware, the LDS-tiled order-5 kernels of this tier are not)
    if ((ctx->fused_ok || (walls_lean && ctx->weno_R != 3)) && ctx->fuse_rk && ctx->scalar_R == ctx->weno_R && (ctx->weno_R == 3 || ctx->n_tracers == 0) && ctx->dg.formulation == 0 &&
        ctx->dg.microphysics != 2 && !ctx->bounded_mask && !ctx->has_relaxation && !(ctx->has_forcings && ctx->tune.no_fuse_forcing)) {
        // Tendencies and the following RK update in one pass (bz_tendency.hip: bzi_tendencies_fused_rk): the
        // tendency of stage s is evaluated where the reference applies it (at the start of stage s, from the state
        // left by stage s-1), the predictor momentum goes to the G arrays and is projected from there into the
        // state arrays, rho_theta / rho_q advance in place.  G never makes the round trip through HBM.
        const DevGrid &g = ctx->dg;
        BZ_HIP(hipMemsetAsync(G->rho_w + g.Sxy * g.Hz, 0, g.Sxy * sizeof(double), ctx->stream));                 // wall faces of the
        BZ_HIP(hipMemsetAsync(G->rho_w + g.Sxy * (g.Hz + g.Nz), 0, g.Sxy * sizeof(double), ctx->stream));        // predictor stay 0
        for (int stage = 0; stage < 3; ++stage) {
            const double alpha = alphas[stage];
            // momentum terms of the forcing stack in the epilogues of the stored-velocity momentum kernels (round 5): the stage's subsidence
            // profiles are built first (from the stored u, v, theta, q of the stage start, which the tendency kernels do not touch)
            const bool fold = ctx->has_forcings && ctx->weno_R == 3 && bzi_k6_stored_ok(ctx) && ctx->tend_lds && !ctx->tune.no_fuse_forcing && !ctx->tune.no_fold_forcing;
            if (stage == 0) ctx->lsum_fresh = lsum_in;
            if (fold && (rc = bzi_compute_forcings(ctx, s))) return rc;
            ctx->fold_momentum_forcing = fold;
            rc = bzi_tendencies_fused_rk(ctx, s, U0, G, dt, alpha, stage == 0);
            ctx->fold_momentum_forcing = false;
            if (rc) return rc;
            if (ctx->n_tracers) {       // tracers ride beside the fused kernels: tendency from the previous-stage state, RK in place
                if ((rc = bzi_tracer_tendencies(ctx, s))) return rc;
                if ((rc = bzi_tracer_rk3(ctx, dt, alpha, stage == 0))) return rc;
            }
            if (ctx->has_closure &&
                (rc = bzi_apply_closure(ctx, s, G->rho_u, G->rho_v, G->rho_w, s->rho_theta, s->rho_q, alpha * dt))) return rc;
            if (ctx->has_forcings) {
                // forcing, Coriolis and bottom fluxes of the stage, evaluated from the still-intact previous-stage state and
                // added to what the fused RK update just wrote, weighted alpha dt
                if ((rc = bzi_apply_forcings(ctx, s, G->rho_u, G->rho_v, s->rho_theta, s->rho_q, alpha * dt, fold))) return rc;
            }
            if ((ctx->has_forcings || ctx->has_bulk) &&
                (rc = bzi_flux_bc(ctx, s, G->rho_u, G->rho_v, s->rho_theta, s->rho_q, alpha * dt))) return rc;
            if (g.bounded_y && (rc = bzi_fill_halo(ctx, G->rho_v, BZ_HALO_YFACE))) return rc;      // wall faces j = 0, Ny of the predictor (the source term reads face Ny)
            if ((rc = bzi_poisson_from_momentum(ctx, s, alpha * dt, G))) return rc;
            // pressure_anomaly is a diagnostic nobody reads inside the step: only the last stage scatters it
            // (the horizontal sums of the u, v, theta, q it stores ride along: the next stage's subsidence profiles, bz_forcing.hip)
            const bool ride = bzi_level_sums_ride(ctx) && ctx->d_lsum_rows;
            if ((rc = bzi_project_diagnose(ctx, s, alpha * dt, nullptr, nullptr, G, stage == 2, nullptr, nullptr, ride))) return rc;
            ctx->lsum_fresh = ride;
            if ((rc = bzi_tracer_specific(ctx))) return rc;
        }
        ctx->lsum_step_last = ctx->lsum_fresh;
        ctx->G_is_predictor = true;
        return BZ_OK;
    }
    if (ctx->fused_ok) {
        if (ctx->G_is_predictor && (rc = bz_compute_tendencies(ctx, s, G))) return rc;
        // Same arithmetic, fewer passes over HBM (bz_fused.hip): store_initial_state! rides on the first
        // RK update, the source term uses wrap indexing instead of a halo fill, and the projection,
        // velocity / thermodynamic diagnosis and every halo fill of update_state! are one kernel.
        for (int stage = 0; stage < 3; ++stage) {
            const double alpha = alphas[stage];
            if ((rc = bz_compute_flux_bc_tendencies(ctx, s, G))) return rc;          // :229,243,257
            if ((rc = bzi_rk3_fused(ctx, s, U0, G, dt, alpha, stage == 0))) return rc;
            if (ctx->dg.microphysics == 2 && (rc = bzi_kessler_rk3(ctx, dt, alpha, stage == 0))) return rc;
            if ((rc = bzi_tracer_rk3(ctx, dt, alpha, stage == 0))) return rc;
            if ((rc = bzi_poisson_from_momentum(ctx, s, alpha * dt, nullptr))) return rc;
            if ((rc = bzi_project_diagnose(ctx, s, alpha * dt))) return rc;
            if ((rc = bzi_tracer_specific(ctx))) return rc;
            if ((rc = bz_compute_tendencies(ctx, s, G))) return rc;
        }
        // microphysics_model_update!(model.microphysics, model) closes the step (ssp_runge_kutta_3.jl:262-263)
        if (ctx->dg.microphysics == 2 && (rc = bzi_kessler_update(ctx, s, G, dt))) return rc;
        return BZ_OK;
    }
    if (ctx->G_is_predictor && (rc = bz_compute_tendencies(ctx, s, G))) return rc;
    if ((rc = bz_store_initial_state(ctx, s, U0))) return rc;               // :223
    for (int stage = 0; stage < 3; ++stage) {
        const double alpha = alphas[stage];
        if ((rc = bz_compute_flux_bc_tendencies(ctx, s, G))) return rc;              // :229,243,257
        if ((rc = bz_ssp_rk3_substep(ctx, s, U0, G, dt, alpha))) return rc;          // :230,244,258
        if ((rc = bz_compute_pressure_correction(ctx, s, alpha * dt))) return rc;    // :232,246,260
        if ((rc = bz_make_pressure_correction(ctx, s, alpha * dt))) return rc;       // :233,247,261
        if ((rc = bz_update_state(ctx, s, G, 1))) return rc;                         // :236,250,270
    }
    if (ctx->dg.microphysics == 2 && (rc = bzi_kessler_update(ctx, s, G, dt))) return rc;
    return BZ_OK;
}
