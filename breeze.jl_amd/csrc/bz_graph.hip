// hipGraph replay of whole time steps (opt-in: bz_graph_enable / BZ_GRAPH=1).
//
// On small grids a step is a chain of 40 (anelastic lean seam) to several hundred (compressible: 4 kernels per acoustic substep,
// per-operator 2-D anelastic: ~60) kernels of a few microseconds each.  A step is a pure function of (argument structs, dt, context
// configuration): no host decision inside it depends on device data, nothing is allocated, nothing synchronises.  So the second time
// a step is requested with the same key it is recorded with stream capture into a hipGraph, and from then on one hipGraphLaunch
// replaces the launches.  A different dt (TimeStepWizard) or different arrays is a different key: two graphs are kept, least
// recently recorded replaced.  Profiling (HIP events around kernel groups) and library-owned communicators (side streams with
// pending halo state across steps) switch replay off.  Replayed steps are bit-identical to launched ones (tests/test_graph_replay.py).
//
// Measured on MI355X / ROCm 7.2 (tools/small_grid_latency.py, BZ_GRAPH=0 vs 1): 256 x 256 2-D bubble (BASELINE configs[0]) 0.444 ->
// 0.420 ms/step, 32^3 0.221 -> 0.216, 64^3 and larger unchanged.  The host was already enqueueing ahead of the device; what bounds
// these steps is the dependency latency between consecutive small kernels (~4 us each), which a ROCm graph launch — one AQL packet
// per node, barrier bit set — does not remove.  Recording costs about a millisecond, so with a dt that changes every few steps
// replay loses; hence opt-in, not default.  Fewer, larger kernels (what the lean seam does for the 3-D model) is what moves this floor.
#include "bz_internal.h"

#include <cstdio>
#include <cstdlib>
#include <cstring>

// replay switches itself off for the context the first time a recording fails; BZ_GRAPH_DEBUG=1 says why
static void give_up(bz_ctx *ctx, const char *where, hipError_t e, int body_rc)
{
    ctx->graph_mode = 0;
    if (ctx->tune.graph_debug)
        fprintf(stderr, "[bz_graph] replay disabled at %s: hip error %d (%s), body rc %d (%s)\n", where, (int)e, hipGetErrorString(e), body_rc,
                ctx->last_error.c_str());
    (void)hipGetLastError();
}

static uint64_t fnv(uint64_t h, const void *p, size_t n)
{
    const unsigned char *b = (const unsigned char *)p;
    for (size_t i = 0; i < n; ++i) { h ^= b[i]; h *= 1099511628211ULL; }
    return h;
}

uint64_t bzi_graph_key(const bz_ctx *ctx, int kind, double dt, const void *a, size_t na, const void *b, size_t nb, const void *c, size_t nc,
                       const void *d, size_t nd)
{
    uint64_t h = 1469598103934665603ULL;
    h = fnv(h, &kind, sizeof(kind));
    h = fnv(h, &dt, sizeof(dt));
    h = fnv(h, &ctx->config_epoch, sizeof(ctx->config_epoch));
    h = fnv(h, &ctx->stream, sizeof(ctx->stream));
    if (a) h = fnv(h, a, na);
    if (b) h = fnv(h, b, nb);
    if (c) h = fnv(h, c, nc);
    if (d) h = fnv(h, d, nd);
    return h ? h : 1;
}

static void slot_clear(bz_ctx::GraphSlot &s)
{
    if (s.exec) hipGraphExecDestroy(s.exec);
    if (s.graph) hipGraphDestroy(s.graph);
    s = bz_ctx::GraphSlot();
}

void bzi_graph_destroy(bz_ctx *ctx)
{
    for (auto &s : ctx->graph_slots) slot_clear(s);
    if (ctx->graph_stream) { hipStreamDestroy(ctx->graph_stream); ctx->graph_stream = nullptr; }
}

int bzi_graph_begin(bz_ctx *ctx, uint64_t key, bool *capture)
{
    *capture = false;
    if (ctx->graph_mode == 0 || ctx->profiling || ctx->comm || ctx->graph_capturing) return 0;
    bz_ctx::GraphSlot *slot = nullptr;
    for (auto &s : ctx->graph_slots)
        if (s.key == key) slot = &s;
    if (slot && slot->exec) {
        if (hipGraphLaunch(slot->exec, ctx->stream) != hipSuccess) {      // a stale graph is dropped, the step runs the ordinary way
            (void)hipGetLastError();
            slot_clear(*slot);
            return 0;
        }
        ctx->G_is_predictor = slot->g_is_predictor;
        ctx->lean_step_last = slot->lean_step_last;
        ctx->lsum_step_last = slot->lsum_step_last;
        ++ctx->graph_replays;
        return 1;
    }
    if (!slot) {      // first sighting: run it the ordinary way (this also warms every lazily initialised library path up)
        slot = &ctx->graph_slots[ctx->graph_next];
        ctx->graph_next ^= 1;
        slot_clear(*slot);
        slot->key = key;
        slot->seen = 1;
        return 0;
    }
    // second sighting: record
    // the step is recorded on a stream of the library's own (PyTorch's current stream is normally the legacy default stream, which
    // cannot be captured) and replayed on the caller's stream
    if (!ctx->graph_stream) {
        if (const hipError_t e = hipStreamCreate(&ctx->graph_stream)) { give_up(ctx, "hipStreamCreate", e, 0); return 0; }
    }
    ctx->graph_user_stream = ctx->stream;
    if (bzi_apply_stream(ctx, ctx->graph_stream) != BZ_OK) {
        bzi_apply_stream(ctx, ctx->graph_user_stream);
        give_up(ctx, "hipfftSetStream", hipSuccess, 0);
        return 0;
    }
    if (const hipError_t e = hipStreamBeginCapture(ctx->stream, hipStreamCaptureModeThreadLocal)) {
        bzi_apply_stream(ctx, ctx->graph_user_stream);
        give_up(ctx, "hipStreamBeginCapture", e, 0);
        return 0;
    }
    ctx->graph_capturing = true;
    *capture = true;
    return 0;
}

int bzi_graph_end(bz_ctx *ctx, uint64_t key, int body_rc)
{
    ctx->graph_capturing = false;
    hipGraph_t graph = nullptr;
    const hipError_t e = hipStreamEndCapture(ctx->stream, &graph);
    bzi_apply_stream(ctx, ctx->graph_user_stream);
    bz_ctx::GraphSlot *slot = nullptr;
    for (auto &s : ctx->graph_slots)
        if (s.key == key) slot = &s;
    if (body_rc != BZ_OK || e != hipSuccess || !graph || !slot) {
        // nothing of the recorded step has executed; the caller runs it the ordinary way and replay stays off for this context
        give_up(ctx, "hipStreamEndCapture", e, body_rc);
        if (graph) hipGraphDestroy(graph);
        return body_rc != BZ_OK ? body_rc : -1;
    }
    hipGraphExec_t exec = nullptr;
    const hipError_t ei = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
    if (ei != hipSuccess || !exec) {
        give_up(ctx, "hipGraphInstantiate", ei, 0);
        hipGraphDestroy(graph);
        return -1;
    }
    slot->graph = graph;
    slot->exec = exec;
    slot->g_is_predictor = ctx->G_is_predictor;
    slot->lean_step_last = ctx->lean_step_last;
    slot->lsum_step_last = ctx->lsum_step_last;
    ++ctx->graph_captures;
    if (const hipError_t el = hipGraphLaunch(exec, ctx->stream)) {
        give_up(ctx, "hipGraphLaunch", el, 0);
        slot_clear(*slot);
        return -1;
    }
    return BZ_OK;
}

// off unless BZ_GRAPH=1 or bz_graph_enable(ctx, 1)
void bzi_graph_configure(bz_ctx *ctx)
{
    ctx->graph_mode = 0;
    if (ctx->tune.graph >= 0) ctx->graph_mode = ctx->tune.graph != 0 ? 1 : 0;
}

extern "C" int bz_graph_info(bz_ctx *ctx, int32_t *enabled, int64_t *captures, int64_t *replays)
{
    if (!ctx) return BZ_ERR_INVALID;
    if (enabled) *enabled = ctx->graph_mode;
    if (captures) *captures = ctx->graph_captures;
    if (replays) *replays = ctx->graph_replays;
    return BZ_OK;
}

extern "C" int bz_graph_enable(bz_ctx *ctx, int on)
{
    if (!ctx) return BZ_ERR_INVALID;
    ctx->graph_mode = on ? 1 : 0;
    if (!on) bzi_graph_destroy(ctx);
    return BZ_OK;
}
