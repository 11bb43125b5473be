// bz_comm.hip — communication of the y-slab decomposition BEHIND the C ABI (BASELINE.json north_star: "Julia host code calls ...
// through a thin C-ABI ... RCCL halo exchange and FFT all-to-all over xGMI").  One process per GPU; rank r of y_nranks owns
// rows [r Ny, (r+1) Ny) of a (Periodic, Periodic, Bounded) domain (bz_create_slab).  The reference has no distributed code of
// its own (it re-exports Oceananigans' MPI-based Distributed, /root/reference/src/Breeze.jl:172,183,209): this is this repo's
// design for 8 GPUs joined by point-to-point xGMI links.
//
//   transports   * RCCL (bz_comm_init_rccl): librccl is dlopen'ed (the copy PyTorch already loaded, or ROCm's), the communicator
//                  is created from a 128-byte unique id that the host distributes (rank 0: bz_comm_unique_id); every exchange is
//                  one ncclGroupStart / ncclSend / ncclRecv / ncclGroupEnd on a HIP stream — point-to-point messages, one per
//                  xGMI link, no ring collective in the data path.
//                * local (bz_comm_init_local): the ranks are contexts of ONE process (host threads), messages are peer copies
//                  ordered by events.  This is how the whole distributed step is tested on a 1-GPU box (tests/test_comm.py) and it
//                  also serves a single-process multi-GPU host.
//   exchanges    y halos: the rows a neighbour needs, of all fields of the exchange, packed into ONE message per direction;
//                Poisson: local x R2C -> transposing pack per peer -> all-to-all (W-1 messages out, all links busy) -> y FFT,
//                Thomas solve in z, inverse y FFT -> all-to-all back -> x C2R;  the phi row below the slab: one row.
//   overlap      the y-halo exchange that closes a stage runs on a side stream while the next stage's tendency kernels work on
//                the interior tile rows; the two edge tile rows are launched when the halos have landed.
//   step         bz_time_step_anelastic on a slab context with a communicator = the lean whole-step seam (bz_step.hip) with these
//                exchanges in between; no host synchronisation inside a step with the RCCL transport.
#include <dlfcn.h>

#include <condition_variable>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <map>
#include <mutex>

#include "bz_internal.h"

// ---- kernels defined elsewhere -------------------------------------------------------------------------------------------------
__global__ void k_tridiag_solve(int NXH, int Ny, int Nz, const double *__restrict__ lower, const double *__restrict__ ibeta,
                                const double *__restrict__ tfac, double2 *__restrict__ hat, double scale, int mean_column);

// out[(r) * (W B) + q B + b] = in[q][r * B + b]: gathers the W received blocks (each `rows` rows of B complex) side by side
__global__ __launch_bounds__(256) void k_concat_blocks(const double2 *__restrict__ in, double2 *__restrict__ out, long long rows, int B, int W)
{
    const long long total = rows * B * W;
    for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
        const long long r = e / ((long long)B * W);
        const int rem = (int)(e % ((long long)B * W));
        const int q = rem / B, b = rem % B;
        out[e] = in[(long long)q * rows * B + r * B + b];
    }
}

// ---- transports ----------------------------------------------------------------------------------------------------------------
struct Transport {
    virtual ~Transport() {}
    virtual int group_start() = 0;
    virtual int send(const void *buf, size_t bytes, int peer, hipStream_t st) = 0;
    virtual int recv(void *buf, size_t bytes, int peer, hipStream_t st) = 0;
    virtual int group_end(hipStream_t st) = 0;
    virtual const char *name() const = 0;
    std::string err;
};

// RCCL through dlopen: no link-time dependency, and the process keeps a single librccl (PyTorch's when it is loaded first)
namespace rccl {
typedef struct { char internal[128]; } UniqueId;
typedef void *Comm;
typedef int (*GetUniqueId_t)(UniqueId *);
typedef int (*CommInitRank_t)(Comm *, int, UniqueId, int);
typedef int (*CommDestroy_t)(Comm);
typedef int (*Group_t)();
typedef int (*Send_t)(const void *, size_t, int, int, Comm, hipStream_t);
typedef int (*Recv_t)(void *, size_t, int, int, Comm, hipStream_t);
typedef const char *(*ErrStr_t)(int);
struct Api {
    void *h = nullptr;
    GetUniqueId_t GetUniqueId = nullptr;
    CommInitRank_t CommInitRank = nullptr;
    CommDestroy_t CommDestroy = nullptr;
    Group_t GroupStart = nullptr, GroupEnd = nullptr;
    Send_t Send = nullptr;
    Recv_t Recv = nullptr;
    ErrStr_t ErrStr = nullptr;
};
static Api *api(std::string &err)
{
    static Api A;
    static std::mutex m;
    std::lock_guard<std::mutex> lk(m);
    if (A.h) return &A;
    const char *names[] = {getenv("BZ_RCCL_LIB"), "librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so.1"};
    for (const char *n : names) {
        if (!n) continue;
        A.h = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
        if (A.h) break;
    }
    if (!A.h) { err = std::string("cannot dlopen librccl: ") + dlerror(); return nullptr; }
    A.GetUniqueId = (GetUniqueId_t)dlsym(A.h, "ncclGetUniqueId");
    A.CommInitRank = (CommInitRank_t)dlsym(A.h, "ncclCommInitRank");
    A.CommDestroy = (CommDestroy_t)dlsym(A.h, "ncclCommDestroy");
    A.GroupStart = (Group_t)dlsym(A.h, "ncclGroupStart");
    A.GroupEnd = (Group_t)dlsym(A.h, "ncclGroupEnd");
    A.Send = (Send_t)dlsym(A.h, "ncclSend");
    A.Recv = (Recv_t)dlsym(A.h, "ncclRecv");
    A.ErrStr = (ErrStr_t)dlsym(A.h, "ncclGetErrorString");
    if (!A.GetUniqueId || !A.CommInitRank || !A.CommDestroy || !A.GroupStart || !A.GroupEnd || !A.Send || !A.Recv) {
        err = "librccl lacks a required symbol";
        A.h = nullptr;
        return nullptr;
    }
    return &A;
}
}  // namespace rccl

struct RcclTransport : Transport {
    rccl::Api *A = nullptr;
    rccl::Comm comm = nullptr;
    int check(int rc, const char *what)
    {
        if (rc == 0) return BZ_OK;
        err = std::string(what) + ": " + (A->ErrStr ? A->ErrStr(rc) : "nccl error") + " (" + std::to_string(rc) + ")";
        return -2000 - rc;
    }
    ~RcclTransport() override { if (comm) A->CommDestroy(comm); }
    int group_start() override { return check(A->GroupStart(), "ncclGroupStart"); }
    int send(const void *buf, size_t bytes, int peer, hipStream_t st) override { return check(A->Send(buf, bytes, 0 /* ncclInt8 */, peer, comm, st), "ncclSend"); }
    int recv(void *buf, size_t bytes, int peer, hipStream_t st) override { return check(A->Recv(buf, bytes, 0, peer, comm, st), "ncclRecv"); }
    int group_end(hipStream_t) override { return check(A->GroupEnd(), "ncclGroupEnd"); }
    const char *name() const override { return "rccl"; }
};

// in-process transport: a named group shared by the contexts (host threads) of one process
struct LocalMsg { const void *ptr; size_t bytes; hipEvent_t ready; };
struct LocalGroup {
    int nranks = 0, joined = 0;
    std::mutex m;
    std::condition_variable cv;
    std::map<std::pair<int, int>, std::deque<LocalMsg>> box;          // (src, dst) -> posted sends
    std::map<std::pair<int, int>, std::deque<hipEvent_t>> done;        // (src, dst) -> "copied out" events, in send order
    bool aborted = false;
};
static std::mutex g_groups_m;
static std::map<std::string, LocalGroup *> g_groups;

struct LocalTransport : Transport {
    LocalGroup *G = nullptr;
    int rank = 0;
    std::vector<int> pending_peers;          // sends of the open group
    ~LocalTransport() override
    {
        std::lock_guard<std::mutex> lk(g_groups_m);
        if (G && --G->joined == 0) {
            for (auto it = g_groups.begin(); it != g_groups.end(); ++it)
                if (it->second == G) { g_groups.erase(it); break; }
            delete G;
        }
    }
    int group_start() override { pending_peers.clear(); return BZ_OK; }
    int send(const void *buf, size_t bytes, int peer, hipStream_t st) override
    {
        hipEvent_t ev;
        if (hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess || hipEventRecord(ev, st) != hipSuccess) { err = "local send: event"; return BZ_ERR_ALLOC; }
        {
            std::lock_guard<std::mutex> lk(G->m);
            G->box[{rank, peer}].push_back({buf, bytes, ev});
        }
        G->cv.notify_all();
        pending_peers.push_back(peer);
        return BZ_OK;
    }
    int recv(void *buf, size_t bytes, int peer, hipStream_t st) override
    {
        LocalMsg msg;
        {
            std::unique_lock<std::mutex> lk(G->m);
            auto &q = G->box[{peer, rank}];
            if (!G->cv.wait_for(lk, std::chrono::seconds(120), [&] { return !q.empty() || G->aborted; }) || G->aborted) {
                G->aborted = true;
                G->cv.notify_all();
                err = "local recv: no matching send within 120 s (a peer rank failed?)";
                return BZ_ERR_INVALID;
            }
            msg = q.front();
            q.pop_front();
        }
        if (msg.bytes != bytes) { err = "local recv: message size mismatch"; return BZ_ERR_INVALID; }
        hipEvent_t dn;
        if (hipStreamWaitEvent(st, msg.ready, 0) != hipSuccess || hipMemcpyAsync(buf, msg.ptr, bytes, hipMemcpyDefault, st) != hipSuccess ||
            hipEventCreateWithFlags(&dn, hipEventDisableTiming) != hipSuccess || hipEventRecord(dn, st) != hipSuccess) {
            err = "local recv: copy";
            return BZ_ERR_INVALID;
        }
        hipEventDestroy(msg.ready);
        {
            std::lock_guard<std::mutex> lk(G->m);
            G->done[{peer, rank}].push_back(dn);
        }
        G->cv.notify_all();
        return BZ_OK;
    }
    int group_end(hipStream_t st) override
    {   // my send buffers may be reused once the receivers' copies are ordered before whatever I enqueue next
        for (int peer : pending_peers) {
            hipEvent_t dn;
            {
                std::unique_lock<std::mutex> lk(G->m);
                auto &q = G->done[{rank, peer}];
                if (!G->cv.wait_for(lk, std::chrono::seconds(120), [&] { return !q.empty() || G->aborted; }) || G->aborted) {
                    G->aborted = true;
                    G->cv.notify_all();
                    err = "local group_end: a receiver never took the message";
                    return BZ_ERR_INVALID;
                }
                dn = q.front();
                q.pop_front();
            }
            hipStreamWaitEvent(st, dn, 0);
            hipEventDestroy(dn);
        }
        pending_peers.clear();
        return BZ_OK;
    }
    const char *name() const override { return "local"; }
};

// ---- the communicator of a context ----------------------------------------------------------------------------------------------
#define BZ_COMM_MAX_FIELDS 24
static_assert(BZ_COMM_MAX_FIELDS >= 16 + BZ_MAX_TRACERS, "a stage exchange carries up to 16 model fields and every user tracer");
struct BzComm {
    Transport *T = nullptr;
    int W = 1, rank = 0, upper = 0, lower = 0;
    hipStream_t side = nullptr;                  // exchanges that overlap kernels of the main stream
    hipEvent_t ev_main = nullptr, ev_side = nullptr;
    hipEvent_t ev_chunk[4] = {nullptr, nullptr, nullptr, nullptr};      // level chunks of the pipelined all-to-all (BZ_A2A_MAX_CHUNKS)
    bool overlap = true, halo_pending = false;
    bool self_messages = false;                  // BZ_COMM_SELF_MESSAGES=1: a single rank sends to itself through the transport (tests)
    // persistent buffers (device)
    double *halo_send[2] = {nullptr, nullptr}, *halo_recv[2] = {nullptr, nullptr};
    size_t halo_cap = 0;
    double *rhs = nullptr;                       // Nx Ny Nz real: source term, then the solution phi
    double *hatx = nullptr;                      // (Nz, Ny, nxh) complex; reused as (Nz, Ny, nxh_pad) on the way back
    double *xsend = nullptr, *xrecv = nullptr;   // W blocks of Nz nkx Ny complex
    double *spec = nullptr;                      // (Nz, nkx, Ny_global) complex
    double *row_send = nullptr, *phi_below = nullptr;   // Nz Nx
    double *gather = nullptr;                    // W blocks of gather_cap doubles (all-reduce of small column data)
    size_t gather_cap = 0;
    // accounting
    long long bytes_sent = 0;
    int exchanges = 0;
};

static int comm_fail(bz_ctx *ctx, int rc, const char *where)
{
    if (rc) ctx->last_error = std::string(where) + ": " + ctx->comm->T->err;
    return rc;
}

void bzi_comm_teardown(bz_ctx *ctx)
{
    BzComm *c = ctx->comm;
    if (!c) return;
    if (c->side) hipStreamSynchronize(c->side);
    for (int d = 0; d < 2; ++d) { if (c->halo_send[d]) hipFree(c->halo_send[d]); if (c->halo_recv[d]) hipFree(c->halo_recv[d]); }
    double *bufs[] = {c->rhs, c->hatx, c->xsend, c->xrecv, c->spec, c->row_send, c->phi_below, c->gather};
    for (double *b : bufs) if (b) hipFree(b);
    if (c->ev_main) hipEventDestroy(c->ev_main);
    if (c->ev_side) hipEventDestroy(c->ev_side);
    for (hipEvent_t e : c->ev_chunk) if (e) hipEventDestroy(e);
    if (c->side) hipStreamDestroy(c->side);
    delete c->T;
    delete c;
    ctx->comm = nullptr;
}

static int comm_attach(bz_ctx *ctx, Transport *T)
{
    if (ctx->comm) bzi_comm_teardown(ctx);
    const DevGrid &g = ctx->dg;
    BzComm *c = new BzComm();
    ctx->comm = c;
    c->T = T;
    c->W = ctx->y_nranks;
    c->rank = ctx->y_rank;
    c->upper = (c->rank + 1) % c->W;
    c->lower = (c->rank + c->W - 1) % c->W;
    c->overlap = !ctx->tune.comm_no_overlap;
    c->self_messages = ctx->tune.comm_self_messages;
    BZ_HIP(hipStreamCreateWithFlags(&c->side, hipStreamNonBlocking));
    BZ_HIP(hipEventCreateWithFlags(&c->ev_main, hipEventDisableTiming));
    BZ_HIP(hipEventCreateWithFlags(&c->ev_side, hipEventDisableTiming));
    for (hipEvent_t &e : c->ev_chunk) BZ_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    if (ctx->compressible) return BZ_OK;          // halo exchanges only: the acoustic column solve is rank-local
    const size_t nreal = (size_t)g.Nx * g.Ny * g.Nz;
    const int nxh = g.Nx / 2 + 1, nxh_pad = ctx->nkx * c->W;
    const size_t nhat = (size_t)g.Nz * g.Ny * (size_t)(nxh_pad > nxh ? nxh_pad : nxh);
    const size_t nblk = (size_t)g.Nz * ctx->nkx * g.Ny;       // complex elements of one transposed block
    BZ_HIP(hipMalloc(&c->rhs, nreal * sizeof(double)));
    BZ_HIP(hipMalloc(&c->hatx, nhat * 2 * sizeof(double)));
    BZ_HIP(hipMalloc(&c->xsend, nblk * c->W * 2 * sizeof(double)));
    BZ_HIP(hipMalloc(&c->xrecv, nblk * c->W * 2 * sizeof(double)));
    BZ_HIP(hipMalloc(&c->spec, nblk * c->W * 2 * sizeof(double)));
    BZ_HIP(hipMalloc(&c->row_send, (size_t)g.Nz * g.Nx * sizeof(double)));
    BZ_HIP(hipMalloc(&c->phi_below, (size_t)g.Nz * g.Nx * sizeof(double)));
    return BZ_OK;
}

extern "C" int bz_comm_unique_id(void *out128)
{
    if (!out128) return BZ_ERR_INVALID;
    std::string err;
    rccl::Api *A = rccl::api(err);
    if (!A) { fprintf(stderr, "bz_comm_unique_id: %s\n", err.c_str()); return BZ_ERR_UNSUPPORTED; }
    rccl::UniqueId id;
    const int rc = A->GetUniqueId(&id);
    if (rc) return -2000 - rc;
    std::memcpy(out128, &id, 128);
    return BZ_OK;
}

extern "C" int bz_comm_init_rccl(bz_ctx *ctx, const void *id128)
{
    if (!ctx || !id128) return BZ_ERR_INVALID;
    if (!ctx->slab_mode) { ctx->last_error = "bz_comm_init_rccl: y-slab contexts only (bz_create_slab)"; return BZ_ERR_UNSUPPORTED; }
    RcclTransport *T = new RcclTransport();
    T->A = rccl::api(T->err);
    if (!T->A) { ctx->last_error = "bz_comm_init_rccl: " + T->err; delete T; return BZ_ERR_UNSUPPORTED; }
    rccl::UniqueId id;
    std::memcpy(&id, id128, 128);
    const int rc = T->A->CommInitRank(&T->comm, ctx->y_nranks, id, ctx->y_rank);
    if (rc) { ctx->last_error = std::string("ncclCommInitRank: ") + (T->A->ErrStr ? T->A->ErrStr(rc) : "error"); delete T; return -2000 - rc; }
    return comm_attach(ctx, T);
}

extern "C" int bz_comm_init_local(bz_ctx *ctx, const char *group_name)
{
    if (!ctx || !group_name) return BZ_ERR_INVALID;
    if (!ctx->slab_mode) { ctx->last_error = "bz_comm_init_local: y-slab contexts only (bz_create_slab)"; return BZ_ERR_UNSUPPORTED; }
    LocalTransport *T = new LocalTransport();
    {
        std::lock_guard<std::mutex> lk(g_groups_m);
        LocalGroup *&G = g_groups[group_name];
        if (!G) { G = new LocalGroup(); G->nranks = ctx->y_nranks; }
        if (G->nranks != ctx->y_nranks) { ctx->last_error = "bz_comm_init_local: group exists with another size"; delete T; return BZ_ERR_INVALID; }
        ++G->joined;
        T->G = G;
    }
    T->rank = ctx->y_rank;
    return comm_attach(ctx, T);
}

extern "C" int bz_comm_destroy(bz_ctx *ctx)
{
    if (!ctx) return BZ_ERR_INVALID;
    bzi_comm_teardown(ctx);
    return BZ_OK;
}

extern "C" int bz_comm_info(bz_ctx *ctx, const char **transport, int64_t *bytes_sent, int32_t *exchanges)
{
    if (!ctx || !ctx->comm) return BZ_ERR_INVALID;
    if (transport) *transport = ctx->comm->T->name();
    if (bytes_sent) *bytes_sent = ctx->comm->bytes_sent;
    if (exchanges) *exchanges = ctx->comm->exchanges;
    return BZ_OK;
}

// ---- y-halo exchange --------------------------------------------------------------------------------------------------------------
// Fill `width` rows of the upper halo (parent rows Hy+Ny ...) and / or of the lower halo (rows Hy-width .. Hy-1) of n parent arrays
// from the ring neighbours, on stream st.  levels[m]: z levels of array m (Nz + 2 Hz, or + 1 for z-face fields).
// half: the fields are Float32 arrays of a Float64 model (the substepper's working fields with substep_floattype = Float32): their rows
// travel as Sx / 2 doubles (Sx is even: Nx is even on every context that decomposes)
static int halo_exchange(bz_ctx *ctx, double *const *fields, const int32_t *levels, int n, int width, bool upper_halo, bool lower_halo,
                         hipStream_t st, bool half = false)
{
    BzComm *c = ctx->comm;
    const DevGrid &g = ctx->dg;
    if (n < 1 || n > BZ_COMM_MAX_FIELDS || width < 1 || width > g.Hy) return BZ_ERR_INVALID;
    if (half && ((g.Sx & 1) || sizeof(double) != 8)) return BZ_ERR_UNSUPPORTED;
    const int sx = half ? g.Sx / 2 : g.Sx;
    const long long sxy = half ? g.Sxy / 2 : g.Sxy;
    size_t total = 0;
    for (int m = 0; m < n; ++m) total += (size_t)levels[m] * width * sx;
    if (total > c->halo_cap) {
        BZ_HIP(hipStreamSynchronize(st));
        for (int d = 0; d < 2; ++d) {
            if (c->halo_send[d]) hipFree(c->halo_send[d]);
            if (c->halo_recv[d]) hipFree(c->halo_recv[d]);
            BZ_HIP(hipMalloc(&c->halo_send[d], total * sizeof(double)));
            BZ_HIP(hipMalloc(&c->halo_recv[d], total * sizeof(double)));
        }
        c->halo_cap = total;
    }
    hipStream_t keep = ctx->stream;
    ctx->stream = st;                            // bz_pack_rows launches on the context's stream
    int rc = BZ_OK;
    const size_t bytes = total * sizeof(double);
    // what the UPPER neighbour's lower halo needs: my top interior rows; what the LOWER neighbour's upper halo needs: my first rows
    if (lower_halo) rc = bzi_pack_rows_geom(ctx, fields, levels, n, g.Hy + g.Ny - width, width, c->halo_send[0], 0, sx, sxy);
    if (!rc && upper_halo) rc = bzi_pack_rows_geom(ctx, fields, levels, n, g.Hy, width, c->halo_send[1], 0, sx, sxy);
    if (!rc && c->W == 1 && !c->self_messages) {  // periodic wrap onto myself: no message
        if (lower_halo) rc = bzi_pack_rows_geom(ctx, fields, levels, n, g.Hy - width, width, c->halo_send[0], 1, sx, sxy);
        if (!rc && upper_halo) rc = bzi_pack_rows_geom(ctx, fields, levels, n, g.Hy + g.Ny, width, c->halo_send[1], 1, sx, sxy);
        ctx->stream = keep;
        return rc;
    }
    if (!rc) {
        // order matters when upper == lower (two ranks): sends "up, then down", receives "from lower, then from upper"
        rc = comm_fail(ctx, c->T->group_start(), "halo exchange");
        if (!rc && lower_halo) rc = comm_fail(ctx, c->T->send(c->halo_send[0], bytes, c->upper, st), "halo exchange");
        if (!rc && upper_halo) rc = comm_fail(ctx, c->T->send(c->halo_send[1], bytes, c->lower, st), "halo exchange");
        if (!rc && lower_halo) rc = comm_fail(ctx, c->T->recv(c->halo_recv[0], bytes, c->lower, st), "halo exchange");
        if (!rc && upper_halo) rc = comm_fail(ctx, c->T->recv(c->halo_recv[1], bytes, c->upper, st), "halo exchange");
        if (!rc) rc = comm_fail(ctx, c->T->group_end(st), "halo exchange");
        c->bytes_sent += (long long)bytes * ((lower_halo ? 1 : 0) + (upper_halo ? 1 : 0));
        c->exchanges++;
    }
    if (!rc && lower_halo) rc = bzi_pack_rows_geom(ctx, fields, levels, n, g.Hy - width, width, c->halo_recv[0], 1, sx, sxy);
    if (!rc && upper_halo) rc = bzi_pack_rows_geom(ctx, fields, levels, n, g.Hy + g.Ny, width, c->halo_recv[1], 1, sx, sxy);
    ctx->stream = keep;
    return rc;
}

extern "C" int bz_comm_exchange_y_halos(bz_ctx *ctx, double *const *fields, const int32_t *levels, int32_t n)
{
    if (!ctx || !ctx->comm || !fields || !levels) return BZ_ERR_INVALID;
    ProfileScope ps(ctx, "comm_halo_exchange");
    return halo_exchange(ctx, fields, levels, n, ctx->dg.Hy, true, true, ctx->stream);
}

// ---- distributed Fourier-tridiagonal solve: c->rhs (source term) -> c->rhs (zero-mean solution), all on the main stream ------------
// part of every block: `count` doubles starting `offset` doubles into each of the W blocks (count = 0: whole blocks), on stream st
static int all_to_all(bz_ctx *ctx, const double *send, double *recv, size_t block_doubles, size_t offset = 0, size_t count = 0,
                      hipStream_t st = nullptr)
{
    BzComm *c = ctx->comm;
    if (!st) st = ctx->stream;
    if (!count) count = block_doubles;
    const size_t bytes = count * sizeof(double);
    const bool self = c->self_messages;
    if (!self) BZ_HIP(hipMemcpyAsync(recv + block_doubles * c->rank + offset, send + block_doubles * c->rank + offset, bytes, hipMemcpyDeviceToDevice, st));
    if (c->W == 1 && !self) return BZ_OK;
    int rc = comm_fail(ctx, c->T->group_start(), "all-to-all");
    for (int p = 0; p < c->W && !rc; ++p)
        if (p != c->rank || self) rc = comm_fail(ctx, c->T->send(send + block_doubles * p + offset, bytes, p, st), "all-to-all");
    for (int p = 0; p < c->W && !rc; ++p)
        if (p != c->rank || self) rc = comm_fail(ctx, c->T->recv(recv + block_doubles * p + offset, bytes, p, st), "all-to-all");
    if (!rc) rc = comm_fail(ctx, c->T->group_end(st), "all-to-all");
    c->bytes_sent += (long long)bytes * (c->W - 1);
    c->exchanges++;
    return rc;
}

// level chunks of the pipelined transposes: the x transforms walk 16 levels per block, so chunk edges are multiples of 16; columns
// shorter than 64 levels go in one piece
#define BZ_A2A_MAX_CHUNKS 4
static int a2a_chunks(const DevGrid &g, int edges[BZ_A2A_MAX_CHUNKS + 1])
{
    int n = (g.Nz >= 64 && g.Nz % 16 == 0) ? BZ_A2A_MAX_CHUNKS : 1;
    const int units = g.Nz / 16;
    if (n > 1 && units < n) n = units;
    edges[0] = 0;
    for (int i = 1; i <= n; ++i) edges[i] = (n == 1) ? g.Nz : 16 * (int)(((long long)units * i) / n);
    return n;
}

// With the hand-written x transforms (ctx->xf_slab, bz_xfft_kernels.h) the source term is evaluated inside the forward x pass, which
// stores the W messages of the first all-to-all directly (no rhs round trip, no library x transform, no pack launches), and the inverse x
// pass reads the W messages of the second all-to-all where they landed (no concatenation); the blocks sent back are plain strided copies.
static int dist_poisson(bz_ctx *ctx, const bz_state *s, const bz_prognostic *predictor, double dt)
{
    BzComm *c = ctx->comm;
    const DevGrid &g = ctx->dg;
    const int W = c->W, nkx = ctx->nkx, nxh = g.Nx / 2 + 1, NyG = ctx->Ny_global;
    const size_t blk = (size_t)g.Nz * nkx * g.Ny * 2;            // doubles per transposed block
    const bool direct = (W == 1 && !c->self_messages);          // one rank: the packs write the transposed arrays in place
    int rc;
    if (ctx->xf_slab) {
        // The all-to-all is pipelined with the x transforms in level chunks (a block of a message is (Nz, nkx, Ny): a level range is
        // contiguous): chunk c's messages travel on the communication stream while the main stream transforms chunk c + 1, and on the
        // way back chunk c is transformed while chunk c + 1 travels.  Only the first / last chunk's transform stays exposed.
        int edges[BZ_A2A_MAX_CHUNKS + 1];
        const int nch = direct ? 1 : a2a_chunks(g, edges);
        const size_t lev_doubles = (size_t)nkx * g.Ny * 2;      // doubles per level of one block
        if (direct) {
            ProfileScope ps(ctx, "poisson_source_term+fft_x");
            if ((rc = bzi_xf_forward(ctx, s, dt, predictor, nullptr, c->spec, W))) return rc;
        } else {
            for (int ch = 0; ch < nch; ++ch) {
                {
                    ProfileScope ps(ctx, "poisson_source_term+fft_x");
                    if ((rc = bzi_xf_forward(ctx, s, dt, predictor, nullptr, c->xsend, W, edges[ch], edges[ch + 1]))) return rc;
                }
                BZ_HIP(hipEventRecord(c->ev_chunk[ch], ctx->stream));
                BZ_HIP(hipStreamWaitEvent(c->side, c->ev_chunk[ch], 0));
                hipStream_t keep = ctx->stream;
                ctx->stream = c->side;
                {
                    ProfileScope ps(ctx, "comm_all_to_all");
                    rc = all_to_all(ctx, c->xsend, c->xrecv, blk, lev_doubles * edges[ch], lev_doubles * (edges[ch + 1] - edges[ch]), c->side);
                }
                ctx->stream = keep;
                if (rc) return rc;
            }
            BZ_HIP(hipEventRecord(c->ev_side, c->side));
            BZ_HIP(hipStreamWaitEvent(ctx->stream, c->ev_side, 0));
            hipLaunchKernelGGL(k_concat_blocks, dim3(4096), dim3(256), 0, ctx->stream, (const double2 *)c->xrecv, (double2 *)c->spec,
                               (long long)g.Nz * nkx, g.Ny, W);                                            // (Nz, nkx, Ny_global)
            BZ_LAUNCH_CHECK();
        }
        if ((rc = bz_slab_transform(ctx, 1, c->spec, c->spec, 0))) return rc;
        if ((rc = bz_spectral_tridiagonal_solve(ctx, c->spec, 1.0 / ((double)g.Nx * (double)NyG)))) return rc;
        if ((rc = bz_slab_transform(ctx, 2, c->spec, c->spec, 0))) return rc;
        if (!direct) {
            {
                ProfileScope ps(ctx, "poisson_split_rows");
                const size_t rowb = (size_t)g.Ny * 2 * sizeof(double);                                     // the destination's rows of one (k, kx)
                for (int q = 0; q < W; ++q)
                    BZ_HIP(hipMemcpy2DAsync(c->xsend + blk * q, rowb, c->spec + (size_t)q * g.Ny * 2, (size_t)NyG * 2 * sizeof(double), rowb,
                                            (size_t)g.Nz * nkx, hipMemcpyDeviceToDevice, ctx->stream));
            }
            BZ_HIP(hipEventRecord(c->ev_main, ctx->stream));
            BZ_HIP(hipStreamWaitEvent(c->side, c->ev_main, 0));
            hipStream_t keep = ctx->stream;
            ctx->stream = c->side;
            for (int ch = 0; ch < nch && !rc; ++ch) {
                {
                    ProfileScope ps(ctx, "comm_all_to_all");
                    rc = all_to_all(ctx, c->xsend, c->xrecv, blk, lev_doubles * edges[ch], lev_doubles * (edges[ch + 1] - edges[ch]), c->side);
                }
                if (!rc && hipEventRecord(c->ev_chunk[ch], c->side) != hipSuccess) rc = BZ_ERR_INVALID;
            }
            ctx->stream = keep;
            if (rc) return rc;
            for (int ch = 0; ch < nch; ++ch) {
                BZ_HIP(hipStreamWaitEvent(ctx->stream, c->ev_chunk[ch], 0));
                ProfileScope ps(ctx, "poisson_fft_x_inverse");
                if ((rc = bzi_xf_inverse(ctx, c->xrecv, c->rhs, W, edges[ch], edges[ch + 1]))) return rc;
            }
            return BZ_OK;
        }
        ProfileScope ps(ctx, "poisson_fft_x_inverse");
        return bzi_xf_inverse(ctx, c->spec, c->rhs, W);
    }
    if ((rc = bzi_poisson_source_fused(ctx, s, dt, c->rhs, predictor))) return rc;
    if ((rc = bz_slab_transform(ctx, 0, c->rhs, c->hatx, 0))) return rc;                                  // (Nz, Ny, nxh)
    for (int p = 0; p < W; ++p)                                                                            // -> W x (Nz, nkx, Ny)
        if ((rc = bz_pack_transpose(ctx, c->hatx, (direct ? c->spec : c->xsend) + blk * p, g.Nz, g.Ny, nxh, p * nkx, nkx, nxh))) return rc;
    if (!direct) {
        ProfileScope ps(ctx, "comm_all_to_all");
        if ((rc = all_to_all(ctx, c->xsend, c->xrecv, blk))) return rc;
        hipLaunchKernelGGL(k_concat_blocks, dim3(4096), dim3(256), 0, ctx->stream, (const double2 *)c->xrecv, (double2 *)c->spec,
                           (long long)g.Nz * nkx, g.Ny, W);                                                // (Nz, nkx, Ny_global)
        BZ_LAUNCH_CHECK();
    }
    if ((rc = bz_slab_transform(ctx, 1, c->spec, c->spec, 0))) return rc;
    if ((rc = bz_spectral_tridiagonal_solve(ctx, c->spec, 1.0 / ((double)g.Nx * (double)NyG)))) return rc;
    if ((rc = bz_slab_transform(ctx, 2, c->spec, c->spec, 0))) return rc;
    for (int q = 0; q < W; ++q)                                                                            // -> W x (Nz, Ny, nkx)
        if ((rc = bz_pack_transpose(ctx, c->spec, (direct ? c->hatx : c->xsend) + blk * q, g.Nz, nkx, NyG, q * g.Ny, g.Ny, NyG))) return rc;
    if (!direct) {
        ProfileScope ps(ctx, "comm_all_to_all");
        if ((rc = all_to_all(ctx, c->xsend, c->xrecv, blk))) return rc;
        hipLaunchKernelGGL(k_concat_blocks, dim3(4096), dim3(256), 0, ctx->stream, (const double2 *)c->xrecv, (double2 *)c->hatx,
                           (long long)g.Nz * g.Ny, nkx, W);                                                // (Nz, Ny, nkx W)
        BZ_LAUNCH_CHECK();
    }
    return bz_slab_transform(ctx, 3, c->hatx, c->rhs, nkx * W);
}

// phi of the row below the slab (the neighbour's top row) into c->phi_below [k][i]
static int exchange_phi_below(bz_ctx *ctx)
{
    BzComm *c = ctx->comm;
    const DevGrid &g = ctx->dg;
    ProfileScope ps(ctx, "comm_phi_row");
    const size_t rowb = (size_t)g.Nx * sizeof(double);
    const bool direct = (c->W == 1 && !c->self_messages);
    double *dst = direct ? c->phi_below : c->row_send;
    BZ_HIP(hipMemcpy2DAsync(dst, rowb, c->rhs + (size_t)(g.Ny - 1) * g.Nx, rowb * g.Ny, rowb, g.Nz, hipMemcpyDeviceToDevice, ctx->stream));
    if (direct) return BZ_OK;
    const size_t bytes = rowb * g.Nz;
    int rc = comm_fail(ctx, c->T->group_start(), "phi row");
    if (!rc) rc = comm_fail(ctx, c->T->send(c->row_send, bytes, c->upper, ctx->stream), "phi row");
    if (!rc) rc = comm_fail(ctx, c->T->recv(c->phi_below, bytes, c->lower, ctx->stream), "phi row");
    if (!rc) rc = comm_fail(ctx, c->T->group_end(ctx->stream), "phi row");
    c->bytes_sent += (long long)bytes;
    c->exchanges++;
    return rc;
}

// compute_pressure_correction! + make_pressure_correction! + the diagnosis, from the momentum in `s` (predictor == nullptr) or from
// the predictor arrays; lean: momentum-only projection (stages whose diagnostics nobody reads)
static int dist_projection(bz_ctx *ctx, const bz_state *s, const bz_prognostic *predictor, double dt, bool lean, double *oa, double *ob,
                           const double *rtheta_in, const double *rq_in, bool join_side = false)
{
    BzComm *c = ctx->comm;
    const DevGrid &g = ctx->dg;
    int rc;
    {   // the divergence needs row Ny of rho_v: the upper neighbour's first row
        ProfileScope ps(ctx, "comm_halo_exchange");
        double *f[1] = {predictor ? predictor->rho_v : s->rho_v};
        int32_t lev[1] = {g.Nz + 2 * g.Hz};
        if ((rc = halo_exchange(ctx, f, lev, 1, 1, true, false, ctx->stream))) return rc;
    }
    if ((rc = dist_poisson(ctx, s, predictor, dt))) return rc;
    if ((rc = exchange_phi_below(ctx))) return rc;
    if (join_side) BZ_HIP(hipStreamWaitEvent(ctx->stream, ctx->ev_join, 0));      // the scalar-pair kernel of this stage has finished
    if (lean) return bzi_project_lean(ctx, s, dt, c->rhs, c->phi_below, predictor, oa, ob);
    return bzi_project_diagnose(ctx, s, dt, c->rhs, c->phi_below, predictor, true, rtheta_in, rq_in);
}

// the y halos every consumer of the state may read: momentum, the two scalars' densities, and the diagnostics the per-operator
// tendency kernels use
static int state_halo_exchange(bz_ctx *ctx, const bz_state *s, double *pa, double *pb, bool diagnostics, hipStream_t st)
{
    const DevGrid &g = ctx->dg;
    const int nc = g.Nz + 2 * g.Hz, nf = nc + 1;
    double *f[BZ_COMM_MAX_FIELDS] = {s->rho_u, s->rho_v, s->rho_w, pa, pb};
    int32_t lev[BZ_COMM_MAX_FIELDS] = {nc, nc, nf, nc, nc};
    int n = 5;
    if (diagnostics) {
        double *d[5] = {s->u, s->v, s->w, s->theta, s->q};
        const int32_t dl[5] = {nc, nc, nf, nc, nc};
        for (int m = 0; m < 5; ++m) { f[n] = d[m]; lev[n] = dl[m]; ++n; }
        if (ctx->has_closure || g.microphysics == 1) { f[n] = s->T; lev[n++] = nc; }      // the viscosity kernel covers the rows next to the slab
        if (g.microphysics == 1) { f[n] = g.qv_field; lev[n++] = nc; f[n] = g.ql_field; lev[n++] = nc; }
        if (g.microphysics == 2) {      // the Kessler species are advected as specific fields
            f[n] = ctx->kessler.cloud_liquid_mass_fraction; lev[n++] = nc;
            f[n] = ctx->kessler.rain_mass_fraction; lev[n++] = nc;
        }
        for (int t = 0; t < ctx->n_tracers; ++t) { f[n] = ctx->tracers[t].specific; lev[n++] = nc; }
    }
    return halo_exchange(ctx, f, lev, n, g.Hy, true, true, st);
}

// update_state!(model; compute_tendencies = false) of set! + the initial projection with dt (set_atmosphere_model.jl:121-128) on slabs
extern "C" int bz_comm_update_state_and_project(bz_ctx *ctx, const bz_state *s, const bz_prognostic *G, double dt, int project)
{
    if (!ctx || !ctx->comm || !s) return BZ_ERR_INVALID;
    int rc = bz_update_state(ctx, s, G, 0);                      // x / z halos and diagnostics locally
    if (rc) return rc;
    ctx->G_is_predictor = true;                                  // the state changed and no tendencies were computed: G is stale
    {
        ProfileScope ps(ctx, "comm_halo_exchange");
        if ((rc = state_halo_exchange(ctx, s, s->rho_theta, s->rho_q, true, ctx->stream))) return rc;
    }
    if (!project) return BZ_OK;
    if ((rc = dist_projection(ctx, s, nullptr, dt, false, nullptr, nullptr, nullptr, nullptr))) return rc;
    ProfileScope ps(ctx, "comm_halo_exchange");
    return state_halo_exchange(ctx, s, s->rho_theta, s->rho_q, true, ctx->stream);
}

// time_step!(model, dt) on y-slabs: the lean whole-step seam of bz_step.hip with the exchanges in between
// ---- small all-reduce (column data: horizontal averages of the forcing stack) ---------------------------------------------------------
__global__ void k_sum_blocks(const double *__restrict__ blocks, double *__restrict__ out, int n, int W)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    double acc = blocks[t];
    for (int p = 1; p < W; ++p) acc += blocks[(size_t)p * n + t];      // rank order: the same bits on every rank
    out[t] = acc;
}

// buf[0 .. n) <- sum over the ranks of the slab communicator.  Every rank sends its block to every other one (n is a few thousand) and
// adds the W blocks in rank order, so all ranks hold identical bits whatever the transport.
int bzi_comm_allreduce_sum(bz_ctx *ctx, double *buf, int n)
{
    BzComm *c = ctx->comm;
    if (!c || n < 1) return BZ_ERR_INVALID;
    if (c->W == 1 && !c->self_messages) return BZ_OK;
    ProfileScope ps(ctx, "comm_allreduce");
    if ((size_t)n > c->gather_cap) {
        BZ_HIP(hipStreamSynchronize(ctx->stream));
        if (c->gather) hipFree(c->gather);
        BZ_HIP(hipMalloc(&c->gather, (size_t)n * c->W * sizeof(double)));
        c->gather_cap = (size_t)n;
    }
    const size_t bytes = (size_t)n * sizeof(double);
    const bool self = c->self_messages;
    if (!self) BZ_HIP(hipMemcpyAsync(c->gather + (size_t)c->rank * n, buf, bytes, hipMemcpyDeviceToDevice, ctx->stream));
    int rc = comm_fail(ctx, c->T->group_start(), "all-reduce");
    for (int p = 0; p < c->W && !rc; ++p)
        if (p != c->rank || self) rc = comm_fail(ctx, c->T->send(buf, bytes, p, ctx->stream), "all-reduce");
    for (int p = 0; p < c->W && !rc; ++p)
        if (p != c->rank || self) rc = comm_fail(ctx, c->T->recv(c->gather + (size_t)p * n, bytes, p, ctx->stream), "all-reduce");
    if (!rc) rc = comm_fail(ctx, c->T->group_end(ctx->stream), "all-reduce");
    if (rc) return rc;
    c->bytes_sent += (long long)bytes * (c->W - 1);
    c->exchanges++;
    hipLaunchKernelGGL(k_sum_blocks, dim3((n + 255) / 256), dim3(256), 0, ctx->stream, c->gather, buf, n, c->W);
    BZ_LAUNCH_CHECK();
    return BZ_OK;
}

// ---- distributed step of the models the lean seam does not cover: saturation adjustment, SmagorinskyLilly, column forcings, bottom
// fluxes (the physics list of BASELINE configs[2]) — the fused-RK tier of bz_time_step_anelastic (bz_step.hip) with the exchanges of the
// slab decomposition: per stage the tendency kernels with the RK update folded in, the closure (its viscosity kernel covers one row
// beyond each slab edge, so nu_e needs no exchange), the forcing stack (horizontal averages all-reduced over the ranks), bottom fluxes,
// the distributed pressure solve + projection + diagnosis, and one y-halo exchange of everything the next stage's stencils read.
// Operator-by-operator distributed step: the reference's own call order (ssp_runge_kutta_3.jl:223-270) with the exchanges of
// bz_comm_update_state_and_project in place of the local halo fills.  Every model option the single-GPU per-operator tier runs takes
// this path on slabs when the fused tiers do not apply: StaticEnergy, Kessler species, bounds-preserving advection, WENO(order = 7 / 9).
static int dist_time_step_operators(bz_ctx *ctx, const bz_state *s, const bz_prognostic *U0, const bz_prognostic *G, double dt)
{
    const double alphas[3] = {1.0, 1.0 / 4.0, 2.0 / 3.0};
    int rc;
    if (ctx->G_is_predictor && (rc = bz_compute_tendencies(ctx, s, G))) return rc;      // state and halos are current since the last step's end
    if ((rc = bz_store_initial_state(ctx, s, U0))) return rc;
    for (int stage = 0; stage < 3; ++stage) {
        const double alpha = alphas[stage];
        if ((rc = bz_compute_flux_bc_tendencies(ctx, s, G))) return rc;
        if ((rc = bz_ssp_rk3_substep(ctx, s, U0, G, dt, alpha))) return rc;
        // pressure solve with its all-to-alls, projection + diagnostics in one kernel, exchange.  The diagnostics run ONCE per stage, as
        // in update_state!: with Kessler species the temperature is diagnosed from the condensate fractions of the previous update
        // (dcmip2016_kessler.jl:222-227 — the microphysical fields are refreshed after the thermodynamic variables), so a second
        // diagnosis in the same stage would not reproduce the reference
        if ((rc = dist_projection(ctx, s, nullptr, alpha * dt, false, nullptr, nullptr, nullptr, nullptr))) return rc;
        {
            ProfileScope ps(ctx, "comm_halo_exchange");
            if ((rc = state_halo_exchange(ctx, s, s->rho_theta, s->rho_q, true, ctx->stream))) return rc;
        }
        if ((rc = bz_compute_tendencies(ctx, s, G))) return rc;
    }
    if (ctx->dg.microphysics == 2) {      // microphysics_model_update! closes the step: rank-local columns, then update_state! with exchanges
        const bz_kessler_model_fields &K = ctx->kessler;
        bz_kessler_fields F;
        F.density = nullptr; F.pressure = nullptr;
        F.potential_temperature = s->theta; F.potential_temperature_density = s->rho_theta;
        F.moisture_density = s->rho_q; F.cloud_liquid_density = K.cloud_liquid_density; F.rain_density = K.rain_density;
        F.vapor_mass_fraction = K.vapor_mass_fraction; F.cloud_liquid_mass_fraction = K.cloud_liquid_mass_fraction;
        F.rain_mass_fraction = K.rain_mass_fraction; F.rain_terminal_velocity = K.rain_terminal_velocity;
        F.precipitation_rate = K.precipitation_rate;
        if ((rc = bz_kessler_microphysics_update(ctx, &ctx->kessler_params, &F, dt, ctx->kessler_pst))) return rc;
        if ((rc = bz_comm_update_state_and_project(ctx, s, G, dt, 0))) return rc;
        if ((rc = bz_compute_tendencies(ctx, s, G))) return rc;
    }
    return BZ_OK;
}

static int dist_time_step_general(bz_ctx *ctx, const bz_state *s, const bz_prognostic *U0, const bz_prognostic *G, double dt)
{
    const DevGrid &g = ctx->dg;
    if (!(ctx->fused_ok && ctx->fuse_rk && g.formulation == 0 && g.microphysics != 2 && !ctx->bounded_mask && ctx->weno_R == 3 && ctx->scalar_R == 3 && !ctx->has_relaxation))
        return dist_time_step_operators(ctx, s, U0, G, dt);
    const int32_t nc = g.Nz + 2 * g.Hz, nf = nc + 1;
    const double alphas[3] = {1.0, 1.0 / 4.0, 2.0 / 3.0};
    int rc;
    BZ_HIP(hipMemsetAsync(G->rho_w + g.Sxy * g.Hz, 0, g.Sxy * sizeof(double), ctx->stream));
    BZ_HIP(hipMemsetAsync(G->rho_w + g.Sxy * (g.Hz + g.Nz), 0, g.Sxy * sizeof(double), ctx->stream));
    for (int stage = 0; stage < 3; ++stage) {
        const double alpha = alphas[stage];
        if ((rc = bzi_tendencies_fused_rk(ctx, s, U0, G, dt, alpha, stage == 0))) return rc;
        if (ctx->n_tracers) {       // tracers ride beside the fused kernels: tendency from the previous-stage state, RK in place
            if ((rc = bzi_tracer_tendencies(ctx, s))) return rc;
            if ((rc = bzi_tracer_rk3(ctx, dt, alpha, stage == 0))) return rc;
        }
        if (ctx->has_closure && (rc = bzi_apply_closure(ctx, s, G->rho_u, G->rho_v, G->rho_w, s->rho_theta, s->rho_q, alpha * dt))) return rc;
        if (ctx->has_forcings && (rc = bzi_apply_forcings(ctx, s, G->rho_u, G->rho_v, s->rho_theta, s->rho_q, alpha * dt))) return rc;
        if ((ctx->has_forcings || ctx->has_bulk) && (rc = bzi_flux_bc(ctx, s, G->rho_u, G->rho_v, s->rho_theta, s->rho_q, alpha * dt))) return rc;
        if ((rc = dist_projection(ctx, s, G, alpha * dt, false, nullptr, nullptr, nullptr, nullptr))) return rc;
        if ((rc = bzi_tracer_specific(ctx))) return rc;
        double *f[BZ_COMM_MAX_FIELDS] = {s->rho_u, s->rho_v, s->rho_w, s->u, s->v, s->w, s->theta, s->q, s->T, s->rho_theta, s->rho_q};
        int32_t lev[BZ_COMM_MAX_FIELDS] = {nc, nc, nf, nc, nc, nf, nc, nc, nc, nc, nc};
        int n = 11;
        if (g.microphysics == 1) { f[n] = g.qv_field; lev[n++] = nc; f[n] = g.ql_field; lev[n++] = nc; }
        for (int t = 0; t < ctx->n_tracers; ++t) { f[n] = ctx->tracers[t].specific; lev[n++] = nc; }
        ProfileScope ps(ctx, "comm_halo_exchange");
        if ((rc = halo_exchange(ctx, f, lev, n, g.Hy, true, true, ctx->stream))) return rc;
    }
    ctx->G_is_predictor = true;
    return BZ_OK;
}

// an undiagnosed last stage leaves its halo exchange on the side stream: whoever reads `s` next on the main stream joins it first
int bzi_comm_join_pending(bz_ctx *ctx)
{
    BzComm *c = ctx->comm;
    if (c && c->halo_pending) {
        BZ_HIP(hipStreamWaitEvent(ctx->stream, c->ev_side, 0));
        c->halo_pending = false;
    }
    return BZ_OK;
}

int bzi_dist_time_step(bz_ctx *ctx, const bz_state *s, const bz_prognostic *U0, const bz_prognostic *G, double dt, bool diagnose)
{
    BzComm *c = ctx->comm;
    const DevGrid &g = ctx->dg;
    ctx->lean_step_last = false;
    if (!(ctx->fused_ok && ctx->weno_R == 3 && ctx->scalar_R == 3 && ctx->dg.formulation == 0 && ctx->dg.microphysics == 0 && !ctx->has_forcings && !ctx->has_relaxation && !ctx->has_bulk &&
          !ctx->has_closure && ctx->n_tracers == 0 && !ctx->bounded_mask && (long long)g.Sxy * (g.Nz + 2 * g.Hz + 1) < (1LL << 32))) {
        // these tiers start from the stored diagnostics: rebuild them if undiagnosed lean steps came before
        const int rcs = ctx->diagnostics_stale ? bz_comm_update_state_and_project(ctx, s, G, 1.0, 0) : BZ_OK;
        return rcs ? rcs : dist_time_step_general(ctx, s, U0, G, dt);
    }
    int rc;
    const double alphas[3] = {1.0, 1.0 / 4.0, 2.0 / 3.0};
    BZ_HIP(hipMemsetAsync(G->rho_w + g.Sxy * g.Hz, 0, g.Sxy * sizeof(double), ctx->stream));
    BZ_HIP(hipMemsetAsync(G->rho_w + g.Sxy * (g.Hz + g.Nz), 0, g.Sxy * sizeof(double), ctx->stream));
    BZ_HIP(hipMemsetAsync(U0->rho_w + g.Sxy * g.Hz, 0, g.Sxy * sizeof(double), ctx->stream));
    BZ_HIP(hipMemsetAsync(U0->rho_w + g.Sxy * (g.Hz + g.Nz), 0, g.Sxy * sizeof(double), ctx->stream));
    for (int stage = 0; stage < 3; ++stage) {
        const double alpha = alphas[stage];
        // buffer rotation and the undiagnosed last stage of bz_time_steps_anelastic: as in bz_step.hip (bzi_lean_stage)
        const bool full = diagnose && stage == 2;
        LeanStage LS;
        bzi_lean_stage(s, U0, G, stage, &LS);
        const bz_state *sin = &LS.sin, *sout = &LS.sout;
        const bz_prognostic *u0 = &LS.u0;
        const double *pa = LS.pa, *pb = LS.pb;
        double *oa = LS.oa, *ob = LS.ob;
        // The scalar-pair kernel feeds nothing of the pressure solve: with messages in flight (W > 1) it runs on the context's second
        // stream beside the source term, the transforms and both all-to-alls, and is joined before the projection kernel.
        const bool fork = (c->W > 1 || c->self_messages || ctx->side_scalar) && !ctx->tune.comm_no_side_scalar;
        const int first_part = fork ? 1 : 3;
        if (c->halo_pending) {
            // the halos of the stage-start state are still travelling on the side stream: interior tile rows first
            if ((rc = bzi_tendencies_lean(ctx, sin, u0, G, pa, pb, oa, ob, dt, alpha, stage == 0, 1, first_part))) return rc;
            BZ_HIP(hipStreamWaitEvent(ctx->stream, c->ev_side, 0));
            c->halo_pending = false;
            if ((rc = bzi_tendencies_lean(ctx, sin, u0, G, pa, pb, oa, ob, dt, alpha, stage == 0, 2, first_part))) return rc;
        } else if ((rc = bzi_tendencies_lean(ctx, sin, u0, G, pa, pb, oa, ob, dt, alpha, stage == 0, 0, first_part))) return rc;
        if (fork) {
            BZ_HIP(hipEventRecord(ctx->ev_fork, ctx->stream));
            BZ_HIP(hipStreamWaitEvent(ctx->side_stream, ctx->ev_fork, 0));
            hipStream_t keep = ctx->stream;
            ctx->stream = ctx->side_stream;
            rc = bzi_tendencies_lean(ctx, sin, u0, G, pa, pb, oa, ob, dt, alpha, stage == 0, 0, 2);
            ctx->stream = keep;
            if (rc) return rc;
            BZ_HIP(hipEventRecord(ctx->ev_join, ctx->side_stream));
        }
        if ((rc = dist_projection(ctx, full ? s : sout, G, alpha * dt, !full, oa, ob, oa, ob, fork))) return rc;
        // halos of the new state.  After a diagnosed stage 3 rho theta / rho q are back in `s` and the diagnostics are current: exchange
        // them too, so that every field of `s` is what the per-operator sequence leaves
        double *na = oa, *nb = ob;      // (a diagnosed stage 3 writes in place: oa, ob are the state arrays)
        const bool async = c->overlap && !full && (c->W > 1 || c->self_messages);
        hipStream_t st = async ? c->side : ctx->stream;
        if (async) {
            BZ_HIP(hipEventRecord(c->ev_main, ctx->stream));
            BZ_HIP(hipStreamWaitEvent(c->side, c->ev_main, 0));
        }
        {
            ProfileScope ps(ctx, "comm_halo_exchange");
            if ((rc = state_halo_exchange(ctx, full ? s : sout, na, nb, full, st))) return rc;
        }
        if (async) {
            BZ_HIP(hipEventRecord(c->ev_side, c->side));
            c->halo_pending = true;
        }
    }
    bzi_lean_step_done(ctx, diagnose);
    return BZ_OK;
}

// ---------------------------------------------------------------------------------------------------------------------------------
// Library-owned distributed COMPRESSIBLE step (BASELINE configs[4]: split-explicit WS-RK3 + acoustic substeps [+ Kessler] on y-slabs).
// The sequence of breeze.jl_amd/compressible.py: SlabCompressibleModel.time_step_slab, issued from here so that a host needs one call
// per step: inside a stage only the two perturbation fields the next substep reads across the slab edge are exchanged per substep
// ((rho theta)' and (rho v)' of the current ping-pong buffer, Hy rows each way); G_rho_v, the recovered prognostics, the diagnostics
// and the time-averaged velocities travel once per stage, after the diagnosis kernel has written the x-halo images of the edge rows.
// Reference: acoustic_rk3_substep! / time_step! (/root/reference/src/TimeSteppers/acoustic_runge_kutta_3.jl:172-208,264-319),
// halo fills of the substep loop (/root/reference/src/CompressibleEquations/acoustic_substepping.jl:1462-1463,1493,1537-1545).
// ---------------------------------------------------------------------------------------------------------------------------------
struct FieldList {
    double *f[BZ_COMM_MAX_FIELDS];
    int32_t lev[BZ_COMM_MAX_FIELDS];
    int n = 0;
    bool half = false;          // Float32 working fields of a Float64 model (see halo_exchange)
    bool overflow = false;      // a field beyond BZ_COMM_MAX_FIELDS must fail the exchange, not silently stay unexchanged
    void add(double *p, int32_t levels)
    {
        if (!p) return;
        if (n >= BZ_COMM_MAX_FIELDS) { overflow = true; return; }
        f[n] = p; lev[n] = levels; ++n;
    }
};

static int cmp_exchange(bz_ctx *ctx, const FieldList &L)
{
    if (L.overflow) {
        ctx->last_error = "halo exchange list exceeds BZ_COMM_MAX_FIELDS";
        return BZ_ERR_UNSUPPORTED;
    }
    if (!L.n) return BZ_OK;
    ProfileScope ps(ctx, "comm_halo_exchange");
    return halo_exchange(ctx, L.f, L.lev, L.n, ctx->dg.Hy, true, true, ctx->stream, L.half);
}

// update_state! with the neighbour exchanges: rho_d first (face velocities divide by the dry density of row -1), then everything the
// next stage's stencils read across the slab edge
static int cmp_update_state(bz_ctx *ctx, const bz_compressible_state *s, const bz_compressible_prognostic *G,
                            const bz_acoustic_substepper *sub, bool tendencies)
{
    const DevGrid &g = ctx->dg;
    const int32_t nc = g.Nz + 2 * g.Hz, nf = nc + 1;
    int rc;
    FieldList A;
    A.add(s->rho_d, nc);
    if ((rc = cmp_exchange(ctx, A))) return rc;
    if ((rc = bz_compressible_update_state(ctx, s, G, sub, 0))) return rc;
    FieldList B;
    B.add(s->rho_d, nc); B.add(s->rho_u, nc); B.add(s->rho_v, nc); B.add(s->rho_w, nf); B.add(s->rho_theta, nc); B.add(s->rho_q, nc);
    B.add(sub->time_averaged_u, nc); B.add(sub->time_averaged_v, nc); B.add(sub->time_averaged_w, nf);
    B.add(s->rho, nc); B.add(s->p, nc); B.add(s->u, nc); B.add(s->v, nc); B.add(s->w, nf); B.add(s->theta, nc); B.add(s->q, nc); B.add(s->T, nc);
    if (g.microphysics == 2) {
        const bz_kessler_model_fields &K = ctx->kessler;
        B.add(K.cloud_liquid_density, nc); B.add(K.rain_density, nc);
        B.add(K.vapor_mass_fraction, nc); B.add(K.cloud_liquid_mass_fraction, nc); B.add(K.rain_mass_fraction, nc);
    }
    if (g.microphysics == 1) { B.add(g.qv_field, nc); B.add(g.ql_field, nc); }      // the halo rows' linearisation reads the liquid fraction
    if ((rc = cmp_exchange(ctx, B))) return rc;
    return tendencies ? bz_compute_moisture_tendency(ctx, s, G, sub) : BZ_OK;
}

extern "C" int bz_comm_compressible_update_state(bz_ctx *ctx, const bz_compressible_state *s, const bz_compressible_prognostic *G,
                                                 const bz_acoustic_substepper *sub, int compute_tendencies)
{
    if (!ctx || !ctx->comm || !ctx->compressible || !s || !G || !sub) return BZ_ERR_INVALID;
    return cmp_update_state(ctx, s, G, sub, compute_tendencies != 0);
}

int bzi_dist_time_step_compressible(bz_ctx *ctx, const bz_compressible_state *s, const bz_compressible_prognostic *U0,
                                    const bz_compressible_prognostic *G, const bz_acoustic_substepper *sub, double dt)
{
    const DevGrid &g = ctx->dg;
    const int32_t nc = g.Nz + 2 * g.Hz;
    int rc = bzi_compressible_store_initial_state(ctx, s, U0);
    if (rc) return rc;
    double *th_buf[2] = {sub->density_potential_temperature_perturbation, sub->previous_density_potential_temperature_perturbation};
    double *v_buf[2] = {sub->momentum_perturbation_v, ctx->vp2_user ? ctx->vp2_user : ctx->d_vp2};
    double *u_buf[2] = {sub->momentum_perturbation_u, ctx->up2_user ? ctx->up2_user : ctx->d_up2};
    const double betas[3] = {1.0 / 3.0, 1.0 / 2.0, 1.0};
    for (int st = 0; st < 3; ++st) {
        if ((rc = bz_refresh_linearization(ctx, s, sub))) return rc;               // prepare_acoustic_cache!
        if ((rc = bz_compute_slow_tendencies(ctx, s, G))) return rc;
        FieldList Gv;
        Gv.add(G->rho_v, nc);
        if ((rc = cmp_exchange(ctx, Gv))) return rc;
        int32_t ntau = 0, cur = 0;
        if ((rc = bz_acoustic_stage_begin(ctx, s, U0, G, sub, dt, betas[st], &ntau, &cur))) return rc;
        for (int32_t k = 1; k <= ntau; ++k) {
            FieldList P;
            P.half = ctx->substep_f32;      // substep_floattype = Float32: the perturbation fields are Float32 arrays (round 5)
            P.add(th_buf[cur], nc); P.add(v_buf[cur], nc);
            if ((rc = cmp_exchange(ctx, P))) return rc;
            if ((rc = bz_acoustic_substep(ctx, s, U0, G, sub, k, &cur))) return rc;
            if (ctx->se.direct_divergence_damping && ctx->se.damping_coefficient >= 0.0) {      // apply_divergence_damping!(::DirectDivergenceDamping)
                FieldList D;
                D.half = ctx->substep_f32;
                D.add(u_buf[cur], nc); D.add(v_buf[cur], nc);
                if ((rc = cmp_exchange(ctx, D))) return rc;
                if ((rc = bz_acoustic_direct_damping(ctx, s, U0, G, sub))) return rc;
            }
        }
        FieldList T;
        T.half = ctx->substep_f32;
        T.add(th_buf[cur], nc);
        if ((rc = cmp_exchange(ctx, T))) return rc;
        if ((rc = bz_acoustic_stage_end(ctx, s, U0, G, sub, dt, betas[st], 1))) return rc;
        if ((rc = cmp_update_state(ctx, s, G, sub, true))) return rc;
    }
    if (g.microphysics == 2) {      // microphysics_model_update!: rank-local columns, then the exchanging update_state!
        if ((rc = bz_compressible_kessler_update(ctx, s, G, sub, dt))) return rc;
        if ((rc = cmp_update_state(ctx, s, G, sub, true))) return rc;
    }
    return BZ_OK;
}
