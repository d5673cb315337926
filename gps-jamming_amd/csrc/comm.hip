// gj_comm_*: the one real exchange of the path (TDOA slots and per-stream result vectors to the
// solving rank) as RCCL collectives over xGMI, one communicator rank per GPU / process, for hosts
// that do not bring torch.distributed.  The reference has no equivalent (single process, numpy).
//
// RCCL is bound at run time (dlopen): the hot-path library keeps no link dependency on the
// 570-MB librccl, and inside a PyTorch process the copy torch already loaded is reused instead
// of mapping a second one.  Search order: $GPSJAM_RCCL, an already-loaded librccl, librccl.so.1,
// /opt/rocm/lib/librccl.so.1.
#include <dlfcn.h>
#include <unistd.h>

#include <atomic>

#include <rccl/rccl.h>   // types and prototypes only; nothing is linked

#include "gj_common.h"
#include "host_io.h"

struct gj_comm {
    gj_ctx* ctx = nullptr;            // nullptr: detached (its context is gone, or gj_comm_destroy has begun)
    ncclComm_t comm = nullptr;
    int rank = 0, n_ranks = 1;
    std::atomic<int> in_flight{0};    // collectives between comm_call_begin and the return of their RCCL call
};

namespace gj {

struct Rccl {
    void* handle = nullptr;
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclGather) Gather = nullptr;
    decltype(&ncclBroadcast) Broadcast = nullptr;
    decltype(&ncclAllGather) AllGather = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    decltype(&ncclCommCount) CommCount = nullptr;
    decltype(&ncclCommUserRank) CommUserRank = nullptr;
    decltype(&ncclCommCuDevice) CommCuDevice = nullptr;
    char why[256] = {0};
};

static Rccl* rccl() {
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] {
        const char* env = getenv("GPSJAM_RCCL");
        const char* tries[] = {env, "librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so.1"};
        for (int pass = 0; pass < 2 && !r.handle; ++pass)       // pass 0: only what the process already holds
            for (const char* name : tries) {
                if (!name || !*name) continue;
                r.handle = dlopen(name, RTLD_NOW | RTLD_LOCAL | (pass == 0 ? RTLD_NOLOAD : 0));
                if (r.handle) break;
            }
        if (!r.handle) {
            snprintf(r.why, sizeof(r.why), "cannot load librccl (%s)", dlerror());
            return;
        }
#define GJ_SYM(field, name)                                                                  \
    r.field = reinterpret_cast<decltype(r.field)>(dlsym(r.handle, name));                    \
    if (!r.field && !r.why[0]) snprintf(r.why, sizeof(r.why), "librccl has no symbol %s", name)
        GJ_SYM(GetUniqueId, "ncclGetUniqueId");
        GJ_SYM(CommInitRank, "ncclCommInitRank");
        GJ_SYM(CommDestroy, "ncclCommDestroy");
        GJ_SYM(Gather, "ncclGather");   // RCCL extension (rccl.h:745)
        GJ_SYM(Broadcast, "ncclBroadcast");
        GJ_SYM(AllGather, "ncclAllGather");
        GJ_SYM(GetErrorString, "ncclGetErrorString");
        GJ_SYM(CommCount, "ncclCommCount");
        GJ_SYM(CommUserRank, "ncclCommUserRank");
        GJ_SYM(CommCuDevice, "ncclCommCuDevice");
#undef GJ_SYM
    });
    return &r;
}

static int rccl_fail(gj_ctx* ctx, const char* what, ncclResult_t rc) {
    Rccl* r = rccl();
    return fail(ctx, GJ_ERR_HIP, "%s failed: %s", what, r->GetErrorString ? r->GetErrorString(rc) : "?");
}

// The RCCL calls are made WITHOUT the context lock (they may wait for a late rank), so the lock no longer keeps a
// communicator alive under a collective that another thread is still enqueueing.  What does: `in_flight`
// counts the calls between comm_call_begin and the return of their RCCL call; attaching / detaching a communicator
// and starting a call are serialised by one process-wide mutex (held for a few loads and stores, never across RCCL);
// whoever takes a communicator down -- gj_comm_destroy, or gj_destroy of its context -- first detaches it (no new
// call can start), then waits with no lock held until the calls in flight have returned, then destroys it.
static std::mutex& comm_mu() {
    static std::mutex m;
    return m;
}
static void comm_quiesce(gj_ctx* ctx, gj_comm* c) {
    while (c->in_flight.load(std::memory_order_acquire) > 0) {
        if (ctx) wait_hook(ctx, kWaitStream);
        usleep(100);
    }
}

// gj_destroy: communicators made on the context go down with it; their handles stay valid for gj_comm_destroy
// (which then only frees the handle), so the order in which a host drops the two does not matter.
void comm_detach_all(gj_ctx* ctx) {
    std::vector<gj_comm*> mine;
    {
        std::lock_guard<std::mutex> l(comm_mu());
        mine.swap(ctx->comms);
        for (gj_comm* c : mine) c->ctx = nullptr;
    }
    for (gj_comm* c : mine) comm_quiesce(ctx, c);
    // The calls in flight have returned -- but a collective that another thread enqueued a moment ago may still be
    // QUEUED on the context's stream (gj_destroy synchronised before it came here; the hammering thread of
    // tests/hip_stub kept going until the detach above).  RCCL's handle must outlive its queued work: drain the
    // streams once more, with no lock held, before the communicators go (round 6, found under ASan with the stand-in).
    if (!mine.empty()) {
        if (ctx->own_stream) (void)hipStreamSynchronize(ctx->own_stream);
        if (ctx->stream != ctx->own_stream) (void)hipStreamSynchronize(ctx->stream);
    }
    for (gj_comm* c : mine) {
        ncclComm_t h = nullptr;
        {
            std::lock_guard<std::mutex> l(comm_mu());
            h = c->comm;
            c->comm = nullptr;
        }
        if (h) (void)rccl()->CommDestroy(h);
    }
}

}   // namespace gj

using namespace gj;

extern "C" {

int gj_comm_unique_id(void* id) {
    if (!id) return GJ_ERR_INVALID;
    Rccl* r = rccl();
    if (!r->handle || r->why[0]) return GJ_ERR_UNSUPPORTED;
    static_assert(sizeof(ncclUniqueId) == GJ_COMM_ID_BYTES, "gpsjam.h promises 128 bytes");
    ncclUniqueId u;
    if (r->GetUniqueId(&u) != ncclSuccess) return GJ_ERR_HIP;
    memcpy(id, &u, sizeof(u));
    return GJ_OK;
}

int gj_comm_init_rank(gj_ctx* ctx, const void* id, int rank, int n_ranks, gj_comm** out) {
    if (!ctx || !id || !out) return GJ_ERR_INVALID;
    *out = nullptr;
    if (n_ranks < 1 || rank < 0 || rank >= n_ranks) return fail(ctx, GJ_ERR_INVALID, "rank %d of %d", rank, n_ranks);
    Rccl* r = rccl();
    if (!r->handle || r->why[0]) return fail(ctx, GJ_ERR_UNSUPPORTED, "%s", r->why);
    ncclUniqueId u;
    memcpy(&u, id, sizeof(u));
    gj_comm* c = new (std::nothrow) gj_comm();
    if (!c) return GJ_ERR_NOMEM;
    c->ctx = ctx;
    c->rank = rank;
    c->n_ranks = n_ranks;
    // the rendezvous blocks until every rank has arrived: outside the context lock
    (void)hipSetDevice(ctx->device);
    const ncclResult_t rc = r->CommInitRank(&c->comm, n_ranks, u, rank);   // binds to the current device = ctx's
    if (rc != ncclSuccess) {
        delete c;
        return rccl_fail(ctx, "ncclCommInitRank", rc);
    }
    {
        std::lock_guard<std::mutex> l(comm_mu());
        ctx->comms.push_back(c);
    }
    *out = c;
    return GJ_OK;
}

int gj_comm_rank(gj_comm* c, int* rank, int* n_ranks) {
    if (!c) return GJ_ERR_INVALID;
    // read from the LIVE communicator, not from what gj_comm_init_rank was told: a caller that reports "N ranks"
    // (bench.py's rccl_ranks) reports what RCCL itself holds.  A communicator whose context is gone answers 0 of 0.
    int r = -1, n = 0;
    std::lock_guard<std::mutex> l(comm_mu());   // two local queries: short, never blocks on a peer
    if (c->comm) {
        Rccl* lib = rccl();
        if (lib->CommUserRank(c->comm, &r) != ncclSuccess || lib->CommCount(c->comm, &n) != ncclSuccess) return GJ_ERR_HIP;
    }
    if (rank) *rank = r;
    if (n_ranks) *n_ranks = n;
    return GJ_OK;
}

int gj_comm_device(gj_comm* c, int* hip_device) {
    if (!c || !hip_device) return GJ_ERR_INVALID;
    *hip_device = -1;
    std::lock_guard<std::mutex> l(comm_mu());
    if (c->comm && rccl()->CommCuDevice(c->comm, hip_device) != ncclSuccess) return GJ_ERR_HIP;
    return GJ_OK;
}

// The collectives: arguments are checked and the stream is read under the context lock; the RCCL call itself is made
// WITHOUT it.  A communicator's first collective sets up its peer connections inside the call and can block until
// every peer has arrived -- a host-side wait like any other, and include/gpsjam.h promises that the lock is never
// held across one (a second thread scanning on the same context must not stall behind a late rank).  The order of
// collectives on one communicator is the caller's business, as it is with RCCL itself.
struct CommCall {
    gj_comm* c = nullptr;
    gj_ctx* ctx = nullptr;
    ncclComm_t comm = nullptr;
    hipStream_t stream = nullptr;
    int rank = 0, n_ranks = 1;
    CommCall() = default;
    CommCall(const CommCall&) = delete;
    CommCall& operator=(const CommCall&) = delete;
    ~CommCall() {
        if (c) c->in_flight.fetch_sub(1, std::memory_order_release);   // after the RCCL call has returned
    }
};

static int comm_call_begin(gj_comm* c, CommCall& k) {
    if (!c) return GJ_ERR_INVALID;
    {
        std::lock_guard<std::mutex> l(comm_mu());
        if (!c->ctx || !c->comm) return GJ_ERR_INVALID;   // destroyed, being destroyed, or its context is gone
        k.ctx = c->ctx;
        k.comm = c->comm;
        k.rank = c->rank;
        k.n_ranks = c->n_ranks;
        c->in_flight.fetch_add(1, std::memory_order_acquire);   // from here on neither the communicator nor its context goes away
        k.c = c;
    }
    Guard g(k.ctx);
    k.stream = k.ctx->stream;
    return GJ_OK;
}

int gj_comm_gather_dev(gj_comm* c, const void* d_send, size_t bytes, void* d_recv, int root) {
    CommCall k;
    if (int rc = comm_call_begin(c, k)) return rc;
    if (root < 0 || root >= k.n_ranks) return fail(k.ctx, GJ_ERR_INVALID, "root %d of %d", root, k.n_ranks);
    if (!d_send || (k.rank == root && !d_recv)) return fail(k.ctx, GJ_ERR_INVALID, "null buffer");
    NoCancel nc;
    (void)hipSetDevice(k.ctx->device);
    const ncclResult_t rc = rccl()->Gather(d_send, d_recv, bytes, ncclUint8, root, k.comm, k.stream);
    if (rc != ncclSuccess) return rccl_fail(k.ctx, "ncclGather", rc);
    return GJ_OK;
}

int gj_comm_allgather_dev(gj_comm* c, const void* d_send, size_t bytes, void* d_recv) {
    CommCall k;
    if (int rc = comm_call_begin(c, k)) return rc;
    if (!d_send || !d_recv) return fail(k.ctx, GJ_ERR_INVALID, "null buffer");
    NoCancel nc;
    (void)hipSetDevice(k.ctx->device);
    const ncclResult_t rc = rccl()->AllGather(d_send, d_recv, bytes, ncclUint8, k.comm, k.stream);
    if (rc != ncclSuccess) return rccl_fail(k.ctx, "ncclAllGather", rc);
    return GJ_OK;
}

int gj_comm_bcast_dev(gj_comm* c, void* d_buf, size_t bytes, int root) {
    CommCall k;
    if (int rc = comm_call_begin(c, k)) return rc;
    if (root < 0 || root >= k.n_ranks) return fail(k.ctx, GJ_ERR_INVALID, "root %d of %d", root, k.n_ranks);
    if (!d_buf) return fail(k.ctx, GJ_ERR_INVALID, "null buffer");
    NoCancel nc;
    (void)hipSetDevice(k.ctx->device);
    const ncclResult_t rc = rccl()->Broadcast(d_buf, d_buf, bytes, ncclUint8, root, k.comm, k.stream);
    if (rc != ncclSuccess) return rccl_fail(k.ctx, "ncclBroadcast", rc);
    return GJ_OK;
}

int gj_comm_destroy(gj_comm* c) {
    if (!c) return GJ_OK;
    gj_ctx* ctx = nullptr;
    {
        std::lock_guard<std::mutex> l(comm_mu());
        ctx = c->ctx;
        c->ctx = nullptr;                 // detached: no new call starts on it
        if (ctx)
            for (size_t k = 0; k < ctx->comms.size(); ++k)
                if (ctx->comms[k] == c) {
                    ctx->comms.erase(ctx->comms.begin() + (long)k);
                    break;
                }
    }
    if (ctx) {   // was still attached: calls in flight return first, collectives queued on the context's stream finish
        comm_quiesce(ctx, c);
        (void)wait_stream(ctx, current_stream(ctx));
        (void)hipSetDevice(ctx->device);
        if (c->comm) (void)rccl()->CommDestroy(c->comm);
    }
    delete c;
    return GJ_OK;
}

}   // extern "C"
