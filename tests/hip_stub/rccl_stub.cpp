// A stand-in for librccl, loaded by the library through $GPSJAM_RCCL exactly like the real one (dlopen + dlsym:
// gps-jamming_amd/csrc/comm.hip) -- TEST INFRASTRUCTURE ONLY, for running the host side of gj_comm_* under the
// sanitizers without a GPU.  Ranks are contexts of ONE process (one thread each); a collective is queued on the
// caller's stream (hip_stub_enqueue) and, when the stream reaches it, meets the other ranks at a barrier, copies
// with memcpy and leaves through a second barrier -- i.e. it completes asynchronously and blocks its stream until
// every rank has arrived, like the real thing.
#include <rccl/rccl.h>

#include <condition_variable>
#include <cstring>
#include <map>
#include <memory>
#include <mutex>
#include <vector>

#include "hip_stub.h"

namespace {

struct Group {
    std::mutex m;
    std::condition_variable cv;
    int n = 0, arrived_init = 0;
    // one barrier generation per collective step
    int waiting = 0;
    unsigned long long gen = 0;
    std::vector<const void*> send;
    std::vector<void*> recv;
    void barrier() {
        std::unique_lock<std::mutex> l(m);
        const unsigned long long g = gen;
        if (++waiting == n) {
            waiting = 0;
            ++gen;
            cv.notify_all();
        } else {
            cv.wait(l, [&] { return gen != g; });
        }
    }
};

struct Registry {
    std::mutex m;
    std::map<unsigned long long, std::shared_ptr<Group>> groups;
    unsigned long long next_id = 1;
};
Registry& reg() {
    static Registry r;
    return r;
}

}   // namespace

struct ncclComm {
    std::shared_ptr<Group> g;
    int rank = 0, n = 1, device = 0;
    bool dead = false;
};

namespace {

enum Kind { kAllGather, kGather, kBroadcast };
struct Op {
    ncclComm* c;
    Kind kind;
    const void* send;
    void* recv;
    size_t bytes;
    int root;
};

void run_op(void* p) {
    std::unique_ptr<Op> op(static_cast<Op*>(p));
    Group& g = *op->c->g;
    const int r = op->c->rank, n = op->c->n;
    {
        std::lock_guard<std::mutex> l(g.m);
        g.send[(size_t)r] = op->send;
        g.recv[(size_t)r] = op->recv;
    }
    g.barrier();   // every rank's pointers are in
    if (op->kind == kAllGather || (op->kind == kGather && r == op->root)) {
        for (int k = 0; k < n; ++k) {
            const void* src;
            {
                std::lock_guard<std::mutex> l(g.m);
                src = g.send[(size_t)k];
            }
            if (op->bytes) memcpy(static_cast<char*>(op->recv) + (size_t)k * op->bytes, src, op->bytes);
        }
    } else if (op->kind == kBroadcast && r != op->root) {
        const void* src;
        {
            std::lock_guard<std::mutex> l(g.m);
            src = g.send[(size_t)op->root];
        }
        if (op->bytes) memcpy(op->recv, src, op->bytes);
    }
    g.barrier();   // nobody's send buffer is reused before everyone has read it
}

// test control: while the gate is closed every collective call blocks at entry (inside "RCCL", where the library holds
// no lock) -- how a test puts a call in flight and keeps it there
std::mutex gate_m;
std::condition_variable gate_cv;
bool gate_closed = false;
int gate_waiting = 0;

ncclResult_t queue(ncclComm* c, Kind kind, const void* send, void* recv, size_t bytes, int root, hipStream_t s) {
    if (!c) return ncclInvalidArgument;
    {
        std::unique_lock<std::mutex> l(gate_m);
        ++gate_waiting;
        gate_cv.wait(l, [] { return !gate_closed; });
        --gate_waiting;
    }
    if (c->dead) abort();   // the library let a call through on a destroyed communicator
    Op* op = new Op{c, kind, send, recv, bytes, root};
    if (hip_stub_enqueue(s, run_op, op) != 0) {
        delete op;
        return ncclInvalidArgument;
    }
    return ncclSuccess;
}

size_t type_bytes(ncclDataType_t t) {
    switch (t) {
        case ncclInt8: case ncclUint8: return 1;
        case ncclFloat16: case ncclBfloat16: return 2;
        case ncclInt32: case ncclUint32: case ncclFloat32: return 4;
        default: return 8;
    }
}

}   // namespace

extern "C" {

void rccl_stub_gate(int closed) {
    {
        std::lock_guard<std::mutex> l(gate_m);
        gate_closed = closed != 0;
    }
    gate_cv.notify_all();
}
int rccl_stub_waiting(void) {
    std::lock_guard<std::mutex> l(gate_m);
    return gate_closed ? gate_waiting : 0;
}

ncclResult_t ncclGetUniqueId(ncclUniqueId* id) {
    if (!id) return ncclInvalidArgument;
    memset(id, 0, sizeof(*id));
    std::lock_guard<std::mutex> l(reg().m);
    const unsigned long long v = reg().next_id++;
    memcpy(id->internal, &v, sizeof(v));
    return ncclSuccess;
}

ncclResult_t ncclCommInitRank(ncclComm_t* out, int n, ncclUniqueId id, int rank) {
    if (!out || n < 1 || rank < 0 || rank >= n) return ncclInvalidArgument;
    unsigned long long key;
    memcpy(&key, id.internal, sizeof(key));
    std::shared_ptr<Group> g;
    {
        std::lock_guard<std::mutex> l(reg().m);
        auto& slot = reg().groups[key];
        if (!slot) {
            slot = std::make_shared<Group>();
            slot->n = n;
            slot->send.assign((size_t)n, nullptr);
            slot->recv.assign((size_t)n, nullptr);
        }
        g = slot;
    }
    if (g->n != n) return ncclInvalidArgument;
    ncclComm* c = new ncclComm();
    c->g = g;
    c->rank = rank;
    c->n = n;
    c->device = hip_stub_current_device();
    g->barrier();   // the rendezvous: returns when every rank has arrived
    *out = c;
    return ncclSuccess;
}

ncclResult_t ncclCommDestroy(ncclComm_t c) {
    delete c;
    return ncclSuccess;
}

ncclResult_t ncclAllGather(const void* send, void* recv, size_t count, ncclDataType_t t, ncclComm_t c, hipStream_t s) {
    return queue(c, kAllGather, send, recv, count * type_bytes(t), 0, s);
}
ncclResult_t ncclGather(const void* send, void* recv, size_t count, ncclDataType_t t, int root, ncclComm_t c, hipStream_t s) {
    return queue(c, kGather, send, recv, count * type_bytes(t), root, s);
}
ncclResult_t ncclBroadcast(const void* send, void* recv, size_t count, ncclDataType_t t, int root, ncclComm_t c, hipStream_t s) {
    return queue(c, kBroadcast, send, recv, count * type_bytes(t), root, s);
}
const char* ncclGetErrorString(ncclResult_t r) { return r == ncclSuccess ? "no error" : "error (rccl stub)"; }
ncclResult_t ncclCommCount(const ncclComm_t c, int* n) {
    if (!c || !n) return ncclInvalidArgument;
    *n = c->n;
    return ncclSuccess;
}
ncclResult_t ncclCommUserRank(const ncclComm_t c, int* r) {
    if (!c || !r) return ncclInvalidArgument;
    *r = c->rank;
    return ncclSuccess;
}
ncclResult_t ncclCommCuDevice(const ncclComm_t c, int* d) {
    if (!c || !d) return ncclInvalidArgument;
    *d = c->device;
    return ncclSuccess;
}

}   // extern "C"
