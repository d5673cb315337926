// Host-buffer side of the C-ABI (include/gpsjam.h): resident-capture uploads and the "*_u8" entry points that take
// numpy arrays.  Every call works in a LANE of its own (device staging, pinned bounce buffers, pinned result area,
// events: gj_common.h) and follows one shape:
//     check a lane out (lock, short) -> stage the input (no lock) -> enqueue kernels + result copy (lock, short)
//     -> wait for the call's own event (no lock) -> copy the results out of the lane's pinned area -> check in.
// Nothing waits, reads a file or copies a capture while holding ctx->mu, so a caller killed at any of those points
// (QThread.terminate(), GpsJammerApp/app/ui_mainwindow.py:818-826) blocks nobody; the lane it held is taken back
// at the next check-out after its thread has ended (gj_lane::owner, a robust mutex: gj_common.h).
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <sys/syscall.h>
#include <unistd.h>

#include <sched.h>

#include <atomic>
#include <chrono>
#include <cmath>
#include <new>
#include <string>
#include <thread>
#include <vector>

#include "gj_common.h"
#include "host_io.h"

namespace gj {

constexpr int kMaxLanes = 8;

static int this_tid() { return (int)syscall(SYS_gettid); }

// Everything a lane holds except the lane itself and its owner mutex.  hipFree / hipHostFree wait for the device.
static void lane_strip(gj_lane* L) {
    if (L->stage) (void)hipFree(L->stage);
    L->stage = nullptr;
    L->stage_bytes = 0;
    for (int k = 0; k < gj_lane::kPinBufs; ++k) {
        if (L->pin[k]) (void)hipHostFree(L->pin[k]);
        if (L->pin_ev[k]) (void)hipEventDestroy(L->pin_ev[k]);
        L->pin[k] = nullptr;
        L->pin_cap[k] = 0;
        L->pin_ev[k] = nullptr;
    }
    if (L->rpin) (void)hipHostFree(L->rpin);
    L->rpin = nullptr;
    L->rpin_bytes = 0;
    for (hipEvent_t* e : {&L->ev_start, &L->ev_stop, &L->ev_done}) {
        if (*e) (void)hipEventDestroy(*e);
        *e = nullptr;
    }
    for (size_t k = 0; k < L->n_piece_ev; ++k)
        if (L->piece_ev[k]) (void)hipEventDestroy(L->piece_ev[k]);
    delete[] L->piece_ev;
    L->piece_ev = nullptr;
    L->n_piece_ev = 0;
    if (L->copy_stream) (void)hipStreamDestroy(L->copy_stream);
    L->copy_stream = nullptr;
    if (L->ws) (void)hipFree(L->ws);
    L->ws = nullptr;
    L->ws_bytes = 0;
}

void lane_free(gj_lane* L) {
    if (!L) return;
    lane_strip(L);
    if (L->owner_ready) (void)pthread_mutex_destroy(&L->owner);
    delete L;
}

static bool lane_owner_init(gj_lane* L) {
    pthread_mutexattr_t at;
    if (pthread_mutexattr_init(&at) != 0) return false;
    (void)pthread_mutexattr_setrobust(&at, PTHREAD_MUTEX_ROBUST);
    L->owner_ready = pthread_mutex_init(&L->owner, &at) == 0;
    pthread_mutexattr_destroy(&at);
    return L->owner_ready;
}

// Try to become the lane's owner.  kOwnerLive: a live thread holds it.  kOwnerFree: it was free, now ours.
// kOwnerDead: its owner ended without checking in (the kernel said so), now ours.
enum OwnerProbe { kOwnerLive, kOwnerFree, kOwnerDead };
static OwnerProbe lane_owner_take(gj_lane* L) {
    const int r = pthread_mutex_trylock(&L->owner);
    if (r == 0) return kOwnerFree;
    if (r == EOWNERDEAD) {
        (void)pthread_mutex_consistent(&L->owner);
        return kOwnerDead;
    }
    if (r == ENOTRECOVERABLE) {   // left inconsistent by an earlier recovery that itself died: start over
        (void)pthread_mutex_destroy(&L->owner);
        L->owner_ready = false;
        if (lane_owner_init(L) && pthread_mutex_trylock(&L->owner) == 0) return kOwnerDead;
    }
    return kOwnerLive;
}

// Under the context lock: every busy lane whose owner has ended becomes the calling thread's (still marked busy) and
// is appended to `orphans`.
static void lane_sweep_locked(gj_ctx* ctx, int me, std::vector<gj_lane*>& orphans) {
    for (gj_lane* L : ctx->lanes) {
        if (!L->busy) continue;
        if (ctx->inject_owner_alive > 0) {   // gj_debug_inject: this probe answers "alive"
            --ctx->inject_owner_alive;
            continue;
        }
        if (lane_owner_take(L) == kOwnerLive) continue;
        // kOwnerDead, or "busy" with nobody holding it (its owner ended between taking the mutex and raising the flag)
        L->owner_tid = me;
        ++ctx->lanes_reclaimed;
        orphans.push_back(L);
    }
}

// With NO lock held, by the thread that now owns the orphan: what the dead caller queued may still be reading or
// writing the lane's buffers -- kernels on the context's stream(s), and, if it died inside gj_ingest_* / a staged
// upload, H2D copies on the lane's own copy stream or out of its bounce buffers.  Drain them, then give
// back everything the dead caller had grown the lane to (its staging arena and up to 32 pinned bounce buffers: hundreds
// of MiB that nobody asked the next caller to inherit).  Reclaims are rare; the next call re-creates what it needs.
// (Fill threads of a caller that was killed outright may still be writing a bounce buffer for a few milliseconds;
// INTEGRATION.md says that a hard kill DURING a staged copy is not covered.)
static void lane_recover(gj_ctx* ctx, gj_lane* L, hipStream_t s0, hipStream_t s1) {
    (void)hipSetDevice(ctx->device);
    (void)wait_stream(ctx, s0);
    if (s1 != s0) (void)wait_stream(ctx, s1);
    if (L->copy_stream) (void)wait_stream(ctx, L->copy_stream);
    for (int k = 0; k < gj_lane::kPinBufs; ++k)
        if (L->pin_ev[k]) (void)wait_event(ctx, L->pin_ev[k]);
    (void)hipGetLastError();
    lane_strip(L);
}

void lane_checkin(gj_ctx* ctx, gj_lane* L) {
    if (!L) return;
    Guard g(ctx);
    L->busy = false;
    L->owner_tid = 0;
    (void)pthread_mutex_unlock(&L->owner);   // by the thread that checked the lane out
}

// Take back the lanes of callers that have ended; returns how many.  Called by every check-out and by
// gj_debug_counters, so a dead caller's lane comes back at the next call whether or not a free lane exists.
int lane_sweep(gj_ctx* ctx) {
    std::vector<gj_lane*> orphans;
    hipStream_t s0 = nullptr, s1 = nullptr;
    {
        Guard g(ctx);
        lane_sweep_locked(ctx, this_tid(), orphans);
        s0 = ctx->stream;
        s1 = ctx->own_stream;
    }
    for (gj_lane* L : orphans) {
        lane_recover(ctx, L, s0, s1);
        lane_checkin(ctx, L);
    }
    return (int)orphans.size();
}

gj_lane* lane_checkout(gj_ctx* ctx) {
    reap_retired(ctx);
    const int me = this_tid();
    for (;;) {
        gj_lane* taken = nullptr;
        std::vector<gj_lane*> orphans;
        hipStream_t s0 = nullptr, s1 = nullptr;
        {
            Guard g(ctx);
            lane_sweep_locked(ctx, me, orphans);
            s0 = ctx->stream;
            s1 = ctx->own_stream;
            if (!orphans.empty()) {
                taken = orphans.front();   // already ours and marked busy
            } else {
                for (gj_lane* L : ctx->lanes)
                    if (!L->busy && lane_owner_take(L) != kOwnerLive) {
                        taken = L;
                        break;
                    }
                if (!taken && (int)ctx->lanes.size() < kMaxLanes) {
                    gj_lane* L = new (std::nothrow) gj_lane();
                    if (!L || !lane_owner_init(L) || pthread_mutex_trylock(&L->owner) != 0) {
                        lane_free(L);
                        fail(ctx, GJ_ERR_NOMEM, "lane");
                        return nullptr;
                    }
                    ctx->lanes.push_back(L);
                    taken = L;
                }
                if (taken) {
                    taken->busy = true;
                    taken->owner_tid = me;
                }
            }
        }
        for (gj_lane* L : orphans) {
            lane_recover(ctx, L, s0, s1);
            if (L != taken) lane_checkin(ctx, L);
        }
        if (taken) return taken;
        wait_hook(ctx, kWaitLane);   // every lane is held by a live caller: wait for one, holding nothing
        usleep(200);
    }
}

namespace {

struct LaneHold {
    NoCancel nc;   // from check-out to check-in the calling thread owns buffers, events and (while copying) helper threads
    gj_ctx* ctx;
    gj_lane* L;
    explicit LaneHold(gj_ctx* c) : ctx(c), L(lane_checkout(c)) { (void)hipSetDevice(c->device); }
    ~LaneHold() { lane_checkin(ctx, L); }
    LaneHold(const LaneHold&) = delete;
    LaneHold& operator=(const LaneHold&) = delete;
};

// The lane belongs to the calling thread: its buffers are grown without the lock.  hipFree waits for the device, so
// nothing queued earlier can still be using the old arena.
int lane_stage(gj_ctx* ctx, gj_lane* L, size_t bytes) {
    if (bytes <= L->stage_bytes) return GJ_OK;
    bytes = align_up(bytes, 1 << 20);
    if (L->stage) (void)hipFree(L->stage);
    L->stage = nullptr;
    L->stage_bytes = 0;
    void* p = nullptr;
    if (hipMalloc(&p, bytes) != hipSuccess) return fail(ctx, GJ_ERR_NOMEM, "staging buffer of %zu bytes", bytes);
    L->stage = static_cast<unsigned char*>(p);
    L->stage_bytes = bytes;
    return GJ_OK;
}

int lane_rpin(gj_ctx* ctx, gj_lane* L, size_t bytes) {
    if (bytes <= L->rpin_bytes) return GJ_OK;
    bytes = align_up(bytes, 1 << 16);
    if (L->rpin) (void)hipHostFree(L->rpin);
    L->rpin = nullptr;
    L->rpin_bytes = 0;
    void* p = nullptr;
    if (hipHostMalloc(&p, bytes, hipHostMallocDefault) != hipSuccess) return fail(ctx, GJ_ERR_NOMEM, "pinned result area of %zu bytes", bytes);
    L->rpin = static_cast<unsigned char*>(p);
    L->rpin_bytes = bytes;
    return GJ_OK;
}

int lane_events(gj_ctx* ctx, gj_lane* L) {
    if (!L->ev_start) GJ_HIP(ctx, hipEventCreate(&L->ev_start));
    if (!L->ev_stop) GJ_HIP(ctx, hipEventCreate(&L->ev_stop));
    if (!L->ev_done) GJ_HIP(ctx, hipEventCreateWithFlags(&L->ev_done, hipEventDisableTiming));
    return GJ_OK;
}

// Is `p` device memory (a resident capture from gj_upload* / gj_malloc)?  Host memory the runtime has never seen
// answers with an error or "unregistered", depending on the ROCm release.
bool is_device_ptr(const void* p) {
    if (!p) return false;
    hipPointerAttribute_t a;
    memset(&a, 0, sizeof(a));
    if (hipPointerGetAttributes(&a, p) != hipSuccess) {
        (void)hipGetLastError();
        return false;
    }
    return a.type == hipMemoryTypeDevice;
}

// Captures go through pinned bounce buffers, piece by piece: the host copy of one piece overlaps the DMA of another
// on every fill thread (29 GB/s with one fill thread against 21 GB/s for a pageable hipMemcpy of 1 GiB on the MI355X
// box, tools/h2d_bench.hip; 51 GB/s with eight).  Pieces are 16 MiB for the GiB-class captures of the benchmark and
// smaller for the sizes the reference is used at (a 10-s capture is 41 MB, a minute 246 MB: worker.py:184-196), so
// that those, too, keep every fill thread and the link busy; only a few MiB take one plain copy.
constexpr size_t kPinBytes = 16u << 20;       // largest piece
constexpr size_t kMinPiece = 1u << 20;        // smallest: whole 64-KiB scan tiles either way
constexpr size_t kPinThreshold = 4u << 20;    // below this: one copy, no helper threads
constexpr int kMaxFillThreads = gj_lane::kPinBufs / 2;

// fill threads of one staged copy: GPSJAM_FILL_THREADS (1..16) fixes the number; the default is by capture size, at most
// 8 (one memcpy thread tops out at ~31 GB/s end to end, below what the link carries; pread out of the page cache needs
// the extra threads more than memcpy).  By size since round 5: eight threads are right from ~80 MB up, but for the
// reference's 10-s captures (41 MB) starting and joining eight threads costs more than they carry -- three such files one
// after the other: 5.04 ms with eight threads each, 4.54 with four, 4.95 with three, 5.08 with two
// (tools/ingest3_probe.py, profiles/NOTES_r05.md section 8).
int fill_threads_env() {
    static const int n = [] {
        const char* e = getenv("GPSJAM_FILL_THREADS");
        int v = e ? atoi(e) : 0;
        if (v > kMaxFillThreads) v = kMaxFillThreads;
        return v < 0 ? 0 : v;
    }();
    return n;
}
// The shape of ONE staged copy -- fill threads, piece size, piece count -- decided once, at the start of the call, and
// handed down.  Until round 6 every user re-derived it from ctx->fill_threads, which gj_set_fill_threads (and
// gj_ingest_files, temporarily) could change in between: an ingest then sized its piece events for one piece count and
// copied with another (ThreadSanitizer, tests/hip_stub: heap-use-after-free in staged_copy).
//   `override_threads` > 0: this call's own setting (gj_ingest_files lowers it per file), else the context's, else by size.
// Piece size: about four pieces per fill thread, whole MiB, at most 16 MiB (reached from 512 MiB up with eight threads:
// the GiB-class figures of profiles/r0*_ingest*.txt were measured with it).
struct CopyShape {
    int nthreads = 1;
    size_t piece_len = kMinPiece, npieces = 0;
};
CopyShape copy_shape(const gj_ctx* ctx, size_t nbytes, int override_threads = 0) {
    int t = fill_threads_env();
    if (!t) t = override_threads > 0 ? override_threads : ctx->fill_threads.load(std::memory_order_relaxed);   // gj_set_fill_threads
    if (t <= 0) {
        size_t by_size = nbytes / (8u << 20);       // one thread per 8 MiB: four for a 10-s capture
        t = by_size < 2 ? 2 : (by_size > 8 ? 8 : (int)by_size);
    }
    if (t > kMaxFillThreads) t = kMaxFillThreads;
    CopyShape c;
    c.piece_len = align_up(nbytes / (4 * (size_t)t) + 1, kMinPiece);
    if (c.piece_len > kPinBytes) c.piece_len = kPinBytes;
    c.npieces = (nbytes + c.piece_len - 1) / c.piece_len;
    c.nthreads = (int)(c.npieces < (size_t)t ? (c.npieces ? c.npieces : 1) : (size_t)t);
    return c;
}

// the lane's bounce buffer k with room for `bytes` (grow-only; the lane's previous call has left its buffers)
int lane_pin(gj_ctx* ctx, gj_lane* L, int k, size_t bytes) {
    if (L->pin[k] && L->pin_cap[k] >= bytes) return GJ_OK;
    if (L->pin[k]) (void)hipHostFree(L->pin[k]);
    L->pin[k] = nullptr;
    L->pin_cap[k] = 0;
    GJ_HIP(ctx, hipHostMalloc(&L->pin[k], bytes, hipHostMallocDefault));
    L->pin_cap[k] = bytes;
    return GJ_OK;
}

// Host source -> d_dst through the lane's pinned buffers, on `stream`, with no lock held anywhere.
// `fill(dst, off, len)` puts bytes [off, off+len) of the source into a pinned buffer: memcpy from a numpy array or a
// mapped file, or pread from a capture file.  `landed(piece)`, when given, is called by the fill thread right after
// it has queued piece `piece` (piece_bytes(nbytes) each, in HBM once ev fires): the overlapped ingest hangs its kernels there.
struct PieceSink {
    virtual void queued(size_t piece, size_t off, size_t len, hipEvent_t ev) = 0;
    virtual void failed() {}
    virtual ~PieceSink() = default;
};

// With a sink the CALLING thread does not fill: it runs `meanwhile()` (the ingest's dispatcher, which launches kernels
// on the pieces as they land) while `nthreads` workers copy; `piece_events[k]` is recorded behind piece k.
struct NoMeanwhile {
    void operator()() const {}
};
template <typename Fill, typename Meanwhile = NoMeanwhile>
int staged_copy(gj_ctx* ctx, gj_lane* L, hipStream_t stream, unsigned char* d_dst, size_t nbytes, const CopyShape& shape, Fill&& fill,
                PieceSink* sink = nullptr, hipEvent_t* piece_events = nullptr, Meanwhile&& meanwhile = Meanwhile()) {
    if (nbytes == 0) return GJ_OK;
    const size_t piece_len = shape.piece_len, npieces = shape.npieces;
    const int nthreads = shape.nthreads;
    for (int k = 0; k < 2 * nthreads; ++k) {   // two bounce buffers per fill thread, made (or grown) on first use
        const int prc = lane_pin(ctx, L, k, piece_len);
        if (prc) return prc;
        if (!L->pin_ev[k]) GJ_HIP(ctx, hipEventCreateWithFlags(&L->pin_ev[k], hipEventDisableTiming));
    }
    wait_hook(ctx, kWaitPiece);
    std::atomic<int> failed{0};
    const int device = ctx->device;
    auto worker_body = [&](int t) {
        if (hipSetDevice(device) != hipSuccess) { failed.store(1); return; }
        size_t mine = 0;
        for (size_t piece = (size_t)t; piece < npieces; piece += (size_t)nthreads, ++mine) {
            const size_t off = piece * piece_len;
            const size_t len = (nbytes - off < piece_len) ? nbytes - off : piece_len;
            const int b = 2 * t + (int)(mine & 1);
            if (mine >= 2 && hipEventSynchronize(L->pin_ev[b]) != hipSuccess) { failed.store(1); return; }
            if (!fill(static_cast<unsigned char*>(L->pin[b]), off, len)) { failed.store(2); return; }
            if (hipMemcpyAsync(d_dst + off, L->pin[b], len, hipMemcpyHostToDevice, stream) != hipSuccess ||
                hipEventRecord(L->pin_ev[b], stream) != hipSuccess) { failed.store(1); return; }
            if (piece_events && hipEventRecord(piece_events[piece], stream) != hipSuccess) { failed.store(1); return; }
            if (sink) sink->queued(piece, off, len, piece_events ? piece_events[piece] : L->pin_ev[b]);
        }
        // the bounce buffers are reused by the lane's next call: the tail pieces must have left them
        for (int k = 0; k < 2; ++k)
            if (mine > (size_t)k && hipEventSynchronize(L->pin_ev[2 * t + k]) != hipSuccess) failed.store(1);
    };
    auto worker = [&](int t) {
        worker_body(t);
        if (failed.load() && sink) sink->failed();   // the dispatcher must not wait for pieces that will never come
    };
    std::vector<std::thread> pool;
    int started = sink ? 0 : 1;            // the calling thread is worker 0 unless it dispatches
    try {
        pool.reserve((size_t)nthreads);
        for (; started < nthreads; ++started) pool.emplace_back(worker, started);
    } catch (...) {                        // no more threads to be had (EAGAIN under a process limit): never out of an extern "C" call
        failed.store(1);
        if (sink) sink->failed();
    }
    if (sink) {
        if (failed.load()) sink->failed();
        meanwhile();
    } else if (!failed.load()) {
        worker(0);
    }
    for (auto& th : pool) th.join();
    if (failed.load() == 2) return fail(ctx, GJ_ERR_INVALID, "reading the capture failed");
    if (failed.load()) return fail(ctx, GJ_ERR_HIP, "host-to-device staging failed");
    return GJ_OK;
}

int copy_in(gj_ctx* ctx, gj_lane* L, hipStream_t s, unsigned char* d_dst, const uint8_t* host, size_t nbytes) {
    if (nbytes == 0) return GJ_OK;
    if (nbytes < kPinThreshold) {
        GJ_HIP(ctx, hipMemcpyAsync(d_dst, host, nbytes, hipMemcpyHostToDevice, s));
        return GJ_OK;
    }
    return staged_copy(ctx, L, s, d_dst, nbytes, copy_shape(ctx, nbytes), [host](unsigned char* dst, size_t off, size_t len) {
        memcpy(dst, host + off, len);
        return true;
    });
}

// One host-buffer call.  begin(): lane + input (a device pointer is used in place, a host buffer is staged) + a
// result region of `result_bytes` in the lane; run(): kernels under the lock, results queued into the lane's pinned
// area; finish(): wait for this call's event with nothing held, then hand out the results.
struct HostCall {
    gj_ctx* ctx;
    LaneHold hold;
    gj_lane* L;
    hipStream_t s = nullptr;            // stream the input was staged on
    const uint8_t* d_in = nullptr;
    unsigned char* d_res = nullptr;
    size_t result_bytes = 0;
    bool staged = false;

    explicit HostCall(gj_ctx* c) : ctx(c), hold(c), L(hold.L) {}

    int begin(const uint8_t* iq, size_t nbytes, size_t res_bytes) {
        if (!L) return GJ_ERR_NOMEM;
        int rc = lane_events(ctx, L);
        if (rc) return rc;
        result_bytes = res_bytes;
        const bool resident = nbytes && is_device_ptr(iq);
        const size_t in_bytes = resident ? 0 : align_up(nbytes, 256) + 256;
        rc = lane_stage(ctx, L, in_bytes + align_up(res_bytes, 256) + 256);
        if (!rc) rc = lane_rpin(ctx, L, res_bytes + 256);
        if (rc) return rc;
        d_res = L->stage + in_bytes;
        s = current_stream(ctx);
        if (resident) {
            d_in = iq;
        } else {
            d_in = L->stage;
            rc = copy_in(ctx, L, s, L->stage, iq, nbytes);
            if (rc) return rc;
            staged = nbytes != 0;
            if (staged) GJ_HIP(ctx, hipEventRecord(L->ev_done, s));   // "input staged", for a stream change in between
        }
        return GJ_OK;
    }

    // fn(): the launch_* calls, entered under the lock with the timing events around them
    template <typename Fn>
    int run(Fn&& fn) {
        Guard g(ctx);
        if (staged && ctx->stream != s) GJ_HIP(ctx, hipStreamWaitEvent(ctx->stream, L->ev_done, 0));
        s = ctx->stream;
        GJ_HIP(ctx, hipEventRecord(L->ev_start, s));
        const int rc = fn();
        if (rc) return rc;
        GJ_HIP(ctx, hipEventRecord(L->ev_stop, s));
        if (result_bytes) GJ_HIP(ctx, hipMemcpyAsync(L->rpin, d_res, result_bytes, hipMemcpyDeviceToHost, s));
        GJ_HIP(ctx, hipEventRecord(L->ev_done, s));
        return GJ_OK;
    }

    int finish(float* kernel_ms) {
        int rc = wait_event(ctx, L->ev_done);
        if (rc) return rc;
        if (kernel_ms) {
            float t = 0.f;
            GJ_HIP(ctx, hipEventElapsedTime(&t, L->ev_start, L->ev_stop));
            *kernel_ms = t;
        }
        return GJ_OK;
    }

    template <typename T>
    T* dev(size_t off = 0) const { return reinterpret_cast<T*>(d_res + off); }
    const unsigned char* host(size_t off = 0) const { return L->rpin + off; }
};

}   // namespace
}   // namespace gj

using namespace gj;

extern "C" {

// ---------------------------------------------------------------- resident captures
// One upload per capture, then any number of calls on it (gpsjam.Capture): the host-buffer entry points below stage
// a host input on every call (21 ms per GiB of PCIe against 0.2-1.3 ms of kernel time) and use a device pointer in
// place.
int gj_upload(gj_ctx* ctx, const uint8_t* host, size_t nbytes, void** dptr) {
    if (!ctx) return GJ_ERR_INVALID;
    if (!dptr || (nbytes && !host)) return fail(ctx, GJ_ERR_INVALID, "null buffer");
    *dptr = nullptr;
    LaneHold hold(ctx);
    gj_lane* L = hold.L;
    if (!L) return GJ_ERR_NOMEM;
    int rc = lane_events(ctx, L);
    if (rc) return rc;
    void* p = nullptr;
    if (hipMalloc(&p, align_up(nbytes, 256) + 256) != hipSuccess) return fail(ctx, GJ_ERR_NOMEM, "hipMalloc(%zu)", nbytes);
    const hipStream_t s = current_stream(ctx);
    rc = copy_in(ctx, L, s, static_cast<unsigned char*>(p), host, nbytes);
    if (!rc && hipEventRecord(L->ev_done, s) != hipSuccess) rc = fail(ctx, GJ_ERR_HIP, "upload failed");
    if (!rc) rc = wait_event(ctx, L->ev_done);
    if (rc) {
        (void)hipFree(p);
        return rc;
    }
    *dptr = p;
    return GJ_OK;
}

// The reference's ingest (np.fromfile / f.read, worker.py:209-217, triangulateRSSI.py:29) as file -> pinned bounce
// buffers -> HBM.  max_bytes = 0: to the end of the file.
int gj_upload_file(gj_ctx* ctx, const char* path, size_t offset, size_t max_bytes, void** dptr, size_t* nbytes_out) {
    if (!ctx) return GJ_ERR_INVALID;
    if (!path || !dptr || !nbytes_out) return fail(ctx, GJ_ERR_INVALID, "null argument");
    *dptr = nullptr;
    *nbytes_out = 0;
    LaneHold hold(ctx);
    gj_lane* L = hold.L;
    if (!L) return GJ_ERR_NOMEM;
    int rc = lane_events(ctx, L);
    if (rc) return rc;
    const int fd = open(path, O_RDONLY);
    if (fd < 0) return fail(ctx, GJ_ERR_INVALID, "cannot open %s", path);
    struct stat st;
    if (fstat(fd, &st) != 0) {
        close(fd);
        return fail(ctx, GJ_ERR_INVALID, "cannot stat %s", path);
    }
    size_t nbytes = (size_t)st.st_size > offset ? (size_t)st.st_size - offset : 0;
    if (max_bytes && nbytes > max_bytes) nbytes = max_bytes;
    void* p = nullptr;
    if (hipMalloc(&p, align_up(nbytes, 256) + 256) != hipSuccess) {
        close(fd);
        return fail(ctx, GJ_ERR_NOMEM, "hipMalloc(%zu)", nbytes);
    }
    const hipStream_t s = current_stream(ctx);
    // Two ways from the page cache into the pinned bounce buffers (tools/ingest_bench.py, tools/ingest_probe.cpp,
    // profiles/r02_ingest.txt; 1 GiB in /dev/shm):
    //  * mapping the file and copying in user space: 35-42 ms, of which 15-20 ms is the final munmap -- the same
    //    whether the file has been read before or not.  Handing the munmap to a helper thread makes the call return
    //    after 22 ms (49 GB/s) but the kernels and copies that follow then wait on the driver's MMU notifiers for
    //    longer than the munmap took (file -> results 54-61 ms instead of 40), so it stays in the call;
    //  * pread into the pinned buffers: 22-29 ms (37-48 GB/s) on a file that has been read before, but 75-100 ms on
    //    the FIRST read of a freshly written one: the second touch of a page moves it to the active list, and eight
    //    threads doing that fight over the LRU lock (the mapped path pays the same move inside munmap, from one
    //    thread, uncontended).
    // A capture is normally read once, soon after it was recorded, so mapping is the default; GPSJAM_FILE_READ=pread
    // selects the other, which is also what a file that cannot be mapped gets.
    static const bool want_pread = [] {
        const char* e = getenv("GPSJAM_FILE_READ");
        return e && strcmp(e, "pread") == 0;
    }();
    const size_t pg = (size_t)sysconf(_SC_PAGESIZE);
    const size_t map_off = offset / pg * pg, lead = offset - map_off;
    void* m = (nbytes && !want_pread) ? mmap(nullptr, nbytes + lead, PROT_READ, MAP_PRIVATE, fd, (off_t)map_off) : MAP_FAILED;
    if (nbytes == 0) {
        rc = GJ_OK;
    } else if (m != MAP_FAILED) {
        (void)madvise(m, nbytes + lead, MADV_SEQUENTIAL);
        const unsigned char* src = static_cast<const unsigned char*>(m) + lead;
        rc = staged_copy(ctx, L, s, static_cast<unsigned char*>(p), nbytes, copy_shape(ctx, nbytes), [src](unsigned char* dst, size_t off, size_t len) {
            memcpy(dst, src + off, len);
            return true;
        });
        (void)munmap(m, nbytes + lead);
    } else {
        (void)posix_fadvise(fd, 0, 0, POSIX_FADV_NOREUSE);   // regular file systems (Linux >= 6.3): no LRU promotion on read
        rc = staged_copy(ctx, L, s, static_cast<unsigned char*>(p), nbytes, copy_shape(ctx, nbytes), [fd, offset](unsigned char* dst, size_t off, size_t len) {
            size_t done = 0;
            while (done < len) {
                const ssize_t k = pread(fd, dst + done, len - done, (off_t)(offset + off + done));
                if (k <= 0) return false;
                done += (size_t)k;
            }
            return true;
        });
    }
    close(fd);
    if (!rc && hipEventRecord(L->ev_done, s) != hipSuccess) rc = fail(ctx, GJ_ERR_HIP, "upload failed");
    if (!rc) rc = wait_event(ctx, L->ev_done);
    if (rc) {
        (void)hipFree(p);
        return rc;
    }
    *dptr = p;
    *nbytes_out = nbytes;
    return GJ_OK;
}

}   // extern "C"

// ---------------------------------------------------------------- overlapped ingest
// Upload + analysis of one capture with the kernels running on the pieces that have LANDED while the rest is still on
// its way ("end-to-end is 17-35x the kernel time and nothing overlaps it").  The reference's ingest
// (worker.py:209-217 f.read per chunk, triangulateRSSI.py:29 np.fromfile) reads, then computes; here the 16-MiB pieces
// go file/array -> pinned bounce buffers -> HBM on a stream of the lane's own, each followed by an event, and the
// calling thread -- which does not copy -- walks the pieces in order: wait (on the host) until piece k has been QUEUED,
// make the context's stream wait for its event, launch the fused scan on its 256 tiles and K2 on the 1-s chunks that are
// now complete.  After the last piece: the tail kernels (amplitude totals, onset, threshold-free finalize, PSD sum).
// Same kernels, same per-tile / per-workgroup partial results, same fixed-order sums: bit-identical to upload-then-run.
namespace gj {
namespace {

struct IngestSink : PieceSink {
    std::atomic<unsigned char>* queued_flag;
    std::atomic<int> dead{0};
    void queued(size_t piece, size_t, size_t, hipEvent_t) override { queued_flag[piece].store(1, std::memory_order_release); }
    void failed() override { dead.store(1); }
};

int lane_ingest_resources(gj_ctx* ctx, gj_lane* L, size_t npieces, size_t ws_bytes) {
    if (!L->copy_stream) GJ_HIP(ctx, hipStreamCreateWithFlags(&L->copy_stream, hipStreamNonBlocking));
    if (L->n_piece_ev < npieces) {
        hipEvent_t* ev = new (std::nothrow) hipEvent_t[npieces]();
        if (!ev) return fail(ctx, GJ_ERR_NOMEM, "piece events");
        for (size_t k = 0; k < L->n_piece_ev; ++k) ev[k] = L->piece_ev[k];
        delete[] L->piece_ev;
        L->piece_ev = ev;
        for (size_t k = L->n_piece_ev; k < npieces; ++k) {
            L->n_piece_ev = k;
            GJ_HIP(ctx, hipEventCreateWithFlags(&L->piece_ev[k], hipEventDisableTiming));
        }
        L->n_piece_ev = npieces;
    }
    if (L->ws_bytes < ws_bytes) {
        if (L->ws) (void)hipFree(L->ws);
        L->ws = nullptr;
        L->ws_bytes = 0;
        void* p = nullptr;
        if (hipMalloc(&p, align_up(ws_bytes, 1 << 20)) != hipSuccess) return fail(ctx, GJ_ERR_NOMEM, "ingest workspace of %zu bytes", ws_bytes);
        L->ws = static_cast<unsigned char*>(p);
        L->ws_bytes = align_up(ws_bytes, 1 << 20);
    }
    return GJ_OK;
}

template <typename Fill>
int ingest_impl(gj_ctx* ctx, size_t nbytes, Fill&& fill, const gj_ingest_plan& plan, float* power, size_t power_cap, float* psd,
                float* psd_db, size_t psd_cap_floats, gj_ingest_result* res, void** dptr, int fill_override = 0) {
    if (!res || !dptr) return fail(ctx, GJ_ERR_INVALID, "null argument");
    *dptr = nullptr;
    memset(res, 0, sizeof(*res));
    res->nbytes = nbytes;
    const bool want_scan = plan.chunk_bytes != 0;
    const bool want_welch = plan.nperseg != 0;
    const size_t n_chunks = want_scan ? gj_chunk_count(nbytes, plan.chunk_bytes) : 0;
    const size_t rows = want_welch ? gj_welch_rows(nbytes, plan.chunk_samples, plan.nperseg) : 0;
    const size_t nfl = rows * (size_t)(want_welch ? plan.nperseg : 0);
    res->n_chunks = n_chunks;
    res->rows = rows;
    if (want_scan && n_chunks && (!power || power_cap < n_chunks)) return fail(ctx, GJ_ERR_CAPACITY, "power buffer holds %zu, need %zu", power_cap, n_chunks);
    if (want_welch && rows && (!psd || psd_cap_floats < nfl)) return fail(ctx, GJ_ERR_CAPACITY, "psd buffer holds %zu floats, need %zu", psd_cap_floats, nfl);
    if (want_welch && (plan.nperseg < 16 || plan.nperseg > 4096 || (plan.nperseg & (plan.nperseg - 1))))
        return fail(ctx, GJ_ERR_UNSUPPORTED, "nperseg must be a power of two in [16, 4096]");
    const auto t_begin = std::chrono::steady_clock::now();
    LaneHold hold(ctx);
    gj_lane* L = hold.L;
    if (!L) return GJ_ERR_NOMEM;
    int rc = lane_events(ctx, L);
    if (rc) return rc;
    void* p = nullptr;
    if (hipMalloc(&p, align_up(nbytes, 256) + 256) != hipSuccess) return fail(ctx, GJ_ERR_NOMEM, "hipMalloc(%zu)", nbytes);
    unsigned char* d_cap = static_cast<unsigned char*>(p);
    auto bail = [&](int code) {
        (void)hipDeviceSynchronize();   // nothing may still be writing into, or reading from, the capture
        (void)hipFree(p);
        return code;
    };
    // device results of this call: [power][amp][onset][psd][psd_db]
    const size_t off_amp = align_up(n_chunks * sizeof(float), 256);
    const size_t off_onset = off_amp + 256;
    const size_t off_psd = off_onset + 256;
    const size_t off_db = off_psd + align_up(nfl * sizeof(float), 256);
    const size_t res_bytes = off_db + (psd_db ? align_up(nfl * sizeof(float), 256) : 0);
    rc = lane_stage(ctx, L, res_bytes + 256);
    if (!rc) rc = lane_rpin(ctx, L, res_bytes + 256);
    if (rc) return bail(rc);
    float* d_power = reinterpret_cast<float*>(L->stage);
    gj_amp_stats* d_amp = reinterpret_cast<gj_amp_stats*>(L->stage + off_amp);
    gj_onset* d_onset = reinterpret_cast<gj_onset*>(L->stage + off_onset);
    float* d_psd = reinterpret_cast<float*>(L->stage + off_psd);
    float* d_db = psd_db ? reinterpret_cast<float*>(L->stage + off_db) : nullptr;

    // plans (no GPU work yet)
    ScanJob sj;
    WelchJob wj;
    const bool fused = want_scan && n_chunks && scan_fusable(d_cap, nbytes, plan.chunk_bytes);
    if (fused) {
        rc = scan_begin(ctx, d_cap, nbytes, plan.chunk_bytes, plan.eps, plan.power_flags, d_power, plan.rssi_threshold, d_amp,
                        plan.noise_samples, plan.window, plan.factor, d_onset, sj);
        if (rc) return bail(rc);
    }
    if (want_welch) {
        rc = welch_begin(ctx, nbytes, plan.chunk_samples, plan.nperseg, plan.fs, 0, wj);
        if (rc) return bail(rc);
    }
    const CopyShape shape = copy_shape(ctx, nbytes, fill_override);   // once: the piece events below and the copy must agree
    const size_t piece_len = shape.piece_len, npieces = shape.npieces;
    const size_t ws_scan = fused ? sj.ws_bytes : 0;
    rc = lane_ingest_resources(ctx, L, npieces ? npieces : 1, ws_scan + (want_welch ? wj.ws_bytes : 0) + 256);
    if (rc) return bail(rc);
    sj.ws = L->ws;
    wj.partial = reinterpret_cast<float*>(L->ws + ws_scan);
    const size_t psd_chunk_bytes = want_welch ? 2 * plan.chunk_samples : 0;

    hipStream_t s = nullptr;
    {
        Guard g(ctx);
        s = ctx->stream;
        if (hipEventRecord(L->ev_start, s) != hipSuccess) return bail(fail(ctx, GJ_ERR_HIP, "event"));
        if (fused) rc = scan_start(ctx, sj);
    }
    if (rc) return bail(rc);
    // Another thread may switch the context's stream while this call is between two lock sections (gj_set_stream:
    // AntennaStream / SplitStreams do at construction).  The scan / Welch helpers launch on ctx->stream as it is in THEIR
    // lock section, while the piece events were waited for on `s`: chain the new stream behind everything queued on the
    // old one and carry on there (HostCall::run handles the same case).  Called under the Guard.
    auto follow = [&]() -> int {
        if (ctx->stream == s) return GJ_OK;
        if (hipEventRecord(L->ev_done, s) != hipSuccess || hipStreamWaitEvent(ctx->stream, L->ev_done, 0) != hipSuccess)
            return fail(ctx, GJ_ERR_HIP, "following a stream switch failed");
        s = ctx->stream;
        return GJ_OK;
    };

    // kernels on everything that is complete once the first `landed` bytes are in HBM
    size_t tiles_done = 0, chunks_done = 0;
    auto launch_upto = [&](size_t landed, bool last) -> int {
        Guard g(ctx);
        int r = follow();
        if (r) return r;
        if (fused) {
            const size_t t1 = last ? sj.ntiles : landed / 65536;
            if (t1 > tiles_done) r = scan_range(ctx, sj, tiles_done, t1);
            if (t1 > tiles_done) tiles_done = t1;
        }
        if (!r && want_welch && rows) {
            const size_t c1 = last ? rows : landed / psd_chunk_bytes;
            if (c1 > chunks_done) r = welch_range(ctx, wj, d_cap, chunks_done, c1);
            if (c1 > chunks_done) chunks_done = c1 < rows ? c1 : rows;
        }
        return r;
    };

    double upload_ms = 0.0;
    if (nbytes >= kPinThreshold) {
        std::vector<std::atomic<unsigned char>> flags(npieces);
        for (auto& f : flags) f.store(0);
        IngestSink sink;
        sink.queued_flag = flags.data();
        int disp_rc = GJ_OK;
        rc = staged_copy(ctx, L, L->copy_stream, d_cap, nbytes, shape, fill, &sink, L->piece_ev, [&] {
            // dispatcher: the calling thread.  Holds no lock while it waits for a piece to be queued.
            for (size_t k = 0; k < npieces && !disp_rc; ++k) {
                unsigned spins = 0;
                while (!flags[k].load(std::memory_order_acquire)) {
                    if (sink.dead.load()) return;
                    if (++spins > 64) usleep(20); else sched_yield();
                }
                if (hipStreamWaitEvent(s, L->piece_ev[k], 0) != hipSuccess) { disp_rc = fail(ctx, GJ_ERR_HIP, "hipStreamWaitEvent failed"); return; }
                const size_t landed = (k + 1 == npieces) ? nbytes : (k + 1) * piece_len;
                disp_rc = launch_upto(landed, k + 1 == npieces);
            }
        });
        upload_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_begin).count();
        if (!rc) rc = disp_rc;
        if (rc) return bail(rc);
    } else {
        // a few MiB: one piece through the lane's first bounce buffer on the context's stream, then everything
        if (nbytes) {
            rc = lane_pin(ctx, L, 0, align_up(nbytes, 1 << 16));
            if (rc) return bail(rc);
            unsigned char* pin = static_cast<unsigned char*>(L->pin[0]);
            if (!fill(pin, 0, nbytes)) return bail(fail(ctx, GJ_ERR_INVALID, "reading the capture failed"));
            if (hipMemcpyAsync(d_cap, pin, nbytes, hipMemcpyHostToDevice, s) != hipSuccess) return bail(fail(ctx, GJ_ERR_HIP, "host-to-device copy failed"));
            // the next call of this lane may refill the bounce buffer: the copy must have left it when this one returns,
            // which the wait for ev_done below sees to (same stream)
        }
        upload_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_begin).count();
        rc = launch_upto(nbytes, true);
        if (rc) return bail(rc);
    }
    {
        Guard g(ctx);
        rc = follow();
        if (rc) {
        } else if (fused) rc = scan_end(ctx, sj);
        else if (want_scan && n_chunks) {   // odd chunk sizes: K1 alone, then the pass for K3 + K4
            rc = launch_chunk_power(ctx, d_cap, nbytes, plan.chunk_bytes, plan.eps, plan.power_flags, d_power);
            if (!rc) rc = launch_amp_onset(ctx, d_cap, nbytes, plan.rssi_threshold, d_amp, plan.noise_samples, plan.window, plan.factor, d_onset);
        }
        if (!rc && want_welch && rows) rc = welch_end(ctx, wj, plan.welch_flags, d_psd, d_db);
        if (!rc && (hipEventRecord(L->ev_stop, s) != hipSuccess ||
                    (res_bytes && hipMemcpyAsync(L->rpin, L->stage, res_bytes, hipMemcpyDeviceToHost, s) != hipSuccess) ||
                    hipEventRecord(L->ev_done, s) != hipSuccess))
            rc = fail(ctx, GJ_ERR_HIP, "queueing the results failed");
    }
    if (!rc) rc = wait_event(ctx, L->ev_done);
    if (rc) return bail(rc);
    if (want_scan && n_chunks) {
        memcpy(power, L->rpin, n_chunks * sizeof(float));
        memcpy(&res->amp, L->rpin + off_amp, sizeof(gj_amp_stats));
        memcpy(&res->onset, L->rpin + off_onset, sizeof(gj_onset));
    }
    if (want_welch && rows) {
        memcpy(psd, L->rpin + off_psd, nfl * sizeof(float));
        if (psd_db) memcpy(psd_db, L->rpin + off_db, nfl * sizeof(float));
    }
    res->upload_ms = (float)upload_ms;
    res->total_ms = (float)std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_begin).count();
    *dptr = p;
    return GJ_OK;
}

}   // namespace
}   // namespace gj

extern "C" {

int gj_ingest_u8(gj_ctx* ctx, const uint8_t* host, size_t nbytes, const gj_ingest_plan* plan, float* power, size_t power_cap,
                 float* psd, float* psd_db, size_t psd_cap_floats, gj_ingest_result* result, void** dptr) {
    if (!ctx) return GJ_ERR_INVALID;
    if (!plan || (nbytes && !host)) return fail(ctx, GJ_ERR_INVALID, "null argument");
    return ingest_impl(ctx, nbytes, [host](unsigned char* dst, size_t off, size_t len) {
        memcpy(dst, host + off, len);
        return true;
    }, *plan, power, power_cap, psd, psd_db, psd_cap_floats, result, dptr);
}

static int ingest_file_impl(gj_ctx* ctx, const char* path, size_t offset, size_t max_bytes, const gj_ingest_plan* plan, float* power,
                            size_t power_cap, float* psd, float* psd_db, size_t psd_cap_floats, gj_ingest_result* result, void** dptr,
                            int fill_override) {
    if (!ctx) return GJ_ERR_INVALID;
    if (!path || !plan) return fail(ctx, GJ_ERR_INVALID, "null argument");
    const int fd = open(path, O_RDONLY);
    if (fd < 0) return fail(ctx, GJ_ERR_INVALID, "cannot open %s", path);
    struct stat st;
    if (fstat(fd, &st) != 0) {
        close(fd);
        return fail(ctx, GJ_ERR_INVALID, "cannot stat %s", path);
    }
    size_t nbytes = (size_t)st.st_size > offset ? (size_t)st.st_size - offset : 0;
    if (max_bytes && nbytes > max_bytes) nbytes = max_bytes;
    static const bool want_pread = [] {
        const char* e = getenv("GPSJAM_FILE_READ");
        return e && strcmp(e, "pread") == 0;
    }();
    const size_t pg = (size_t)sysconf(_SC_PAGESIZE);
    const size_t map_off = offset / pg * pg, lead = offset - map_off;
    void* m = (nbytes && !want_pread) ? mmap(nullptr, nbytes + lead, PROT_READ, MAP_PRIVATE, fd, (off_t)map_off) : MAP_FAILED;
    int rc;
    if (m != MAP_FAILED) {   // see gj_upload_file for why mapping is the default
        (void)madvise(m, nbytes + lead, MADV_SEQUENTIAL);
        const unsigned char* src = static_cast<const unsigned char*>(m) + lead;
        rc = ingest_impl(ctx, nbytes, [src](unsigned char* dst, size_t off, size_t len) {
            memcpy(dst, src + off, len);
            return true;
        }, *plan, power, power_cap, psd, psd_db, psd_cap_floats, result, dptr, fill_override);
        (void)munmap(m, nbytes + lead);
    } else {
        (void)posix_fadvise(fd, 0, 0, POSIX_FADV_NOREUSE);
        rc = ingest_impl(ctx, nbytes, [fd, offset](unsigned char* dst, size_t off, size_t len) {
            size_t done = 0;
            while (done < len) {
                const ssize_t k = pread(fd, dst + done, len - done, (off_t)(offset + off + done));
                if (k <= 0) return false;
                done += (size_t)k;
            }
            return true;
        }, *plan, power, power_cap, psd, psd_db, psd_cap_floats, result, dptr, fill_override);
    }
    close(fd);
    return rc;
}

int gj_ingest_file(gj_ctx* ctx, const char* path, size_t offset, size_t max_bytes, const gj_ingest_plan* plan, float* power,
                   size_t power_cap, float* psd, float* psd_db, size_t psd_cap_floats, gj_ingest_result* result, void** dptr) {
    return ingest_file_impl(ctx, path, offset, max_bytes, plan, power, power_cap, psd, psd_db, psd_cap_floats, result, dptr, 0);
}

// Several capture files at once: one host thread of the library's own per file (the caller's thread takes the first), a
// lane each, the fill threads per file lowered so that the copies share the cores.  The reference reads a deployment's
// recordings one after the other (skrypty/triangulateRSSI.py:160-174); three 10-s files: 4.7 ms one after the other,
// 4.0 at once (tools/ingest3_probe.py).  Threads are started here, not by the host language: three Python threads cost
// more in start-up and GIL hand-offs than the overlap gains (profiles/NOTES_r05.md section 8).
int gj_ingest_files(gj_ctx* ctx, gj_ingest_job* jobs, int n_jobs, const gj_ingest_plan* plan) {
    if (!ctx) return GJ_ERR_INVALID;
    if (!jobs || !plan || n_jobs < 1 || n_jobs > 64) return fail(ctx, GJ_ERR_INVALID, "bad job list (%d jobs)", n_jobs);
    for (int k = 0; k < n_jobs; ++k) {
        if (!jobs[k].path) return fail(ctx, GJ_ERR_INVALID, "job %d: null path", k);
        jobs[k].status = GJ_OK;
        jobs[k].dptr = nullptr;
    }
    NoCancel nc;   // the joins below are cancellation points: a pending cancellation acts after the call
    // fill threads per file for THIS call (the context's setting is left alone: two of these calls at once, or a
    // gj_set_fill_threads meanwhile, used to end with the wrong value restored -- ADVICE r05)
    const int fill_override = (ctx->fill_threads.load(std::memory_order_relaxed) == 0 && n_jobs > 1) ? ((8 / n_jobs) > 2 ? 8 / n_jobs : 2) : 0;
    std::vector<std::string> msgs;
    std::vector<std::thread> pool;
    std::atomic<int> next{0};
    auto run_jobs = [&] {   // every thread takes the next file that nobody has started
        for (int k = next.fetch_add(1); k < n_jobs; k = next.fetch_add(1)) {
            gj_ingest_job& j = jobs[k];
            j.status = ingest_file_impl(ctx, j.path, j.offset, j.max_bytes, plan, j.power, j.power_cap, j.psd, j.psd_db, j.psd_cap_floats,
                                        &j.result, &j.dptr, fill_override);
            if (j.status) msgs[(size_t)k] = last_error_buf();   // the message is per thread: bring it home
        }
    };
    try {
        msgs.resize((size_t)n_jobs);
        // one thread per file, but never more than there are lanes: the rest would only spin in lane_checkout
        const int extra = (n_jobs < kMaxLanes ? n_jobs : kMaxLanes) - 1;
        pool.reserve((size_t)extra);
        for (int k = 0; k < extra; ++k) pool.emplace_back(run_jobs);
    } catch (...) {
        // no memory / no more threads (EAGAIN under a process limit): the calling thread does what is left below;
        // an exception must never leave an extern "C" entry point with joinable threads behind it (std::terminate)
    }
    if (msgs.size() == (size_t)n_jobs) {
        run_jobs();
    } else {
        for (int k = 0; k < n_jobs; ++k) jobs[k].status = GJ_ERR_NOMEM;
    }
    for (auto& t : pool) t.join();
    if (msgs.size() != (size_t)n_jobs) return fail(ctx, GJ_ERR_NOMEM, "out of memory");
    for (int k = 0; k < n_jobs; ++k)
        if (jobs[k].status) return fail(ctx, jobs[k].status, "%s (file %d of %d: %s)", msgs[(size_t)k].c_str(), k, n_jobs, jobs[k].path);
    return GJ_OK;
}

// ---------------------------------------------------------------- host-buffer entry points
// `iq` is a host buffer (staged into the call's lane) or a DEVICE pointer (a resident capture: used in place).
// kernel_ms excludes the copies.
int gj_chunk_power_u8(gj_ctx* ctx, const uint8_t* iq, size_t nbytes, size_t chunk_bytes, float eps, int flags,
                      float* power, size_t power_cap, size_t* n_out, float* kernel_ms) {
    if (!ctx) return GJ_ERR_INVALID;
    if (chunk_bytes == 0) return fail(ctx, GJ_ERR_INVALID, "chunk_bytes must be > 0");
    const size_t n = gj_chunk_count(nbytes, chunk_bytes);
    if (n_out) *n_out = n;
    if (kernel_ms) *kernel_ms = 0.f;
    if (n == 0) return GJ_OK;
    if (!iq || !power) return fail(ctx, GJ_ERR_INVALID, "null buffer");
    if (power_cap < n) return fail(ctx, GJ_ERR_CAPACITY, "power buffer holds %zu, need %zu", power_cap, n);
    HostCall call(ctx);
    int rc = call.begin(iq, nbytes, n * sizeof(float));
    if (!rc) rc = call.run([&] { return launch_chunk_power(ctx, call.d_in, nbytes, chunk_bytes, eps, flags, call.dev<float>()); });
    if (!rc) rc = call.finish(kernel_ms);
    if (rc) return rc;
    memcpy(power, call.host(), n * sizeof(float));
    return GJ_OK;
}

int gj_welch_u8(gj_ctx* ctx, const uint8_t* iq, size_t nbytes, size_t chunk_samples, int nperseg, double fs, int flags,
                float* psd, float* psd_db, size_t cap_floats, size_t* rows_out, float* kernel_ms) {
    if (!ctx) return GJ_ERR_INVALID;
    const size_t rows = gj_welch_rows(nbytes, chunk_samples, nperseg);
    if (rows_out) *rows_out = rows;
    if (kernel_ms) *kernel_ms = 0.f;
    if (nperseg < 16 || nperseg > 4096 || (nperseg & (nperseg - 1)))
        return fail(ctx, GJ_ERR_UNSUPPORTED, "nperseg must be a power of two in [16, 4096]");
    if (rows == 0) return GJ_OK;
    if (!iq || !psd) return fail(ctx, GJ_ERR_INVALID, "null buffer");
    const size_t nfl = rows * (size_t)nperseg;
    if (cap_floats < nfl) return fail(ctx, GJ_ERR_CAPACITY, "psd buffer holds %zu floats, need %zu", cap_floats, nfl);
    HostCall call(ctx);
    int rc = call.begin(iq, nbytes, (psd_db ? 2 : 1) * nfl * sizeof(float));
    if (!rc)
        rc = call.run([&] {
            return launch_welch(ctx, call.d_in, nbytes, chunk_samples, nperseg, fs, flags, call.dev<float>(),
                                psd_db ? call.dev<float>() + nfl : nullptr);
        });
    if (!rc) rc = call.finish(kernel_ms);
    if (rc) return rc;
    memcpy(psd, call.host(), nfl * sizeof(float));
    if (psd_db) memcpy(psd_db, call.host(nfl * sizeof(float)), nfl * sizeof(float));
    return GJ_OK;
}

int gj_amp_stats_u8(gj_ctx* ctx, const uint8_t* iq, size_t nbytes, float threshold, gj_amp_stats* out, float* kernel_ms) {
    if (!ctx) return GJ_ERR_INVALID;
    if (!out || (nbytes && !iq)) return fail(ctx, GJ_ERR_INVALID, "null buffer");
    HostCall call(ctx);
    int rc = call.begin(iq, nbytes, sizeof(gj_amp_stats));
    if (!rc) rc = call.run([&] { return launch_amp_stats(ctx, call.d_in, nbytes, threshold, call.dev<gj_amp_stats>()); });
    if (!rc) rc = call.finish(kernel_ms);
    if (rc) return rc;
    memcpy(out, call.host(), sizeof(gj_amp_stats));
    return GJ_OK;
}

int gj_onset_u8(gj_ctx* ctx, const uint8_t* iq, size_t nbytes, int noise_samples, int window, float factor, gj_onset* out,
                float* kernel_ms) {
    if (!ctx) return GJ_ERR_INVALID;
    if (!out || (nbytes && !iq)) return fail(ctx, GJ_ERR_INVALID, "null buffer");
    HostCall call(ctx);
    int rc = call.begin(iq, nbytes, sizeof(gj_onset));
    if (!rc) rc = call.run([&] { return launch_onset(ctx, call.d_in, nbytes, noise_samples, window, factor, call.dev<gj_onset>()); });
    if (!rc) rc = call.finish(kernel_ms);
    if (rc) return rc;
    memcpy(out, call.host(), sizeof(gj_onset));
    return GJ_OK;
}

// slices[a] may each be a host buffer (staged) or a device pointer into a resident capture (used in place)
int gj_xcorr_lags_u8(gj_ctx* ctx, const uint8_t* const* slices, int n_ant, size_t n_samples, const int32_t* pairs,
                     int n_pairs, int32_t* lags, float* peaks, float* margins, float* kernel_ms) {
    if (!ctx) return GJ_ERR_INVALID;
    if (!slices || !pairs || !lags || !peaks) return fail(ctx, GJ_ERR_INVALID, "null buffer");
    if (n_ant < 1 || n_ant > GJ_MAX_ANTENNAS) return fail(ctx, GJ_ERR_INVALID, "n_ant must be 1..%d", GJ_MAX_ANTENNAS);
    if (n_pairs < 1 || n_pairs > 4096) return fail(ctx, GJ_ERR_INVALID, "bad n_pairs");
    for (int a = 0; a < n_ant; ++a)
        if (!slices[a]) return fail(ctx, GJ_ERR_INVALID, "null slice %d", a);
    HostCall call(ctx);
    gj_lane* L = call.L;
    if (!L) return GJ_ERR_NOMEM;
    // lane staging: [slot x n_ant of staged slices][starts: 16 x int64][lags | peaks | margins]
    const size_t slot = align_up(2 * n_samples, 256);
    const size_t off_starts = slot * (size_t)n_ant + 256;
    const size_t off_res = off_starts + 128;
    const size_t res_bytes = 12 * (size_t)n_pairs;
    int rc = lane_events(ctx, L);
    if (!rc) rc = lane_stage(ctx, L, off_res + res_bytes + 256);
    if (!rc) rc = lane_rpin(ctx, L, res_bytes + 256);
    if (rc) return rc;
    hipStream_t s = current_stream(ctx);
    const uint8_t* d_ptrs[GJ_MAX_ANTENNAS];
    size_t nbytes[GJ_MAX_ANTENNAS];
    for (int a = 0; a < n_ant; ++a) {
        nbytes[a] = 2 * n_samples;
        if (n_samples && is_device_ptr(slices[a])) {
            d_ptrs[a] = slices[a];
        } else {
            if (n_samples) GJ_HIP(ctx, hipMemcpyAsync(L->stage + slot * a, slices[a], 2 * n_samples, hipMemcpyHostToDevice, s));
            d_ptrs[a] = L->stage + slot * a;
        }
    }
    int64_t* d_starts = reinterpret_cast<int64_t*>(L->stage + off_starts);   // zeros: slices start at their first sample
    int32_t* d_lags = reinterpret_cast<int32_t*>(L->stage + off_res);
    float* d_peaks = reinterpret_cast<float*>(L->stage + off_res + 4 * (size_t)n_pairs);
    float* d_margins = reinterpret_cast<float*>(L->stage + off_res + 8 * (size_t)n_pairs);
    GJ_HIP(ctx, hipMemsetAsync(d_starts, 0, 128, s));
    GJ_HIP(ctx, hipEventRecord(L->ev_done, s));
    const int64_t* sp[GJ_MAX_ANTENNAS];
    for (int a = 0; a < n_ant; ++a) sp[a] = d_starts + a;
    call.s = s;
    call.staged = true;
    call.d_res = L->stage + off_res;
    call.result_bytes = res_bytes;
    rc = call.run([&] { return launch_xcorr(ctx, d_ptrs, nbytes, n_ant, sp, n_samples, pairs, n_pairs, d_lags, d_peaks, d_margins); });
    if (!rc) rc = call.finish(kernel_ms);
    if (rc) return rc;
    memcpy(lags, call.host(), 4 * (size_t)n_pairs);
    memcpy(peaks, call.host(4 * (size_t)n_pairs), 4 * (size_t)n_pairs);
    if (margins) memcpy(margins, call.host(8 * (size_t)n_pairs), 4 * (size_t)n_pairs);
    return GJ_OK;
}

}   // extern "C"
