// Internal definitions shared by the HIP translation units of libgpsjam_hip.so.
#pragma once
#include <hip/hip_runtime.h>

#include <pthread.h>

#include <atomic>
#include <cerrno>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <vector>

#include "../../include/gpsjam.h"
#include "fft_core.h"

struct gj_comm;

// What ONE host-buffer call owns from entry to return: device staging for its input and results, pinned bounce
// buffers, a pinned landing area for its results, its own events.  A call checks a lane out under the context
// lock, then stages, waits and copies with NO lock held; two host threads therefore never share a buffer, and a
// caller that is killed inside a wait (the GUI stops an analysis with QThread.terminate(),
// GpsJammerApp/app/ui_mainwindow.py:818-826) leaves nothing locked: its lane is taken back once its thread is gone.
//
// "Gone" is decided by the kernel, not sampled: `owner` is a robust mutex the calling thread holds from check-out to
// check-in.  When a thread ends, the kernel walks its robust list and marks every mutex it still held "owner died"
// BEFORE it clears the thread's tid word (the thing pthread_join waits for), so from the moment a dead caller can be
// joined a trylock on its lane answers EOWNERDEAD -- no window in which a dying thread still looks alive, and no tid
// that a later thread could inherit (round 4's tgkill probe had both defects: GPUTEST_r04 went red on the first).
// Every check-out (and gj_debug_counters) sweeps ALL busy lanes this way, whether or not a free lane exists.
struct gj_lane {
    static constexpr int kPinBufs = 32;   // pinned bounce buffers (2 per fill thread)
    bool busy = false;
    int owner_tid = 0;                    // kernel thread id of the caller that holds the lane (diagnostics only)
    pthread_mutex_t owner;                // robust; held by the calling thread for the life of the check-out
    bool owner_ready = false;
    unsigned char* stage = nullptr;       // grow-only device staging: [input][results]
    size_t stage_bytes = 0;
    void* pin[kPinBufs] = {};
    size_t pin_cap[kPinBufs] = {};        // bytes each holds (grow-only: a piece is 1..16 MiB, by capture size)
    hipEvent_t pin_ev[kPinBufs] = {};
    unsigned char* rpin = nullptr;        // pinned host memory the results are copied into (async D2H)
    size_t rpin_bytes = 0;
    hipEvent_t ev_start = nullptr, ev_stop = nullptr, ev_done = nullptr;
    // overlapped ingest (gj_ingest_*): uploads run on a stream of the lane's own while the kernels on the context's
    // stream work on the pieces that have landed; one event per 16-MiB piece; a workspace of the lane's own, because
    // the scan / Welch state lives across many short lock sections
    hipStream_t copy_stream = nullptr;
    hipEvent_t* piece_ev = nullptr;
    size_t n_piece_ev = 0;
    unsigned char* ws = nullptr;
    size_t ws_bytes = 0;
};

struct gj_retired {                       // an arena replaced while queued kernels may still read it
    void* p;
    hipEvent_t ev;
};

struct gj_ctx {
    int device = 0;
    int num_cus = 256;
    hipStream_t own_stream = nullptr;
    hipStream_t stream = nullptr;
    hipEvent_t ev_start = nullptr, ev_stop = nullptr;   // gj_timer_*
    hipEvent_t ev_switch = nullptr;                     // gj_set_stream: orders the new stream behind the old one
    // Robust + recursive: held only while work is ENQUEUED (never across a host wait, a file read or a staged
    // copy); if its owner dies inside a critical section the next locker gets EOWNERDEAD and carries on.
    pthread_mutex_t mu;
    bool mu_ready = false;
    int owner_deaths = 0;
    gj::cf* d_twiddle = nullptr;   // W_4096^m
    // Arrival counters of the kernels whose workgroups hand results to the LAST one of them to finish (scan tail: word 0;
    // K5: one word per pair from word kSyncXcorr on).  Zero at creation; every kernel that uses one leaves it at zero
    // (its last arriver resets it), so no launch needs a memset in front of it.  Lives behind the constant tables.
    unsigned* d_sync = nullptr;
    unsigned char* ws = nullptr;   // grow-only workspace for partial results
    size_t ws_bytes = 0;
    std::vector<gj_retired> retired;
    std::vector<gj_lane*> lanes;
    std::vector<gj_comm*> comms;   // communicators created on this context (gj_destroy takes them down)
    int lanes_reclaimed = 0;
    std::atomic<int> fill_threads{0};   // gj_set_fill_threads: 0 = by capture size.  Read once per staged copy, without the lock
    int inject_owner_alive = 0;    // gj_debug_inject: the next probes of a busy lane skip the owner check ("alive")
    // unpack convention (gj_set_unpack): sample = (u8 - offset) * scale; off2 = 2 * offset is an integer
    int off2 = 255;
    double scale = 1.0 / 127.5;
    // diagnostics (gj_debug_set_wait_hook): called with NO lock held right before every host-side wait
    void (*wait_hook)(void*, int) = nullptr;
    void* wait_hook_arg = nullptr;
};

namespace gj {

constexpr int kSyncWords = 1024;    // 4 KiB
constexpr int kSyncTail = 0;        // scan_tail_kernel
constexpr int kSyncXcorr = 64;      // xc_cols_kernel<.., 1>: + pair index

// what the kernels need of the unpack convention: v = 2 u - off2 (exact integer), |sample| = |v| * half_scale
struct Unpack {
    int off2;
    float half_scale;
};
inline Unpack unpack_of(const gj_ctx* ctx) { return Unpack{ctx->off2, (float)(ctx->scale * 0.5)}; }
// (2 / scale)^2, snapped to the integer it is meant to be (65025 for the default 1/127.5)
inline double unpack_norm2(const gj_ctx* ctx) {
    double n = 2.0 / ctx->scale;
    const double r = (double)(long long)(n + 0.5);
    if (n - r < 1e-9 && r - n < 1e-9) n = r;
    return n * n;
}

// detailed message of the calling thread's last failure (gj_last_error): per thread, so that two host threads on
// one context never read each other's text
inline char* last_error_buf() {
    static thread_local char buf[512] = {0};
    return buf;
}
inline int fail(gj_ctx* ctx, int code, const char* fmt, ...) {
    (void)ctx;
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(last_error_buf(), 512, fmt, ap);
    va_end(ap);
    return code;
}

#define GJ_HIP(ctx, call)                                                                         \
    do {                                                                                          \
        hipError_t e__ = (call);                                                                  \
        if (e__ != hipSuccess)                                                                    \
            return gj::fail((ctx), GJ_ERR_HIP, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e__), \
                            __FILE__, __LINE__);                                                  \
    } while (0)

#define GJ_LAUNCH_CHECK(ctx)                                                                      \
    do {                                                                                          \
        hipError_t e__ = hipGetLastError();                                                       \
        if (e__ != hipSuccess)                                                                    \
            return gj::fail((ctx), GJ_ERR_HIP, "kernel launch failed: %s (%s:%d)",                \
                            hipGetErrorString(e__), __FILE__, __LINE__);                          \
    } while (0)

// A thread is never CANCELLED inside the library: QThread.terminate() is pthread_cancel on POSIX, deferred
// cancellation acts at the next cancellation point, and several sit inside a call (usleep, pread, joins, the
// runtime's own waits).  Unwinding from there would run through the HIP runtime's frames and through the join of the
// fill threads (std::terminate).  While one of these guards is alive a cancellation request stays pending; it takes
// effect after the call has returned -- bounded by the call (tens of milliseconds per GiB), with nothing half done.
// (A thread that is killed outright leaves no chance to defer anything: that case is what the lanes, the robust
// mutex and the lock-free waits are for.)
struct NoCancel {
    int old = 0;
    NoCancel() { (void)pthread_setcancelstate(PTHREAD_CANCEL_DISABLE, &old); }
    ~NoCancel() { (void)pthread_setcancelstate(old, nullptr); }
    NoCancel(const NoCancel&) = delete;
    NoCancel& operator=(const NoCancel&) = delete;
};

// The context lock, for the ENQUEUE part of a call only.  Nothing that can block for long is done under it:
// waits go through wait_event / wait_stream below, after the Guard has been left.
struct Guard {
    NoCancel nc;
    gj_ctx* c;
    explicit Guard(gj_ctx* ctx) : c(ctx) {
        if (pthread_mutex_lock(&c->mu) == EOWNERDEAD) {   // the previous owner died inside a critical section
            (void)pthread_mutex_consistent(&c->mu);
            ++c->owner_deaths;
        }
        (void)hipSetDevice(c->device);
    }
    ~Guard() { (void)pthread_mutex_unlock(&c->mu); }
    Guard(const Guard&) = delete;
    Guard& operator=(const Guard&) = delete;
};

// wait sites reported to the diagnostic hook
enum WaitSite { kWaitEvent = 1, kWaitStream = 2, kWaitPiece = 3, kWaitLane = 4, kUnderLock = 5 };
inline void wait_hook(gj_ctx* ctx, int site) {
    void (*h)(void*, int) = ctx->wait_hook;
    if (h) h(ctx->wait_hook_arg, site);
}

// grow-only workspace of the *_dev kernels.  Growing never waits: the old arena is retired behind an event on the
// stream and freed later, outside the lock (reap_retired).
int ensure_workspace(gj_ctx* ctx, size_t bytes);
void reap_retired(gj_ctx* ctx);

inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

// ---- device helpers ---------------------------------------------------------------
__device__ __forceinline__ unsigned wave_sum_u32(unsigned v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}
__device__ __forceinline__ unsigned long long wave_sum_u64(unsigned long long v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}
__device__ __forceinline__ double wave_sum_f64(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

// Sum of `v` over each aligned group of G lanes (G = 1..64, power of two), result in every
// lane of the group -- DPP row operations for the in-row steps (quad_perm swaps, row_half_mirror,
// row_mirror: plain VALU, no LDS round trip per step like __shfl_xor's ds_bpermute chain), then
// v_readlane of the four row totals for the 64-lane case (the result is then wave-uniform).
template <int G>
__device__ __forceinline__ int group_sum_dpp(int v) {
    if constexpr (G >= 2) v += __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xf, 0xf, false);    // quad_perm [1,0,3,2]
    if constexpr (G >= 4) v += __builtin_amdgcn_update_dpp(0, v, 0x4E, 0xf, 0xf, false);    // quad_perm [2,3,0,1]
    if constexpr (G >= 8) v += __builtin_amdgcn_update_dpp(0, v, 0x141, 0xf, 0xf, false);   // row_half_mirror
    if constexpr (G >= 16) v += __builtin_amdgcn_update_dpp(0, v, 0x140, 0xf, 0xf, false);  // row_mirror
    if constexpr (G == 32) v += __shfl_xor(v, 16, 64);
    if constexpr (G == 64)
        v = __builtin_amdgcn_readlane(v, 0) + __builtin_amdgcn_readlane(v, 16) + __builtin_amdgcn_readlane(v, 32) +
            __builtin_amdgcn_readlane(v, 48);
    return v;
}

// the same for exactly-representable float sums
template <int G>
__device__ __forceinline__ float group_sum_dpp_f(float v) {
#define GJ_DPP_F(x, ctrl) __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), ctrl, 0xf, 0xf, false))
    if constexpr (G >= 2) v += GJ_DPP_F(v, 0xB1);
    if constexpr (G >= 4) v += GJ_DPP_F(v, 0x4E);
    if constexpr (G >= 8) v += GJ_DPP_F(v, 0x141);
    if constexpr (G >= 16) v += GJ_DPP_F(v, 0x140);
#undef GJ_DPP_F
    if constexpr (G == 32) v += __shfl_xor(v, 16, 64);
    if constexpr (G == 64) {
        const int b = __float_as_int(v);
        v = (__int_as_float(__builtin_amdgcn_readlane(b, 0)) + __int_as_float(__builtin_amdgcn_readlane(b, 16))) +
            (__int_as_float(__builtin_amdgcn_readlane(b, 32)) + __int_as_float(__builtin_amdgcn_readlane(b, 48)));
    }
    return v;
}

// Wave totals of TWO small non-negative integer-valued floats at once (K2's per-step sums of the raw I and Q bytes:
// at most 16 x 255 per lane).  Both ride in one register, 16 bits each, through the four in-row DPP steps (a row of
// 16 lanes sums to < 65536, so the fields never carry into each other); the four row totals are read into SGPRs and
// added by the scalar unit.  13 vector instructions instead of 2 x 11, results wave-uniform and exact.
__device__ __forceinline__ void wave_sum_pair_u16(float a, float b, float& sum_a, float& sum_b) {
    unsigned p = (unsigned)a | ((unsigned)b << 16);
    p += (unsigned)__builtin_amdgcn_update_dpp(0, (int)p, 0xB1, 0xf, 0xf, false);    // quad_perm [1,0,3,2]
    p += (unsigned)__builtin_amdgcn_update_dpp(0, (int)p, 0x4E, 0xf, 0xf, false);    // quad_perm [2,3,0,1]
    p += (unsigned)__builtin_amdgcn_update_dpp(0, (int)p, 0x141, 0xf, 0xf, false);   // row_half_mirror
    p += (unsigned)__builtin_amdgcn_update_dpp(0, (int)p, 0x140, 0xf, 0xf, false);   // row_mirror
    const unsigned r0 = (unsigned)__builtin_amdgcn_readlane((int)p, 0), r1 = (unsigned)__builtin_amdgcn_readlane((int)p, 16),
                   r2 = (unsigned)__builtin_amdgcn_readlane((int)p, 32), r3 = (unsigned)__builtin_amdgcn_readlane((int)p, 48);
    sum_a = (float)((r0 & 0xffffu) + (r1 & 0xffffu) + (r2 & 0xffffu) + (r3 & 0xffffu));
    sum_b = (float)((r0 >> 16) + (r1 >> 16) + (r2 >> 16) + (r3 >> 16));
}

// ---- results handed from many workgroups to the last one of them to finish (MI355X_MICROARCH.md, "inter-workgroup
// visibility": per-XCD L2s are not coherent, a CU's L1 is never refreshed by other CUs' stores) ----------------------------
// Producer, ONE lane, after ITS OWN plain stores of the handed-off record: drain them, write the XCD's L2 back (agent-scope
// release), then add to the counter.  Returns the add's old value: `expected - 1` tells the caller it arrived last.
__device__ __forceinline__ unsigned arrive_release(unsigned* counter) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    return __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// Consumer, the lane whose add came last: invalidate this CU's L1 (agent-scope acquire) and wait for it; the workgroup's other
// waves may read the records after the barrier that follows.  Also puts the counter back to zero for the next launch.
__device__ __forceinline__ void last_arriver_acquire(unsigned* counter) {
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __hip_atomic_store(counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// ---- TDOA slot: the onset-aligned slice of one capture + a validity header, as one message ----
// [int64 flag: 0 = valid, -1 = invalid][int64 start sample in the sender's capture][2 n bytes of I/Q]
// `sample0` / `total`: iq[0] is sample `sample0` of a capture of `total` samples (0 / nsamples for a whole capture).
// The slice is valid when it lies inside the CAPTURE (the reference's rule, skrypty/triangulateTDOA.py:67-77); a part whose
// buffer does not hold a valid slice says so with flag -2 (a sizing error of the caller, never silently wrong data).
// Thread `g0` of `gstride` cooperating threads; `head` writes the header.
__device__ __forceinline__ void tdoa_slot_body(const uint8_t* __restrict__ iq, size_t nsamples, long long s, size_t n,
                                               uint8_t* __restrict__ slot, long long sample0, size_t total, size_t g0,
                                               size_t gstride, bool head) {
    const bool in_capture = s >= 0 && (unsigned long long)s + n <= total;
    const bool held = s >= sample0 && (unsigned long long)(s - sample0) + n <= nsamples;
    const bool ok = in_capture && held;
    if (head) {
        long long* h = reinterpret_cast<long long*>(slot);
        h[0] = ok ? 0 : (in_capture ? -2 : -1);
        h[1] = s;
    }
    const uint16_t* src = reinterpret_cast<const uint16_t*>(iq) + (ok ? s - sample0 : 0);
    uint4* dst = reinterpret_cast<uint4*>(slot + GJ_SLOT_HEADER);
    // the slot is padded to a multiple of 256 bytes: the padding is written too (zeros), so that a slot is a fully
    // defined message
    const size_t ngroups = ((GJ_SLOT_HEADER + 2 * n + 255) / 256 * 256 - GJ_SLOT_HEADER) / 16;
    // whole 16-byte groups of the slice: one 16-byte load each.  The source is only 2-byte aligned (a slice starts at
    // any sample); global memory takes unaligned vector loads, and eight 2-byte loads per group made the copy of a
    // 2^19-sample slice by ONE workgroup (the scan tail's last arriver) a quarter of a millisecond.
    const size_t nfull = ok ? (2 * n) / 16 : 0;
    struct __attribute__((packed, aligned(2))) Group {
        unsigned x, y, z, w;
        __device__ __forceinline__ uint4 get() const { return uint4{x, y, z, w}; }
    };
    const Group* src16 = reinterpret_cast<const Group*>(src);
    size_t g = g0;
    for (; g + 7 * gstride < nfull; g += 8 * gstride) {          // eight loads in flight per thread
        uint4 q[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) q[k] = src16[g + k * gstride].get();
#pragma unroll
        for (int k = 0; k < 8; ++k) dst[g + k * gstride] = q[k];
    }
    for (; g < nfull; g += gstride) dst[g] = src16[g].get();
    // the slice's ragged end and the padding (an invalid slot: everything)
    for (; g < ngroups; g += gstride) {
        unsigned w[4] = {0, 0, 0, 0};
        if (ok) {
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const size_t i = g * 8 + k;
                const unsigned v = (i < n) ? src[i] : 0u;
                w[k >> 1] |= v << (16 * (k & 1));
            }
        }
        dst[g] = uint4{w[0], w[1], w[2], w[3]};
    }
}

// The capture's onset from its parts' (each already in capture coordinates): the part with the smallest start >= 0
// decides index and margin_hit; guard = smallest guard >= 0; margin_before = the smallest reported by the parts up to
// and including that one (all of them when nothing crossed); noise and threshold are the same on every part.  One thread.
__device__ __forceinline__ void onset_combine(const gj_onset* __restrict__ parts, int n, gj_onset* __restrict__ out) {
    int best = -1;
    for (int k = 0; k < n; ++k)
        if (parts[k].start_index >= 0 && (best < 0 || parts[k].start_index < parts[best].start_index)) best = k;
    long long guard = -1;
    float mb = parts[0].margin_before;
    for (int k = 0; k < n; ++k) {
        if (parts[k].guard_index >= 0 && (guard < 0 || parts[k].guard_index < guard)) guard = parts[k].guard_index;
        if ((best < 0 || k <= best) && parts[k].margin_before < mb) mb = parts[k].margin_before;
    }
    gj_onset o = parts[0];
    o.start_index = best >= 0 ? parts[best].start_index : -1;
    o.margin_hit = best >= 0 ? parts[best].margin_hit : 0.f;
    o.margin_before = mb;
    o.guard_index = guard;
    *out = o;
}

}   // namespace gj

// entry points implemented per translation unit (called from api.hip)
namespace gj {
// K2 and the fused scan in three steps (begin / range / end), so that gj_ingest_* can launch them on the pieces of a
// capture that have landed in HBM while the rest is still being uploaded
struct WelchJob {
    alignas(8) unsigned char plan[96];   // WelchPlan (k_welch.hip)
    int nperseg = 0;
    size_t rows = 0, ws_bytes = 0;
    float* partial = nullptr;            // per-workgroup spectra: ws_bytes of workspace, set by the caller
};
int welch_begin(gj_ctx*, size_t nbytes, size_t chunk_samples, int nperseg, double fs, size_t plan_bytes, WelchJob&);
int welch_range(gj_ctx*, const WelchJob&, const uint8_t* d_iq, size_t chunk0, size_t chunk1);
int welch_end(gj_ctx*, const WelchJob&, int flags, float* d_psd, float* d_psd_db);
struct ScanJob {
    alignas(8) unsigned char state[320];   // ScanState (k_scan.hip)
    size_t ntiles = 0, nchunks = 0, ws_bytes = 0;
    unsigned char* ws = nullptr;           // ws_bytes of workspace, set by the caller before scan_start
};
int scan_begin(gj_ctx*, const uint8_t* d_iq, size_t nbytes, size_t chunk_bytes, float eps, int flags, float* d_power,
               float rssi_threshold, gj_amp_stats* d_amp, int noise_samples, int window, float factor, gj_onset* d_onset,
               ScanJob&);
int scan_start(gj_ctx*, ScanJob&);                              // clears the accumulators (after job.ws is set)
int scan_range(gj_ctx*, const ScanJob&, size_t tile0, size_t tile1);
// what the scan's tail launch can do on top of K1 + K3 + K4 (all optional): the noise-floor threshold of the power map
// (gj_power_threshold_dev) and the TDOA slot cut at the onset (gj_tdoa_slot_dev with d_start = &onset.start_index)
struct ScanExtra {
    float pct = 5.f, rise_db = 6.f;
    float* d_stats = nullptr;      // [3]; nullptr: no threshold
    uint8_t* d_mask = nullptr;     // [n_chunks] or nullptr
    uint8_t* d_slot = nullptr;     // nullptr: no slot
    size_t slice_samples = 0;
    size_t slot_buf_bytes = 0;     // bytes of the buffer a slice may be cut from (a part: its whole buffer, tail included); 0: the scanned bytes
};
int scan_end(gj_ctx*, const ScanJob&, const ScanExtra* extra = nullptr);
bool scan_fusable(const uint8_t* d_iq, size_t nbytes, size_t chunk_bytes);
int launch_chunk_power(gj_ctx*, const uint8_t*, size_t, size_t, float, int, float*);
int launch_power_threshold(gj_ctx*, const float*, size_t, float, float, float*, uint8_t*);
int launch_amp_stats(gj_ctx*, const uint8_t*, size_t, float, gj_amp_stats*);
int launch_onset(gj_ctx*, const uint8_t*, size_t, int, int, float, gj_onset*);
int launch_amp_onset(gj_ctx*, const uint8_t*, size_t, float, gj_amp_stats*, int, int, float, gj_onset*);   // either output may be nullptr
int launch_stream_scan(gj_ctx*, const uint8_t*, size_t, size_t, float, int, float*, float, gj_amp_stats*, int, int,
                       float, gj_onset*, const ScanExtra* extra = nullptr);
int launch_histogram(gj_ctx*, const uint8_t*, size_t, size_t, int, int, unsigned long long*);
int launch_welch(gj_ctx*, const uint8_t*, size_t, size_t, int, double, int, float*, float*, size_t plan_bytes = 0);
size_t welch_workspace(gj_ctx*, size_t, size_t, int, size_t plan_bytes = 0);
int launch_welch_batch(gj_ctx*, const uint8_t* const*, int, size_t, size_t, int, double, int, float* const*);
int launch_pack_results(gj_ctx*, const gj_combine_capture*, int, int, const int32_t*, const int32_t*, const float*, const float*);
int launch_xcorr(gj_ctx*, const uint8_t* const*, const size_t*, int, const int64_t* const*, size_t,
                 const int32_t*, int, int32_t*, float*, float*);
int launch_tdoa_slot(gj_ctx*, const uint8_t*, size_t, const int64_t*, size_t, uint8_t*, long long sample0 = 0, size_t total_samples = 0);
int launch_slots_pick(gj_ctx*, const uint8_t*, size_t, const int*, const int*, int, uint8_t*);
int launch_part_scan(gj_ctx*, const gj_part_view&, size_t, float, int, float*, float, void*, gj_amp_part*, int, int, float, gj_onset*,
                     const ScanExtra* extra = nullptr);
size_t amp_tile_count(size_t nbytes);
int launch_amp_combine(gj_ctx*, const void*, size_t, const gj_amp_part*, int, size_t, gj_amp_stats*);
int launch_onset_combine(gj_ctx*, const gj_onset*, int, gj_onset*);
int launch_pack_part(gj_ctx*, const gj_part_pack&, double*);
int launch_combine_stats(gj_ctx*, const gj_combine_capture*, int, float, float);
int combine_plan_create(gj_ctx*, const gj_combine_copy*, int, const gj_combine_capture*, int, size_t, const void*, size_t, int, float,
                        float, const int32_t*, const int32_t*, const float*, const float*, gj_combine_plan**);
int combine_plan_check(gj_ctx*, const gj_combine_copy*, int, const gj_combine_capture*, int, size_t, const void*, size_t, int, bool, size_t*,
                       size_t*, int*);
void combine_plan_destroy(gj_combine_plan*);
int launch_split_combine(gj_ctx*, const gj_combine_plan*, const double*);
int launch_acq_search(gj_ctx*, const uint8_t*, size_t, size_t, int, int, const int16_t*, int, const uint8_t*, int, int, double,
                      float, gj_acq_result*, double*);
size_t acq_workspace(int, int, int, int, bool);
size_t xcorr_workspace(gj_ctx*, int, size_t, int);
int launch_synth(gj_ctx*, const gj_synth_params&, int64_t, size_t, uint8_t*);
int launch_pack_result(gj_ctx*, size_t, const float*, const float*, const gj_amp_stats*, const gj_onset*, const float*, size_t,
                       int, int, int, int, const int32_t*, const int32_t*, const float*, const float*, double*);
}   // namespace gj
