// C-ABI of libgpsjam_hip.so (see include/gpsjam.h): context, device memory, stopwatch and the device-pointer
// entry points.  The host-buffer entry points (staging, uploads, lanes) are in host_io.hip.
//
// Locking rule: ctx->mu is held while work is ENQUEUED and never across a
// host-side wait.  Every wait in this library is one of wait_stream / wait_event below, entered without the lock.
#include <cmath>
#include <new>
#include <vector>

#include "gj_common.h"
#include "host_io.h"

namespace gj {

constexpr int kWindowFloats = 2 * 4096;   // periodic Hann tables for N = 16..4096 at offset N-16

// Called under the lock (from the launch_* functions).  Kernels queued on the stream may still be using the old
// arena, and waiting for them here would be a wait under the lock: the old arena is retired behind an event and
// freed by reap_retired once that event has passed.
int ensure_workspace(gj_ctx* ctx, size_t bytes) {
    if (bytes <= ctx->ws_bytes) return GJ_OK;
    bytes = align_up(bytes + bytes / 8, 1 << 20);
    void* p = nullptr;
    if (hipMalloc(&p, bytes) != hipSuccess) return fail(ctx, GJ_ERR_NOMEM, "workspace of %zu bytes", bytes);
    if (ctx->ws) {
        gj_retired r{ctx->ws, nullptr};
        if (hipEventCreateWithFlags(&r.ev, hipEventDisableTiming) == hipSuccess) (void)hipEventRecord(r.ev, ctx->stream);
        ctx->retired.push_back(r);
    }
    ctx->ws = static_cast<unsigned char*>(p);
    ctx->ws_bytes = bytes;
    return GJ_OK;
}

// Free the arenas whose last reader has finished.  The list is taken under the lock, the frees happen outside it
// (hipFree may wait for the device).
void reap_retired(gj_ctx* ctx) {
    std::vector<gj_retired> ready;
    {
        Guard g(ctx);
        for (size_t k = 0; k < ctx->retired.size();) {
            gj_retired& r = ctx->retired[k];
            if (!r.ev || hipEventQuery(r.ev) == hipSuccess) {
                ready.push_back(r);
                ctx->retired.erase(ctx->retired.begin() + (long)k);
            } else {
                ++k;
            }
        }
        (void)hipGetLastError();   // hipErrorNotReady from the query is not a failure
    }
    for (gj_retired& r : ready) {
        (void)hipFree(r.p);
        if (r.ev) (void)hipEventDestroy(r.ev);
    }
}

// ---- the only host-side waits of the library: entered WITHOUT the context lock ------------------------------
int wait_stream(gj_ctx* ctx, hipStream_t stream) {
    NoCancel nc;
    wait_hook(ctx, kWaitStream);
    GJ_HIP(ctx, hipStreamSynchronize(stream));
    return GJ_OK;
}

int wait_event(gj_ctx* ctx, hipEvent_t ev) {
    NoCancel nc;
    wait_hook(ctx, kWaitEvent);
    GJ_HIP(ctx, hipEventSynchronize(ev));
    return GJ_OK;
}

// the stream the context currently runs on, read under the lock
hipStream_t current_stream(gj_ctx* ctx) {
    Guard g(ctx);
    return ctx->stream;
}

static float* g_window_dummy = nullptr;
const float* window_table(gj_ctx* ctx, int n) {
    (void)g_window_dummy;
    return reinterpret_cast<const float*>(ctx->d_twiddle + kTwiddleTable) + (n - 16);
}


}   // namespace gj

using namespace gj;

extern "C" {

int gj_version(void) { return GJ_VERSION; }

const char* gj_strerror(int status) {
    switch (status) {
        case GJ_OK: return "ok";
        case GJ_ERR_INVALID: return "invalid argument";
        case GJ_ERR_HIP: return "HIP runtime error";
        case GJ_ERR_NOMEM: return "out of memory";
        case GJ_ERR_NODEVICE: return "no such GPU";
        case GJ_ERR_UNSUPPORTED: return "unsupported size or parameter";
        case GJ_ERR_CAPACITY: return "output buffer too small";
        default: return "unknown status";
    }
}

const char* gj_last_error(gj_ctx* ctx) { return ctx ? last_error_buf() : "null context"; }

int gj_device_count(int* count) {
    if (!count) return GJ_ERR_INVALID;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) n = 0;
    *count = n;
    return GJ_OK;
}

int gj_create(int device_id, gj_ctx** out) {
    if (!out) return GJ_ERR_INVALID;
    *out = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || device_id < 0 || device_id >= n) return GJ_ERR_NODEVICE;
    gj_ctx* ctx = new (std::nothrow) gj_ctx();
    if (!ctx) return GJ_ERR_NOMEM;
    ctx->device = device_id;
    {
        pthread_mutexattr_t at;
        pthread_mutexattr_init(&at);
        pthread_mutexattr_settype(&at, PTHREAD_MUTEX_RECURSIVE);
        pthread_mutexattr_setrobust(&at, PTHREAD_MUTEX_ROBUST);   // an owner that dies does not take the context with it
        ctx->mu_ready = pthread_mutex_init(&ctx->mu, &at) == 0;
        pthread_mutexattr_destroy(&at);
        if (!ctx->mu_ready) {
            delete ctx;
            return GJ_ERR_NOMEM;
        }
    }
    auto bail = [&](int code) {
        gj_destroy(ctx);
        return code;
    };
    if (hipSetDevice(device_id) != hipSuccess) return bail(GJ_ERR_HIP);
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device_id) == hipSuccess) ctx->num_cus = prop.multiProcessorCount;
    if (hipStreamCreateWithFlags(&ctx->own_stream, hipStreamNonBlocking) != hipSuccess) return bail(GJ_ERR_HIP);
    ctx->stream = ctx->own_stream;
    if (hipEventCreate(&ctx->ev_start) != hipSuccess || hipEventCreate(&ctx->ev_stop) != hipSuccess ||
        hipEventCreateWithFlags(&ctx->ev_switch, hipEventDisableTiming) != hipSuccess) return bail(GJ_ERR_HIP);
    // constant tables: W_4096^m, periodic Hann windows
    const size_t tables = align_up(kTwiddleTable * sizeof(cf) + kWindowFloats * sizeof(float) + 256, 256);
    const size_t bytes = tables + kSyncWords * sizeof(unsigned);   // the arrival counters start out (and stay) zero
    void* p = nullptr;
    if (hipMalloc(&p, bytes) != hipSuccess) return bail(GJ_ERR_NOMEM);
    ctx->d_twiddle = static_cast<cf*>(p);
    std::vector<unsigned char> host(bytes, 0);
    cf* tw = reinterpret_cast<cf*>(host.data());
    for (int m = 0; m < kTwiddleTable; ++m) {
        const double a = -2.0 * M_PI * (double)m / (double)kTwiddleTable;
        tw[m] = cf{(float)std::cos(a), (float)std::sin(a)};
    }
    float* win = reinterpret_cast<float*>(host.data() + kTwiddleTable * sizeof(cf));
    for (int nn = 16; nn <= 4096; nn *= 2)
        for (int k = 0; k < nn; ++k) win[nn - 16 + k] = (float)(0.5 - 0.5 * std::cos(2.0 * M_PI * (double)k / (double)nn));
    if (hipMemcpy(p, host.data(), bytes, hipMemcpyHostToDevice) != hipSuccess) return bail(GJ_ERR_HIP);
    ctx->d_sync = reinterpret_cast<unsigned*>(static_cast<unsigned char*>(p) + tables);
    *out = ctx;
    return GJ_OK;
}

int gj_destroy(gj_ctx* ctx) {
    if (!ctx) return GJ_OK;
    (void)hipSetDevice(ctx->device);
    if (ctx->own_stream) (void)hipStreamSynchronize(ctx->own_stream);
    if (ctx->stream && ctx->stream != ctx->own_stream) (void)hipStreamSynchronize(ctx->stream);
    comm_detach_all(ctx);   // communicators created on this context: taken down, their handles stay valid to destroy
    if (ctx->ws) (void)hipFree(ctx->ws);
    for (gj_retired& r : ctx->retired) {
        (void)hipFree(r.p);
        if (r.ev) (void)hipEventDestroy(r.ev);
    }
    for (gj_lane* L : ctx->lanes) lane_free(L);
    if (ctx->d_twiddle) (void)hipFree(ctx->d_twiddle);
    if (ctx->ev_start) (void)hipEventDestroy(ctx->ev_start);
    if (ctx->ev_stop) (void)hipEventDestroy(ctx->ev_stop);
    if (ctx->ev_switch) (void)hipEventDestroy(ctx->ev_switch);
    if (ctx->own_stream) (void)hipStreamDestroy(ctx->own_stream);
    if (ctx->mu_ready) (void)pthread_mutex_destroy(&ctx->mu);
    delete ctx;
    return GJ_OK;
}

int gj_debug_set_wait_hook(gj_ctx* ctx, void (*hook)(void*, int), void* arg) {
    if (!ctx) return GJ_ERR_INVALID;
    Guard g(ctx);
    ctx->wait_hook_arg = arg;
    ctx->wait_hook = hook;
    return GJ_OK;
}

int gj_debug_inject(gj_ctx* ctx, int what, int count) {
    if (!ctx) return GJ_ERR_INVALID;
    if (what != GJ_INJECT_OWNER_ALIVE || count < 0) return fail(ctx, GJ_ERR_INVALID, "unknown injection %d (count %d)", what, count);
    Guard g(ctx);
    ctx->inject_owner_alive = count;
    return GJ_OK;
}

// one wave that spins on the constant-rate (100 MHz) real-time counter: keeps the context's stream -- and the hardware
// queue behind it -- busy for a known time without occupying the chip
__global__ void probe_busy_kernel(unsigned long long ticks) {
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) __builtin_amdgcn_s_sleep(32);
}

int gj_probe_busy_dev(gj_ctx* ctx, float milliseconds) {
    if (!ctx) return GJ_ERR_INVALID;
    if (!(milliseconds >= 0.f) || milliseconds > 100.f) return fail(ctx, GJ_ERR_INVALID, "busy time %g ms (0..100)", (double)milliseconds);
    Guard g(ctx);
    hipLaunchKernelGGL(probe_busy_kernel, dim3(1), dim3(64), 0, ctx->stream, (unsigned long long)((double)milliseconds * 1e5));
    GJ_LAUNCH_CHECK(ctx);
    return GJ_OK;
}

int gj_debug_counters(gj_ctx* ctx, int* lanes, int* lanes_busy, int* lanes_reclaimed, int* owner_deaths) {
    if (!ctx) return GJ_ERR_INVALID;
    (void)lane_sweep(ctx);        // lanes of callers that have ended come back here as well as at a check-out
    Guard g(ctx);
    wait_hook(ctx, kUnderLock);   // the one hook site that runs WITH the lock held (dead-owner recovery test)
    int busy = 0;
    for (gj_lane* L : ctx->lanes) busy += L->busy ? 1 : 0;
    if (lanes) *lanes = (int)ctx->lanes.size();
    if (lanes_busy) *lanes_busy = busy;
    if (lanes_reclaimed) *lanes_reclaimed = ctx->lanes_reclaimed;
    if (owner_deaths) *owner_deaths = ctx->owner_deaths;
    return GJ_OK;
}

// The new stream is ordered behind everything the context has queued on the old one (an event on the old stream, a wait
// on the new): a context's kernels hand results between workgroups through arrival counters that are per CONTEXT
// (ctx->d_sync), and its workspace is one arena -- two launches of one context must never be in flight on two streams
// at once.  A stream that is being captured into a graph is left alone (recording into a capture from
// outside it would end the capture): the pipelines switch streams at construction, before any capture.
int gj_set_stream(gj_ctx* ctx, void* hip_stream, int external) {
    if (!ctx) return GJ_ERR_INVALID;
    Guard g(ctx);
    const hipStream_t to = external ? static_cast<hipStream_t>(hip_stream) : ctx->own_stream;
    if (to != ctx->stream && ctx->ev_switch) {
        hipStreamCaptureStatus a = hipStreamCaptureStatusNone, b = hipStreamCaptureStatusNone;
        if (hipStreamIsCapturing(ctx->stream, &a) == hipSuccess && hipStreamIsCapturing(to, &b) == hipSuccess &&
            a == hipStreamCaptureStatusNone && b == hipStreamCaptureStatusNone) {
            // best effort: an old stream its owner has already destroyed cannot be recorded on -- and has nothing in flight
            if (hipEventRecord(ctx->ev_switch, ctx->stream) != hipSuccess || hipStreamWaitEvent(to, ctx->ev_switch, 0) != hipSuccess)
                (void)hipGetLastError();
        } else {
            (void)hipGetLastError();
        }
    }
    ctx->stream = to;
    return GJ_OK;
}

int gj_set_unpack(gj_ctx* ctx, double offset, double scale) {
    if (!ctx) return GJ_ERR_INVALID;
    Guard g(ctx);
    const double o2 = 2.0 * offset;
    if (!(o2 >= 0.0 && o2 <= 510.0) || o2 != (double)(int)o2) return fail(ctx, GJ_ERR_INVALID, "offset must be a multiple of 0.5 in [0, 255]");
    if (!(scale > 0.0) || !(scale < 1e6)) return fail(ctx, GJ_ERR_INVALID, "scale must be positive");
    ctx->off2 = (int)o2;
    ctx->scale = scale;
    return GJ_OK;
}

int gj_set_fill_threads(gj_ctx* ctx, int n) {
    if (!ctx) return GJ_ERR_INVALID;
    if (n < 0 || n > 16) return fail(ctx, GJ_ERR_INVALID, "fill threads %d (0 = by capture size, 1..16)", n);
    ctx->fill_threads.store(n, std::memory_order_relaxed);
    return GJ_OK;
}

int gj_get_unpack(gj_ctx* ctx, double* offset, double* scale) {
    if (!ctx) return GJ_ERR_INVALID;
    Guard g(ctx);
    if (offset) *offset = 0.5 * ctx->off2;
    if (scale) *scale = ctx->scale;
    return GJ_OK;
}

int gj_synchronize(gj_ctx* ctx) {
    if (!ctx) return GJ_ERR_INVALID;
    const hipStream_t s = current_stream(ctx);
    const int rc = wait_stream(ctx, s);
    reap_retired(ctx);
    return rc;
}

int gj_device_info(gj_ctx* ctx, char* name, size_t name_cap, int* compute_units, uint64_t* hbm_bytes) {
    if (!ctx) return GJ_ERR_INVALID;
    Guard g(ctx);
    hipDeviceProp_t prop;
    GJ_HIP(ctx, hipGetDeviceProperties(&prop, ctx->device));
    if (name && name_cap) snprintf(name, name_cap, "%s (%s)", prop.name, prop.gcnArchName);
    if (compute_units) *compute_units = prop.multiProcessorCount;
    if (hbm_bytes) *hbm_bytes = prop.totalGlobalMem;
    return GJ_OK;
}

int gj_device_identity(gj_ctx* ctx, char* out, size_t cap) {
    if (!ctx || !out || cap < 2) return GJ_ERR_INVALID;
    Guard g(ctx);
    char pci[32] = "?";
    GJ_HIP(ctx, hipDeviceGetPCIBusId(pci, (int)sizeof(pci), ctx->device));
    hipUUID uuid;
    memset(&uuid, 0, sizeof(uuid));
    char hex[2 * sizeof(uuid.bytes) + 1] = {0};
    if (hipDeviceGetUuid(&uuid, ctx->device) == hipSuccess)
        for (size_t k = 0; k < sizeof(uuid.bytes); ++k) snprintf(hex + 2 * k, 3, "%02x", (unsigned char)uuid.bytes[k]);
    else
        (void)hipGetLastError();
    snprintf(out, cap, "pci=%s uuid=%s hip=%d", pci, hex[0] ? hex : "?", ctx->device);
    return GJ_OK;
}

int gj_reserve(gj_ctx* ctx, size_t workspace_bytes) {
    if (!ctx) return GJ_ERR_INVALID;
    int rc;
    {
        Guard g(ctx);
        rc = ensure_workspace(ctx, workspace_bytes);
    }
    reap_retired(ctx);
    return rc;
}

int gj_malloc(gj_ctx* ctx, size_t bytes, void** dptr) {
    if (!ctx || !dptr) return GJ_ERR_INVALID;
    *dptr = nullptr;
    if (bytes == 0) return GJ_OK;
    (void)hipSetDevice(ctx->device);
    if (hipMalloc(dptr, bytes) != hipSuccess) return fail(ctx, GJ_ERR_NOMEM, "hipMalloc(%zu)", bytes);
    return GJ_OK;
}

int gj_free(gj_ctx* ctx, void* dptr) {
    if (!ctx) return GJ_ERR_INVALID;
    if (dptr) {
        const hipStream_t s = current_stream(ctx);   // kernels queued on the context's stream may still read it
        int rc = wait_stream(ctx, s);
        if (rc) return rc;
        GJ_HIP(ctx, hipFree(dptr));
    }
    return GJ_OK;
}

// The two blocking copies run on the context's stream (ordered behind what is queued there) and wait with no lock
// held; with pageable host memory the copy call itself may block, which is why it is outside the lock as well.
int gj_memcpy_h2d(gj_ctx* ctx, void* dst_dev, const void* src_host, size_t bytes) {
    if (!ctx || (bytes && (!dst_dev || !src_host))) return GJ_ERR_INVALID;
    if (bytes) {
        const hipStream_t s = current_stream(ctx);
        GJ_HIP(ctx, hipMemcpyAsync(dst_dev, src_host, bytes, hipMemcpyHostToDevice, s));
        return wait_stream(ctx, s);
    }
    return GJ_OK;
}

int gj_memcpy_d2h(gj_ctx* ctx, void* dst_host, const void* src_dev, size_t bytes) {
    if (!ctx || (bytes && (!dst_host || !src_dev))) return GJ_ERR_INVALID;
    if (bytes) {
        const hipStream_t s = current_stream(ctx);
        GJ_HIP(ctx, hipMemcpyAsync(dst_host, src_dev, bytes, hipMemcpyDeviceToHost, s));
        return wait_stream(ctx, s);
    }
    return GJ_OK;
}

int gj_timer_start(gj_ctx* ctx) {
    if (!ctx) return GJ_ERR_INVALID;
    Guard g(ctx);
    GJ_HIP(ctx, hipEventRecord(ctx->ev_start, ctx->stream));
    return GJ_OK;
}

int gj_timer_stop(gj_ctx* ctx, float* elapsed_ms) {
    if (!ctx || !elapsed_ms) return GJ_ERR_INVALID;
    {
        Guard g(ctx);
        GJ_HIP(ctx, hipEventRecord(ctx->ev_stop, ctx->stream));
    }
    int rc = wait_event(ctx, ctx->ev_stop);
    if (rc) return rc;
    GJ_HIP(ctx, hipEventElapsedTime(elapsed_ms, ctx->ev_start, ctx->ev_stop));
    return GJ_OK;
}

// ---------------------------------------------------------------- sizes
size_t gj_chunk_count(size_t nbytes, size_t chunk_bytes) {
    return chunk_bytes ? (nbytes + chunk_bytes - 1) / chunk_bytes : 0;
}

size_t gj_welch_rows(size_t nbytes, size_t chunk_samples, int nperseg) {
    if (chunk_samples == 0 || nperseg <= 0) return 0;
    const size_t chunk_bytes = 2 * chunk_samples;
    const size_t full = nbytes / chunk_bytes;
    const size_t rem = nbytes - full * chunk_bytes;
    return full + ((rem >= (size_t)2 * nperseg) ? 1 : 0);   // widmo_plot.py:31
}

size_t gj_welch_workspace(gj_ctx* ctx, size_t nbytes, size_t chunk_samples, int nperseg) {
    return ctx ? welch_workspace(ctx, nbytes, chunk_samples, nperseg) : 0;
}

size_t gj_xcorr_workspace(gj_ctx* ctx, int n_ant, size_t n_samples, int n_pairs) {
    return xcorr_workspace(ctx, n_ant, n_samples, n_pairs);
}

// ---------------------------------------------------------------- device entry points
#define GJ_ENTER(ctx)                 \
    if (!(ctx)) return GJ_ERR_INVALID; \
    Guard guard__(ctx)

int gj_chunk_power_dev(gj_ctx* ctx, const uint8_t* d_iq, size_t nbytes, size_t chunk_bytes, float eps, int flags,
                       float* d_power) {
    GJ_ENTER(ctx);
    if (nbytes && (!d_iq || !d_power)) return fail(ctx, GJ_ERR_INVALID, "null buffer");
    return launch_chunk_power(ctx, d_iq, nbytes, chunk_bytes, eps, flags, d_power);
}

int gj_power_threshold_dev(gj_ctx* ctx, const float* d_power, size_t n, float pct, float rise_db, float* d_stats,
                           uint8_t* d_mask) {
    GJ_ENTER(ctx);
    if (!d_power || !d_stats) return fail(ctx, GJ_ERR_INVALID, "null buffer");
    return launch_power_threshold(ctx, d_power, n, pct, rise_db, d_stats, d_mask);
}

int gj_welch_dev(gj_ctx* ctx, const uint8_t* d_iq, size_t nbytes, size_t chunk_samples, int nperseg, double fs, int flags,
                 float* d_psd, float* d_psd_db) {
    GJ_ENTER(ctx);
    if (nbytes && (!d_iq || !d_psd)) return fail(ctx, GJ_ERR_INVALID, "null buffer");
    return launch_welch(ctx, d_iq, nbytes, chunk_samples, nperseg, fs, flags, d_psd, d_psd_db);
}

// gj_welch_dev with HIP events around the transform launch and around the finalize launch (bench.py: K2 at two sizes,
// interleaved, kernel and finalize apart).  Synchronises on its last event, with no lock held.
int gj_welch_timed_dev(gj_ctx* ctx, const uint8_t* d_iq, size_t nbytes, size_t chunk_samples, int nperseg, double fs, int flags,
                       float* d_psd, float* d_psd_db, float* kernel_ms, float* finalize_ms) {
    if (!ctx) return GJ_ERR_INVALID;
    if (!kernel_ms || !finalize_ms || (nbytes && (!d_iq || !d_psd))) return fail(ctx, GJ_ERR_INVALID, "null buffer");
    *kernel_ms = *finalize_ms = 0.f;
    // events of the call's own: the context-wide pair belongs to gj_timer_*, and a second thread re-recording it between
    // this call's lock section and its wait would time something else
    hipEvent_t ev0 = nullptr, mid = nullptr, ev1 = nullptr;
    (void)hipSetDevice(ctx->device);
    if (hipEventCreate(&ev0) != hipSuccess || hipEventCreate(&mid) != hipSuccess || hipEventCreate(&ev1) != hipSuccess) {
        for (hipEvent_t e : {ev0, mid, ev1})
            if (e) (void)hipEventDestroy(e);
        return fail(ctx, GJ_ERR_HIP, "hipEventCreate failed");
    }
    int rc = GJ_OK;
    bool ran = false;
    {
        Guard g(ctx);
        WelchJob job;
        rc = welch_begin(ctx, nbytes, chunk_samples, nperseg, fs, 0, job);
        if (!rc && (reinterpret_cast<uintptr_t>(d_iq) & 1) != 0) rc = fail(ctx, GJ_ERR_INVALID, "capture must be 2-byte aligned");
        if (!rc && job.rows) {
            rc = ensure_workspace(ctx, job.ws_bytes);
            job.partial = reinterpret_cast<float*>(ctx->ws);
            if (!rc && hipEventRecord(ev0, ctx->stream) != hipSuccess) rc = fail(ctx, GJ_ERR_HIP, "event");
            if (!rc) rc = welch_range(ctx, job, d_iq, 0, job.rows);
            if (!rc && hipEventRecord(mid, ctx->stream) != hipSuccess) rc = fail(ctx, GJ_ERR_HIP, "event");
            if (!rc) rc = welch_end(ctx, job, flags, d_psd, d_psd_db);
            if (!rc && hipEventRecord(ev1, ctx->stream) != hipSuccess) rc = fail(ctx, GJ_ERR_HIP, "event");
            ran = !rc;
        }
    }
    if (ran) rc = wait_event(ctx, ev1);
    if (ran && !rc) {
        if (hipEventElapsedTime(kernel_ms, ev0, mid) != hipSuccess || hipEventElapsedTime(finalize_ms, mid, ev1) != hipSuccess)
            rc = fail(ctx, GJ_ERR_HIP, "hipEventElapsedTime failed");
    }
    for (hipEvent_t e : {ev0, mid, ev1}) (void)hipEventDestroy(e);
    reap_retired(ctx);
    return rc;
}

int gj_welch_batch_dev(gj_ctx* ctx, const uint8_t* const* d_iq, int n_captures, size_t nbytes_each, size_t chunk_samples, int nperseg,
                       double fs, int flags, float* const* d_psd) {
    GJ_ENTER(ctx);
    if (!d_iq || !d_psd) return fail(ctx, GJ_ERR_INVALID, "null buffer");
    return launch_welch_batch(ctx, d_iq, n_captures, nbytes_each, chunk_samples, nperseg, fs, flags, d_psd);
}

int gj_byte_histogram_dev(gj_ctx* ctx, const uint8_t* d_iq, size_t nbytes, size_t chunk_samples, int nperseg, int stride,
                          uint64_t* d_hist) {
    GJ_ENTER(ctx);
    if (!d_hist || (nbytes && !d_iq)) return fail(ctx, GJ_ERR_INVALID, "null buffer");
    return launch_histogram(ctx, d_iq, nbytes, chunk_samples, nperseg, stride,
                            reinterpret_cast<unsigned long long*>(d_hist));
}

int gj_amp_stats_dev(gj_ctx* ctx, const uint8_t* d_iq, size_t nbytes, float threshold, gj_amp_stats* d_out) {
    GJ_ENTER(ctx);
    if (!d_out || (nbytes && !d_iq)) return fail(ctx, GJ_ERR_INVALID, "null buffer");
    return launch_amp_stats(ctx, d_iq, nbytes, threshold, d_out);
}

int gj_onset_dev(gj_ctx* ctx, const uint8_t* d_iq, size_t nbytes, int noise_samples, int window, float factor,
                 gj_onset* d_out) {
    GJ_ENTER(ctx);
    if (!d_out || (nbytes && !d_iq)) return fail(ctx, GJ_ERR_INVALID, "null buffer");
    return launch_onset(ctx, d_iq, nbytes, noise_samples, window, factor, d_out);
}

int gj_stream_scan_dev(gj_ctx* ctx, const uint8_t* d_iq, size_t nbytes, size_t chunk_bytes, float eps, int flags,
                       float* d_power, float rssi_threshold, gj_amp_stats* d_amp, int noise_samples, int window,
                       float factor, gj_onset* d_onset) {
    GJ_ENTER(ctx);
    if (!d_power || !d_amp || !d_onset || (nbytes && !d_iq)) return fail(ctx, GJ_ERR_INVALID, "null buffer");
    if (chunk_bytes == 0) return fail(ctx, GJ_ERR_INVALID, "chunk_bytes must be > 0");
    return launch_stream_scan(ctx, d_iq, nbytes, chunk_bytes, eps, flags, d_power, rssi_threshold, d_amp,
                              noise_samples, window, factor, d_onset);
}

static int scan_extra_of(gj_ctx* ctx, const gj_scan_extra* e, ScanExtra& x) {
    if (!e) return GJ_OK;
    x.pct = e->pct;
    x.rise_db = e->rise_db;
    x.d_stats = e->d_stats;
    x.d_mask = e->d_mask;
    x.d_slot = e->d_slot;
    x.slice_samples = e->slice_samples;
    if (e->d_slot && e->slice_samples < 1) return fail(ctx, GJ_ERR_INVALID, "n_samples must be >= 1");
    if (e->d_slot && (reinterpret_cast<uintptr_t>(e->d_slot) & 15)) return fail(ctx, GJ_ERR_INVALID, "slot must be 16-byte aligned");
    return GJ_OK;
}

int gj_capture_scan_dev(gj_ctx* ctx, const uint8_t* d_iq, size_t nbytes, size_t chunk_bytes, float eps, int flags,
                        float* d_power, float rssi_threshold, gj_amp_stats* d_amp, int noise_samples, int window,
                        float factor, gj_onset* d_onset, const gj_scan_extra* extra) {
    GJ_ENTER(ctx);
    if (!d_power || !d_amp || !d_onset || (nbytes && !d_iq)) return fail(ctx, GJ_ERR_INVALID, "null buffer");
    if (chunk_bytes == 0) return fail(ctx, GJ_ERR_INVALID, "chunk_bytes must be > 0");
    ScanExtra x;
    const int rc = scan_extra_of(ctx, extra, x);
    if (rc) return rc;
    return launch_stream_scan(ctx, d_iq, nbytes, chunk_bytes, eps, flags, d_power, rssi_threshold, d_amp,
                              noise_samples, window, factor, d_onset, &x);
}

int gj_xcorr_lags_dev(gj_ctx* ctx, const uint8_t* const* d_iq, const size_t* nbytes, int n_ant, const int64_t* d_starts,
                      size_t n_samples, const int32_t* pairs, int n_pairs, int32_t* d_lags, float* d_peaks,
                      float* d_margins) {
    GJ_ENTER(ctx);
    if (!d_iq || !nbytes || !d_starts || !pairs || !d_lags || !d_peaks) return fail(ctx, GJ_ERR_INVALID, "null buffer");
    if (n_ant < 1 || n_ant > GJ_MAX_ANTENNAS) return fail(ctx, GJ_ERR_INVALID, "n_ant must be 1..%d", GJ_MAX_ANTENNAS);
    const int64_t* sp[GJ_MAX_ANTENNAS];
    for (int a = 0; a < n_ant; ++a) sp[a] = d_starts + a;
    return launch_xcorr(ctx, d_iq, nbytes, n_ant, sp, n_samples, pairs, n_pairs, d_lags, d_peaks, d_margins);
}

size_t gj_tdoa_slot_bytes(size_t n_samples) { return align_up(GJ_SLOT_HEADER + 2 * n_samples, 256); }

int gj_tdoa_slot_dev(gj_ctx* ctx, const uint8_t* d_iq, size_t nbytes, const int64_t* d_start, size_t n_samples,
                     uint8_t* d_slot) {
    GJ_ENTER(ctx);
    if (!d_iq || !d_start || !d_slot) return fail(ctx, GJ_ERR_INVALID, "null buffer");
    return launch_tdoa_slot(ctx, d_iq, nbytes, d_start, n_samples, d_slot);
}

int gj_xcorr_slots_dev(gj_ctx* ctx, const uint8_t* d_slots, size_t slot_stride, int n_ant, size_t n_samples,
                       const int32_t* pairs, int n_pairs, int32_t* d_lags, float* d_peaks, float* d_margins) {
    GJ_ENTER(ctx);
    if (!d_slots || !pairs || !d_lags || !d_peaks) return fail(ctx, GJ_ERR_INVALID, "null buffer");
    if (n_ant < 1 || n_ant > 4096) return fail(ctx, GJ_ERR_INVALID, "n_ant must be 1..4096 slots");
    if (n_pairs < 1) return fail(ctx, GJ_ERR_INVALID, "n_pairs must be >= 1");
    if (slot_stride < GJ_SLOT_HEADER + 2 * n_samples || (slot_stride & 15) || (reinterpret_cast<uintptr_t>(d_slots) & 15))
        return fail(ctx, GJ_ERR_INVALID, "slot stride %zu too small for %zu samples or not 16-byte aligned", slot_stride,
                    n_samples);
    // only the slots the pairs name are transformed: the pairs of one call may touch at most GJ_MAX_ANTENNAS of them
    int local_of[4096];
    for (int a = 0; a < n_ant; ++a) local_of[a] = -1;
    const uint8_t* ptrs[GJ_MAX_ANTENNAS];
    const int64_t* starts[GJ_MAX_ANTENNAS];
    size_t sizes[GJ_MAX_ANTENNAS];
    std::vector<int32_t> local_pairs((size_t)2 * n_pairs);
    int used = 0;
    for (int k = 0; k < 2 * n_pairs; ++k) {
        const int a = pairs[k];
        if (a < 0 || a >= n_ant) return fail(ctx, GJ_ERR_INVALID, "pair %d names slot %d of %d", k / 2, a, n_ant);
        if (local_of[a] < 0) {
            if (used == GJ_MAX_ANTENNAS) return fail(ctx, GJ_ERR_UNSUPPORTED, "the pairs of one call touch more than %d slots", GJ_MAX_ANTENNAS);
            const uint8_t* slot = d_slots + (size_t)a * slot_stride;
            ptrs[used] = slot + GJ_SLOT_HEADER;
            starts[used] = reinterpret_cast<const int64_t*>(slot);   // the flag word: 0 valid / -1 invalid
            sizes[used] = 2 * n_samples;
            local_of[a] = used++;
        }
        local_pairs[k] = local_of[a];
    }
    return launch_xcorr(ctx, ptrs, sizes, used, starts, n_samples, local_pairs.data(), n_pairs, d_lags, d_peaks, d_margins);
}

int gj_pack_result_dev(gj_ctx* ctx, size_t n_chunks, const float* d_power, const float* d_stats, const gj_amp_stats* d_amp,
                       const gj_onset* d_onset, const float* d_psd, size_t rows, int nperseg, int rank, int n_pairs,
                       int pair_capacity, const int32_t* d_pairs, const int32_t* d_lags, const float* d_peaks,
                       const float* d_margins, double* d_out) {
    GJ_ENTER(ctx);
    if (!d_power || !d_stats || !d_amp || !d_onset || !d_out || (rows && !d_psd))
        return fail(ctx, GJ_ERR_INVALID, "null buffer");
    if (n_pairs < 0 || n_pairs > pair_capacity) return fail(ctx, GJ_ERR_INVALID, "n_pairs %d exceeds the capacity %d", n_pairs, pair_capacity);
    if (n_pairs && (!d_pairs || !d_lags || !d_peaks || !d_margins)) return fail(ctx, GJ_ERR_INVALID, "null pair buffer");
    return launch_pack_result(ctx, n_chunks, d_power, d_stats, d_amp, d_onset, d_psd, rows, nperseg, rank, n_pairs,
                              pair_capacity, d_pairs, d_lags, d_peaks, d_margins, d_out);
}

// ---- one capture over several GPUs
size_t gj_amp_tile_count(size_t nbytes) { return amp_tile_count(nbytes); }

int gj_part_scan_dev(gj_ctx* ctx, const gj_part_view* part, size_t chunk_bytes, float eps, int flags, float* d_power,
                     float rssi_threshold, void* d_tiles, gj_amp_part* d_amp, int noise_samples, int window, float factor,
                     gj_onset* d_onset) {
    GJ_ENTER(ctx);
    if (!part || !part->d_buf || !d_power || !d_tiles || !d_amp || !d_onset) return fail(ctx, GJ_ERR_INVALID, "null buffer");
    return launch_part_scan(ctx, *part, chunk_bytes, eps, flags, d_power, rssi_threshold, d_tiles, d_amp, noise_samples,
                            window, factor, d_onset);
}

int gj_part_capture_scan_dev(gj_ctx* ctx, const gj_part_view* part, size_t chunk_bytes, float eps, int flags, float* d_power,
                             float rssi_threshold, void* d_tiles, gj_amp_part* d_amp, int noise_samples, int window, float factor,
                             gj_onset* d_onset, const gj_scan_extra* extra) {
    GJ_ENTER(ctx);
    if (!part || !part->d_buf || !d_power || !d_tiles || !d_amp || !d_onset) return fail(ctx, GJ_ERR_INVALID, "null buffer");
    if (part->buf_first_byte & 1) return fail(ctx, GJ_ERR_INVALID, "the buffer must start on a sample");
    ScanExtra x;
    const int rc = scan_extra_of(ctx, extra, x);
    if (rc) return rc;
    x.slot_buf_bytes = part->buf_bytes;   // a slice may run into the tail behind the own range
    return launch_part_scan(ctx, *part, chunk_bytes, eps, flags, d_power, rssi_threshold, d_tiles, d_amp, noise_samples,
                            window, factor, d_onset, &x);
}

static int part_own(gj_ctx* ctx, const gj_part_view* part, const uint8_t** own) {
    if (!part || !part->d_buf) return fail(ctx, GJ_ERR_INVALID, "null buffer");
    if (part->own_first_byte < part->buf_first_byte || part->own_first_byte + part->own_bytes > part->buf_first_byte + part->buf_bytes)
        return fail(ctx, GJ_ERR_INVALID, "the own range does not lie inside the buffer");
    *own = part->d_buf + (part->own_first_byte - part->buf_first_byte);
    return GJ_OK;
}

int gj_part_welch_dev(gj_ctx* ctx, const gj_part_view* part, size_t chunk_samples, int nperseg, double fs, int flags,
                      float* d_psd, float* d_psd_db) {
    GJ_ENTER(ctx);
    const uint8_t* own = nullptr;
    int rc = part_own(ctx, part, &own);
    if (rc) return rc;
    if (!d_psd) return fail(ctx, GJ_ERR_INVALID, "null buffer");
    if (chunk_samples == 0 || part->own_first_byte % (2 * chunk_samples))
        return fail(ctx, GJ_ERR_UNSUPPORTED, "a part must start on a PSD chunk boundary");
    if (part->own_first_byte + part->own_bytes != part->total_bytes && part->own_bytes % (2 * chunk_samples))
        return fail(ctx, GJ_ERR_UNSUPPORTED, "only the capture's last part may end inside a PSD chunk");
    return launch_welch(ctx, own, part->own_bytes, chunk_samples, nperseg, fs, flags, d_psd, d_psd_db, part->total_bytes);
}

size_t gj_part_welch_workspace(gj_ctx* ctx, const gj_part_view* part, size_t chunk_samples, int nperseg) {
    return (ctx && part) ? welch_workspace(ctx, part->own_bytes, chunk_samples, nperseg, part->total_bytes) : 0;
}

int gj_part_slot_dev(gj_ctx* ctx, const gj_part_view* part, const int64_t* d_start, size_t n_samples, uint8_t* d_slot) {
    GJ_ENTER(ctx);
    if (!part || !part->d_buf || !d_start || !d_slot) return fail(ctx, GJ_ERR_INVALID, "null buffer");
    if (part->buf_first_byte & 1) return fail(ctx, GJ_ERR_INVALID, "the buffer must start on a sample");
    return launch_tdoa_slot(ctx, part->d_buf, part->buf_bytes, d_start, n_samples, d_slot, (long long)(part->buf_first_byte / 2),
                            part->total_bytes / 2);
}

int gj_slots_pick_dev(gj_ctx* ctx, const uint8_t* d_slots, size_t slot_stride, const int32_t* d_offsets, const int32_t* d_members,
                      int n_groups, uint8_t* d_out) {
    GJ_ENTER(ctx);
    if (!d_slots || !d_offsets || !d_members || !d_out) return fail(ctx, GJ_ERR_INVALID, "null buffer");
    return launch_slots_pick(ctx, d_slots, slot_stride, d_offsets, d_members, n_groups, d_out);
}

int gj_amp_combine_dev(gj_ctx* ctx, const void* d_tiles, size_t n_tiles, const gj_amp_part* d_parts, int n_parts,
                       size_t total_bytes, gj_amp_stats* d_out) {
    GJ_ENTER(ctx);
    if (!d_tiles || !d_parts || !d_out) return fail(ctx, GJ_ERR_INVALID, "null buffer");
    return launch_amp_combine(ctx, d_tiles, n_tiles, d_parts, n_parts, total_bytes, d_out);
}

int gj_onset_combine_dev(gj_ctx* ctx, const gj_onset* d_parts, int n_parts, gj_onset* d_out) {
    GJ_ENTER(ctx);
    if (!d_parts || !d_out) return fail(ctx, GJ_ERR_INVALID, "null buffer");
    return launch_onset_combine(ctx, d_parts, n_parts, d_out);
}

size_t gj_part_result_len(size_t chunk_cap, size_t tile_cap, size_t rows_cap, int nperseg, int pair_cap) {
    if (nperseg < 0 || pair_cap < 0) return 0;
    return GJ_RESULT_HEADER + chunk_cap + 2 * tile_cap + (size_t)GJ_RESULT_PAIR_FIELDS * (size_t)pair_cap +
           (rows_cap * (size_t)nperseg + 1) / 2;
}

int gj_pack_part_dev(gj_ctx* ctx, const gj_part_pack* a, double* d_out) {
    GJ_ENTER(ctx);
    if (!a || !d_out || !a->d_power || !a->d_amp || !a->d_onset || !a->d_tiles || (a->rows && !a->d_psd))
        return fail(ctx, GJ_ERR_INVALID, "null buffer");
    if (a->n_chunks > a->chunk_cap || a->n_tiles > a->tile_cap || a->rows > a->rows_cap || a->n_pairs > a->pair_cap || a->n_pairs < 0)
        return fail(ctx, GJ_ERR_CAPACITY, "a part's arrays exceed the capacities of the vector");
    if (a->n_pairs && (!a->d_pairs || !a->d_lags || !a->d_peaks || !a->d_margins)) return fail(ctx, GJ_ERR_INVALID, "null pair buffer");
    return launch_pack_part(ctx, *a, d_out);
}

int gj_combine_plan_create(gj_ctx* ctx, const gj_combine_copy* copies, int n_copies, const gj_combine_capture* captures,
                           int n_captures, size_t rows_bytes, const void* d_arena, size_t arena_bytes, int nperseg, float pct,
                           float rise_db, const int32_t* d_pairs, const int32_t* d_lags, const float* d_peaks,
                           const float* d_margins, gj_combine_plan** out) {
    if (!ctx || !out) return GJ_ERR_INVALID;
    *out = nullptr;
    if (!copies || !captures || !d_arena) return fail(ctx, GJ_ERR_INVALID, "null argument");
    if (n_copies < 1 || n_copies > 65535 || n_captures < 1 || n_captures > 1024)
        return fail(ctx, GJ_ERR_INVALID, "%d copies, %d captures", n_copies, n_captures);
    if (nperseg < 16 || nperseg > 4096 || (nperseg & (nperseg - 1))) return fail(ctx, GJ_ERR_UNSUPPORTED, "nperseg %d", nperseg);
    // no context state is touched: validation on the host, two device allocations and two synchronous copies -- done
    // WITHOUT the context lock (a synchronous copy is a host-side wait)
    NoCancel nc;
    (void)hipSetDevice(ctx->device);
    return combine_plan_create(ctx, copies, n_copies, captures, n_captures, rows_bytes, d_arena, arena_bytes, nperseg, pct, rise_db,
                               d_pairs, d_lags, d_peaks, d_margins, out);
}

int gj_pack_results_dev(gj_ctx* ctx, const gj_combine_capture* captures, int n_captures, int nperseg, const int32_t* d_pairs,
                        const int32_t* d_lags, const float* d_peaks, const float* d_margins) {
    GJ_ENTER(ctx);
    if (!captures) return fail(ctx, GJ_ERR_INVALID, "null argument");
    if (nperseg < 16 || nperseg > 4096 || (nperseg & (nperseg - 1))) return fail(ctx, GJ_ERR_UNSUPPORTED, "nperseg %d", nperseg);
    return launch_pack_results(ctx, captures, n_captures, nperseg, d_pairs, d_lags, d_peaks, d_margins);
}

// the host-side validation of gj_combine_plan_create alone: touches no GPU, needs no context
int gj_combine_plan_check(const gj_combine_copy* copies, int n_copies, const gj_combine_capture* captures, int n_captures,
                          size_t rows_bytes, const void* d_arena, size_t arena_bytes, int nperseg, int have_pairs) {
    if (!copies || !captures || !d_arena) return GJ_ERR_INVALID;
    if (n_copies < 1 || n_copies > 65535 || n_captures < 1 || n_captures > 1024) return GJ_ERR_INVALID;
    if (nperseg < 16 || nperseg > 4096 || (nperseg & (nperseg - 1))) return GJ_ERR_UNSUPPORTED;
    return combine_plan_check(nullptr, copies, n_copies, captures, n_captures, rows_bytes, d_arena, arena_bytes, nperseg,
                              have_pairs != 0, nullptr, nullptr, nullptr);
}

int gj_split_combine_dev(gj_ctx* ctx, const gj_combine_plan* plan, const double* d_rows) {
    GJ_ENTER(ctx);
    if (!plan || !d_rows) return fail(ctx, GJ_ERR_INVALID, "null argument");
    return launch_split_combine(ctx, plan, d_rows);
}

int gj_combine_plan_destroy(gj_ctx* ctx, gj_combine_plan* plan) {
    if (!plan) return GJ_OK;
    if (!ctx) return GJ_ERR_INVALID;
    (void)wait_stream(ctx, current_stream(ctx));   // queued launches read the plan's device arrays
    NoCancel nc;
    (void)hipSetDevice(ctx->device);
    combine_plan_destroy(plan);                    // hipFree may wait for the device: not under the context lock
    return GJ_OK;
}

int gj_acq_search_dev(gj_ctx* ctx, const uint8_t* d_iq, size_t nbytes, size_t first_sample, int nsamp, int intg,
                      const int16_t* d_codes, int n_prn, const uint8_t* d_phase, int n_freq, int nsampchip, double ctime,
                      float threshold, gj_acq_result* d_out, double* d_power) {
    GJ_ENTER(ctx);
    if (!d_iq || !d_codes || !d_phase || !d_out) return fail(ctx, GJ_ERR_INVALID, "null buffer");
    return launch_acq_search(ctx, d_iq, nbytes, first_sample, nsamp, intg, d_codes, n_prn, d_phase, n_freq, nsampchip, ctime,
                             threshold, d_out, d_power);
}

size_t gj_acq_workspace(gj_ctx*, int nsamp, int n_freq, int n_prn, int intg, int with_power) {
    if (nsamp <= 0 || n_freq <= 0 || n_prn <= 0 || intg <= 0) return 0;
    return acq_workspace(nsamp, n_freq, n_prn, intg, with_power == 0);
}

int gj_synth_u8_dev(gj_ctx* ctx, const gj_synth_params* params, int64_t first_sample, size_t n_samples, uint8_t* d_out) {
    GJ_ENTER(ctx);
    if (!params || (n_samples && !d_out)) return fail(ctx, GJ_ERR_INVALID, "null buffer");
    return launch_synth(ctx, *params, first_sample, n_samples, d_out);
}

}   // extern "C"

