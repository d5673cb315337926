// C-ABI of libgpsjam_hip.so (see include/gpsjam.h): context, device memory, stopwatch and the
// host-buffer entry points that stage numpy arrays to HBM around the *_dev kernels.
#include <cmath>
#include <new>
#include <vector>

#include <atomic>
#include <thread>
#include <vector>

#include "gj_common.h"

namespace gj {

constexpr int kWindowFloats = 2 * 4096;   // periodic Hann tables for N = 16..4096 at offset N-16

int ensure_workspace(gj_ctx* ctx, size_t bytes) {
    if (bytes <= ctx->ws_bytes) return GJ_OK;
    bytes = align_up(bytes + bytes / 8, 1 << 20);
    // the old arena may still be in use by kernels queued on the stream
    GJ_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (ctx->ws) (void)hipFree(ctx->ws);
    ctx->ws = nullptr;
    ctx->ws_bytes = 0;
    void* p = nullptr;
    if (hipMalloc(&p, bytes) != hipSuccess) return fail(ctx, GJ_ERR_NOMEM, "workspace of %zu bytes", bytes);
    ctx->ws = static_cast<unsigned char*>(p);
    ctx->ws_bytes = bytes;
    return GJ_OK;
}

int ensure_stage(gj_ctx* ctx, size_t bytes) {
    if (bytes <= ctx->stage_bytes) return GJ_OK;
    bytes = align_up(bytes, 1 << 20);
    GJ_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (ctx->stage) (void)hipFree(ctx->stage);
    ctx->stage = nullptr;
    ctx->stage_bytes = 0;
    void* p = nullptr;
    if (hipMalloc(&p, bytes) != hipSuccess) return fail(ctx, GJ_ERR_NOMEM, "staging buffer of %zu bytes", bytes);
    ctx->stage = static_cast<unsigned char*>(p);
    ctx->stage_bytes = bytes;
    return GJ_OK;
}

static float* g_window_dummy = nullptr;
const float* window_table(gj_ctx* ctx, int n) {
    (void)g_window_dummy;
    return reinterpret_cast<const float*>(ctx->d_twiddle + kTwiddleTable) + (n - 16);
}

// small device scratch for results of the host-buffer entry points (lives behind the tables)
static unsigned char* result_scratch(gj_ctx* ctx) {
    return reinterpret_cast<unsigned char*>(ctx->d_twiddle + kTwiddleTable) + kWindowFloats * sizeof(float);
}
constexpr size_t kResultScratch = 4096;

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

const char* gj_last_error(gj_ctx* ctx) { return ctx ? ctx->last_error : "null context"; }

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
    auto bail = [&](int code) {
        gj_destroy(ctx);
        return code;
    };
    if (hipSetDevice(device_id) != hipSuccess) return bail(GJ_ERR_HIP);
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device_id) == hipSuccess) ctx->num_cus = prop.multiProcessorCount;
    if (hipStreamCreateWithFlags(&ctx->own_stream, hipStreamNonBlocking) != hipSuccess) return bail(GJ_ERR_HIP);
    ctx->stream = ctx->own_stream;
    if (hipEventCreate(&ctx->ev_start) != hipSuccess || hipEventCreate(&ctx->ev_stop) != hipSuccess) return bail(GJ_ERR_HIP);
    // constant tables: W_4096^m, periodic Hann windows, result scratch
    const size_t bytes = kTwiddleTable * sizeof(cf) + kWindowFloats * sizeof(float) + kResultScratch;
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
    *out = ctx;
    return GJ_OK;
}

int gj_destroy(gj_ctx* ctx) {
    if (!ctx) return GJ_OK;
    (void)hipSetDevice(ctx->device);
    if (ctx->own_stream) (void)hipStreamSynchronize(ctx->own_stream);
    if (ctx->ws) (void)hipFree(ctx->ws);
    if (ctx->stage) (void)hipFree(ctx->stage);
    for (int k = 0; k < gj_ctx::kPinBufs; ++k) {
        if (ctx->pin[k]) (void)hipHostFree(ctx->pin[k]);
        if (ctx->pin_ev[k]) (void)hipEventDestroy(ctx->pin_ev[k]);
    }
    if (ctx->d_twiddle) (void)hipFree(ctx->d_twiddle);
    if (ctx->ev_start) (void)hipEventDestroy(ctx->ev_start);
    if (ctx->ev_stop) (void)hipEventDestroy(ctx->ev_stop);
    if (ctx->own_stream) (void)hipStreamDestroy(ctx->own_stream);
    delete ctx;
    return GJ_OK;
}

int gj_set_stream(gj_ctx* ctx, void* hip_stream, int external) {
    if (!ctx) return GJ_ERR_INVALID;
    Guard g(ctx);
    ctx->stream = external ? static_cast<hipStream_t>(hip_stream) : ctx->own_stream;
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

int gj_get_unpack(gj_ctx* ctx, double* offset, double* scale) {
    if (!ctx) return GJ_ERR_INVALID;
    Guard g(ctx);
    if (offset) *offset = 0.5 * ctx->off2;
    if (scale) *scale = ctx->scale;
    return GJ_OK;
}

int gj_synchronize(gj_ctx* ctx) {
    if (!ctx) return GJ_ERR_INVALID;
    Guard g(ctx);
    GJ_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return GJ_OK;
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

int gj_reserve(gj_ctx* ctx, size_t workspace_bytes) {
    if (!ctx) return GJ_ERR_INVALID;
    Guard g(ctx);
    return ensure_workspace(ctx, workspace_bytes);
}

int gj_malloc(gj_ctx* ctx, size_t bytes, void** dptr) {
    if (!ctx || !dptr) return GJ_ERR_INVALID;
    Guard g(ctx);
    *dptr = nullptr;
    if (bytes == 0) return GJ_OK;
    if (hipMalloc(dptr, bytes) != hipSuccess) return fail(ctx, GJ_ERR_NOMEM, "hipMalloc(%zu)", bytes);
    return GJ_OK;
}

int gj_free(gj_ctx* ctx, void* dptr) {
    if (!ctx) return GJ_ERR_INVALID;
    Guard g(ctx);
    if (dptr) {
        GJ_HIP(ctx, hipStreamSynchronize(ctx->stream));
        GJ_HIP(ctx, hipFree(dptr));
    }
    return GJ_OK;
}

int gj_memcpy_h2d(gj_ctx* ctx, void* dst_dev, const void* src_host, size_t bytes) {
    if (!ctx || (bytes && (!dst_dev || !src_host))) return GJ_ERR_INVALID;
    Guard g(ctx);
    if (bytes) {
        GJ_HIP(ctx, hipMemcpyAsync(dst_dev, src_host, bytes, hipMemcpyHostToDevice, ctx->stream));
        GJ_HIP(ctx, hipStreamSynchronize(ctx->stream));
    }
    return GJ_OK;
}

int gj_memcpy_d2h(gj_ctx* ctx, void* dst_host, const void* src_dev, size_t bytes) {
    if (!ctx || (bytes && (!dst_host || !src_dev))) return GJ_ERR_INVALID;
    Guard g(ctx);
    if (bytes) {
        GJ_HIP(ctx, hipMemcpyAsync(dst_host, src_dev, bytes, hipMemcpyDeviceToHost, ctx->stream));
        GJ_HIP(ctx, hipStreamSynchronize(ctx->stream));
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
    Guard g(ctx);
    GJ_HIP(ctx, hipEventRecord(ctx->ev_stop, ctx->stream));
    GJ_HIP(ctx, hipEventSynchronize(ctx->ev_stop));
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

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

// ---------------------------------------------------------------- host-buffer entry points
// Stage the capture into HBM (grow-only staging arena), run the same kernels, copy the small
// results back.  kernel_ms excludes the copies.
// Large captures go through pinned 16 MiB bounce buffers (the host copy of piece k+1 overlaps
// the DMA of piece k: 29 GB/s with one fill thread against 21 GB/s for a pageable hipMemcpy of
// 1 GiB on the MI355X box, tools/h2d_bench.hip); small ones take the plain path.
constexpr size_t kPinBytes = 16u << 20;
constexpr size_t kPinThreshold = 64u << 20;

// Host buffer -> ctx->stage.  Large pageable buffers go through pinned bounce buffers: kFillThreads
// host threads (8) each copy their share of 16-MiB pieces into their own pair of pinned buffers and
// queue the DMA (one memcpy thread tops out at ~31 GB/s end to end, below what the link carries;
// pread out of the page cache needs the extra threads more than memcpy does).
constexpr int kMaxFillThreads = gj_ctx::kPinBufs / 2;

// fill threads of one staged copy: GPSJAM_FILL_THREADS (1..16), default 8
static int fill_threads() {
    static const int n = [] {
        const char* e = getenv("GPSJAM_FILL_THREADS");
        int v = e ? atoi(e) : 8;
        if (v < 1) v = 1;
        if (v > kMaxFillThreads) v = kMaxFillThreads;
        return v;
    }();
    return n;
}

// `fill(dst, off, len)` puts bytes [off, off+len) of the source into a pinned buffer: memcpy from a
// numpy array, or pread from a capture file (then the file goes page cache -> pinned -> HBM with no
// pageable copy in between)
template <typename Fill>
static int staged_copy(gj_ctx* ctx, unsigned char* d_dst, size_t nbytes, Fill&& fill) {
    if (nbytes == 0) return GJ_OK;
    const size_t npieces = (nbytes + kPinBytes - 1) / kPinBytes;
    const int nthreads = (int)(npieces < (size_t)fill_threads() ? npieces : (size_t)fill_threads());
    for (int k = 0; k < 2 * nthreads; ++k) {   // two bounce buffers per fill thread, made on first use
        if (!ctx->pin[k]) GJ_HIP(ctx, hipHostMalloc(&ctx->pin[k], kPinBytes, hipHostMallocDefault));
        if (!ctx->pin_ev[k]) GJ_HIP(ctx, hipEventCreateWithFlags(&ctx->pin_ev[k], hipEventDisableTiming));
    }
    std::atomic<int> failed{0};
    auto worker = [&](int t) {
        if (hipSetDevice(ctx->device) != hipSuccess) { failed.store(1); return; }
        size_t mine = 0;
        for (size_t piece = (size_t)t; piece < npieces; piece += (size_t)nthreads, ++mine) {
            const size_t off = piece * kPinBytes;
            const size_t len = (nbytes - off < kPinBytes) ? nbytes - off : kPinBytes;
            const int b = 2 * t + (int)(mine & 1);
            if (mine >= 2 && hipEventSynchronize(ctx->pin_ev[b]) != hipSuccess) { failed.store(1); return; }
            if (!fill(static_cast<unsigned char*>(ctx->pin[b]), off, len)) { failed.store(2); return; }
            if (hipMemcpyAsync(d_dst + off, ctx->pin[b], len, hipMemcpyHostToDevice, ctx->stream) != hipSuccess ||
                hipEventRecord(ctx->pin_ev[b], ctx->stream) != hipSuccess) { failed.store(1); return; }
        }
        // the bounce buffers are reused by the next call: the tail pieces must have left them
        for (int k = 0; k < 2; ++k)
            if (mine > (size_t)k && hipEventSynchronize(ctx->pin_ev[2 * t + k]) != hipSuccess) failed.store(1);
    };
    std::vector<std::thread> pool;
    for (int t = 1; t < nthreads; ++t) pool.emplace_back(worker, t);
    worker(0);
    for (auto& th : pool) th.join();
    if (failed.load() == 2) return fail(ctx, GJ_ERR_INVALID, "reading the capture failed");
    if (failed.load()) return fail(ctx, GJ_ERR_HIP, "host-to-device staging failed");
    return GJ_OK;
}

static int stage_in(gj_ctx* ctx, const uint8_t* host, size_t nbytes, size_t offset = 0) {
    int rc = ensure_stage(ctx, offset + align_up(nbytes, 256) + 256);
    if (rc) return rc;
    if (nbytes == 0) return GJ_OK;
    if (nbytes < kPinThreshold) {
        GJ_HIP(ctx, hipMemcpyAsync(ctx->stage + offset, host, nbytes, hipMemcpyHostToDevice, ctx->stream));
        return GJ_OK;
    }
    return staged_copy(ctx, ctx->stage + offset, nbytes, [host](unsigned char* dst, size_t off, size_t len) {
        memcpy(dst, host + off, len);
        return true;
    });
}

extern "C" {

// ---------------------------------------------------------------- resident captures
// One upload per capture, then any number of *_dev calls on it (gpsjam.Capture): the host-buffer
// entry points below re-stage their input on every call (21 ms per GiB of PCIe against 0.2-1.3 ms
// of kernel time).
int gj_upload(gj_ctx* ctx, const uint8_t* host, size_t nbytes, void** dptr) {
    GJ_ENTER(ctx);
    if (!dptr || (nbytes && !host)) return fail(ctx, GJ_ERR_INVALID, "null buffer");
    *dptr = nullptr;
    void* p = nullptr;
    if (hipMalloc(&p, align_up(nbytes, 256) + 256) != hipSuccess) return fail(ctx, GJ_ERR_NOMEM, "hipMalloc(%zu)", nbytes);
    int rc = GJ_OK;
    if (nbytes && nbytes < kPinThreshold) {
        if (hipMemcpyAsync(p, host, nbytes, hipMemcpyHostToDevice, ctx->stream) != hipSuccess)
            rc = fail(ctx, GJ_ERR_HIP, "hipMemcpyAsync failed");
    } else {
        rc = staged_copy(ctx, static_cast<unsigned char*>(p), nbytes, [host](unsigned char* dst, size_t off, size_t len) {
            memcpy(dst, host + off, len);
            return true;
        });
    }
    if (!rc && hipStreamSynchronize(ctx->stream) != hipSuccess) rc = fail(ctx, GJ_ERR_HIP, "upload failed");
    if (rc) {
        (void)hipFree(p);
        return rc;
    }
    *dptr = p;
    return GJ_OK;
}

// The reference's ingest (np.fromfile / f.read, worker.py:209-217, triangulateRSSI.py:29) as file ->
// pinned bounce buffers (pread, eight threads) -> HBM.  max_bytes = 0: to the end of the file.
int gj_upload_file(gj_ctx* ctx, const char* path, size_t offset, size_t max_bytes, void** dptr, size_t* nbytes_out) {
    GJ_ENTER(ctx);
    if (!path || !dptr || !nbytes_out) return fail(ctx, GJ_ERR_INVALID, "null argument");
    *dptr = nullptr;
    *nbytes_out = 0;
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
    int rc = GJ_OK;
    static const bool want_pread = [] {
        const char* e = getenv("GPSJAM_FILE_READ");
        return e && strcmp(e, "pread") == 0;
    }();
    const size_t pg = (size_t)sysconf(_SC_PAGESIZE);
    const size_t map_off = offset / pg * pg, lead = offset - map_off;
    void* m = (nbytes && !want_pread) ? mmap(nullptr, nbytes + lead, PROT_READ, MAP_PRIVATE, fd, (off_t)map_off) : MAP_FAILED;
    if (m != MAP_FAILED) {
        (void)madvise(m, nbytes + lead, MADV_SEQUENTIAL);
        const unsigned char* src = static_cast<const unsigned char*>(m) + lead;
        rc = staged_copy(ctx, static_cast<unsigned char*>(p), nbytes, [src](unsigned char* dst, size_t off, size_t len) {
            memcpy(dst, src + off, len);
            return true;
        });
        (void)munmap(m, nbytes + lead);
    } else {
        (void)posix_fadvise(fd, 0, 0, POSIX_FADV_NOREUSE);   // regular file systems (Linux >= 6.3): no LRU promotion on read
        rc = staged_copy(ctx, static_cast<unsigned char*>(p), nbytes, [fd, offset](unsigned char* dst, size_t off, size_t len) {
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
    if (!rc && hipStreamSynchronize(ctx->stream) != hipSuccess) rc = fail(ctx, GJ_ERR_HIP, "upload failed");
    if (rc) {
        (void)hipFree(p);
        return rc;
    }
    *dptr = p;
    *nbytes_out = nbytes;
    return GJ_OK;
}

static int fetch(gj_ctx* ctx, void* host, const void* dev, size_t bytes) {
    if (bytes) GJ_HIP(ctx, hipMemcpyAsync(host, dev, bytes, hipMemcpyDeviceToHost, ctx->stream));
    GJ_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return GJ_OK;
}

#define GJ_TIMED(ctx, ms, stmt)                                                        \
    do {                                                                               \
        GJ_HIP(ctx, hipEventRecord((ctx)->ev_start, (ctx)->stream));                   \
        int rc__ = (stmt);                                                             \
        if (rc__) return rc__;                                                         \
        GJ_HIP(ctx, hipEventRecord((ctx)->ev_stop, (ctx)->stream));                    \
        GJ_HIP(ctx, hipEventSynchronize((ctx)->ev_stop));                              \
        float t__ = 0.f;                                                               \
        GJ_HIP(ctx, hipEventElapsedTime(&t__, (ctx)->ev_start, (ctx)->ev_stop));       \
        if (ms) *(ms) = t__;                                                           \
    } while (0)

int gj_chunk_power_u8(gj_ctx* ctx, const uint8_t* iq, size_t nbytes, size_t chunk_bytes, float eps, int flags,
                      float* power, size_t power_cap, size_t* n_out, float* kernel_ms) {
    GJ_ENTER(ctx);
    if (chunk_bytes == 0) return fail(ctx, GJ_ERR_INVALID, "chunk_bytes must be > 0");
    const size_t n = gj_chunk_count(nbytes, chunk_bytes);
    if (n_out) *n_out = n;
    if (kernel_ms) *kernel_ms = 0.f;
    if (n == 0) return GJ_OK;
    if (!iq || !power) return fail(ctx, GJ_ERR_INVALID, "null buffer");
    if (power_cap < n) return fail(ctx, GJ_ERR_CAPACITY, "power buffer holds %zu, need %zu", power_cap, n);
    const size_t off_out = align_up(nbytes, 256);
    int rc = ensure_stage(ctx, off_out + n * sizeof(float) + 256);
    if (rc) return rc;
    rc = stage_in(ctx, iq, nbytes);
    if (rc) return rc;
    float* d_power = reinterpret_cast<float*>(ctx->stage + off_out);
    GJ_TIMED(ctx, kernel_ms, launch_chunk_power(ctx, ctx->stage, nbytes, chunk_bytes, eps, flags, d_power));
    return fetch(ctx, power, d_power, n * sizeof(float));
}

int gj_welch_u8(gj_ctx* ctx, const uint8_t* iq, size_t nbytes, size_t chunk_samples, int nperseg, double fs, int flags,
                float* psd, float* psd_db, size_t cap_floats, size_t* rows_out, float* kernel_ms) {
    GJ_ENTER(ctx);
    const size_t rows = gj_welch_rows(nbytes, chunk_samples, nperseg);
    if (rows_out) *rows_out = rows;
    if (kernel_ms) *kernel_ms = 0.f;
    if (nperseg < 16 || nperseg > 4096 || (nperseg & (nperseg - 1)))
        return fail(ctx, GJ_ERR_UNSUPPORTED, "nperseg must be a power of two in [16, 4096]");
    if (rows == 0) return GJ_OK;
    if (!iq || !psd) return fail(ctx, GJ_ERR_INVALID, "null buffer");
    const size_t nfl = rows * (size_t)nperseg;
    if (cap_floats < nfl) return fail(ctx, GJ_ERR_CAPACITY, "psd buffer holds %zu floats, need %zu", cap_floats, nfl);
    const size_t off_out = align_up(nbytes, 256);
    int rc = ensure_stage(ctx, off_out + 2 * nfl * sizeof(float) + 256);
    if (rc) return rc;
    rc = stage_in(ctx, iq, nbytes);
    if (rc) return rc;
    float* d_psd = reinterpret_cast<float*>(ctx->stage + off_out);
    float* d_db = psd_db ? d_psd + nfl : nullptr;
    GJ_TIMED(ctx, kernel_ms, launch_welch(ctx, ctx->stage, nbytes, chunk_samples, nperseg, fs, flags, d_psd, d_db));
    if (psd_db) GJ_HIP(ctx, hipMemcpyAsync(psd_db, d_db, nfl * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
    return fetch(ctx, psd, d_psd, nfl * sizeof(float));
}

int gj_amp_stats_u8(gj_ctx* ctx, const uint8_t* iq, size_t nbytes, float threshold, gj_amp_stats* out, float* kernel_ms) {
    GJ_ENTER(ctx);
    if (!out || (nbytes && !iq)) return fail(ctx, GJ_ERR_INVALID, "null buffer");
    int rc = stage_in(ctx, iq, nbytes);
    if (rc) return rc;
    gj_amp_stats* d_out = reinterpret_cast<gj_amp_stats*>(result_scratch(ctx));
    GJ_TIMED(ctx, kernel_ms, launch_amp_stats(ctx, ctx->stage, nbytes, threshold, d_out));
    return fetch(ctx, out, d_out, sizeof(gj_amp_stats));
}

int gj_onset_u8(gj_ctx* ctx, const uint8_t* iq, size_t nbytes, int noise_samples, int window, float factor, gj_onset* out,
                float* kernel_ms) {
    GJ_ENTER(ctx);
    if (!out || (nbytes && !iq)) return fail(ctx, GJ_ERR_INVALID, "null buffer");
    int rc = stage_in(ctx, iq, nbytes);
    if (rc) return rc;
    gj_onset* d_out = reinterpret_cast<gj_onset*>(result_scratch(ctx));
    GJ_TIMED(ctx, kernel_ms, launch_onset(ctx, ctx->stage, nbytes, noise_samples, window, factor, d_out));
    return fetch(ctx, out, d_out, sizeof(gj_onset));
}

int gj_xcorr_lags_u8(gj_ctx* ctx, const uint8_t* const* slices, int n_ant, size_t n_samples, const int32_t* pairs,
                     int n_pairs, int32_t* lags, float* peaks, float* margins, float* kernel_ms) {
    GJ_ENTER(ctx);
    if (!slices || !pairs || !lags || !peaks) return fail(ctx, GJ_ERR_INVALID, "null buffer");
    if (n_ant < 1 || n_ant > GJ_MAX_ANTENNAS) return fail(ctx, GJ_ERR_INVALID, "n_ant must be 1..%d", GJ_MAX_ANTENNAS);
    if (n_pairs < 1 || (size_t)n_pairs * 12 + 256 > kResultScratch) return fail(ctx, GJ_ERR_INVALID, "bad n_pairs");
    const size_t slot = align_up(2 * n_samples, 256);
    int rc = ensure_stage(ctx, slot * n_ant + 256);
    if (rc) return rc;
    const uint8_t* d_ptrs[GJ_MAX_ANTENNAS];
    size_t nbytes[GJ_MAX_ANTENNAS];
    for (int a = 0; a < n_ant; ++a) {
        if (!slices[a]) return fail(ctx, GJ_ERR_INVALID, "null slice %d", a);
        GJ_HIP(ctx, hipMemcpyAsync(ctx->stage + slot * a, slices[a], 2 * n_samples, hipMemcpyHostToDevice, ctx->stream));
        d_ptrs[a] = ctx->stage + slot * a;
        nbytes[a] = 2 * n_samples;
    }
    unsigned char* sc = result_scratch(ctx);
    int64_t* d_starts = reinterpret_cast<int64_t*>(sc);   // zeros: slices start at their first sample
    int32_t* d_lags = reinterpret_cast<int32_t*>(sc + 128);
    float* d_peaks = reinterpret_cast<float*>(sc + 128 + 4 * (size_t)n_pairs);
    float* d_margins = reinterpret_cast<float*>(sc + 128 + 8 * (size_t)n_pairs);
    GJ_HIP(ctx, hipMemsetAsync(d_starts, 0, 128, ctx->stream));
    const int64_t* sp[GJ_MAX_ANTENNAS];
    for (int a = 0; a < n_ant; ++a) sp[a] = d_starts + a;
    GJ_TIMED(ctx, kernel_ms,
             launch_xcorr(ctx, d_ptrs, nbytes, n_ant, sp, n_samples, pairs, n_pairs, d_lags, d_peaks, d_margins));
    GJ_HIP(ctx, hipMemcpyAsync(peaks, d_peaks, 4 * (size_t)n_pairs, hipMemcpyDeviceToHost, ctx->stream));
    if (margins) GJ_HIP(ctx, hipMemcpyAsync(margins, d_margins, 4 * (size_t)n_pairs, hipMemcpyDeviceToHost, ctx->stream));
    return fetch(ctx, lags, d_lags, 4 * (size_t)n_pairs);
}

}   // extern "C"
