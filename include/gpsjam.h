/* gpsjam.h -- C-ABI of libgpsjam_hip.so: the MI355X (gfx950) implementation of the
 * jamming-detection DSP path of mfkiwl/GPS-JAMMING.
 *
 * The reference has no FFI for this path (it is numpy/scipy called from Python); the
 * entry points below are what its Python call sites bind through ctypes instead of
 * numpy/scipy.  Each one names the reference code it replaces (path:line relative to the
 * reference root).  INTEGRATION.md shows the ctypes stubs a maintainer adds.
 *
 * Conventions
 *  - every function returns GJ_OK (0) or a negative gj_status; gj_strerror() gives text,
 *    gj_last_error() the detailed message of the last failure on a context;
 *  - plain pointers and sizes only; the caller owns every buffer;
 *  - one opaque gj_ctx per GPU; different contexts are independent; the library may be entered
 *    from several host threads (the GUI's QThread, its HTTP handler thread, its triangulation
 *    thread).  The context's internal mutex is held only while work is ENQUEUED, never across a
 *    host-side wait, a file read or a staged copy, and it is a robust mutex: a caller that is
 *    killed inside a call (QThread.terminate(), GpsJammerApp/app/ui_mainwindow.py:818-826) does
 *    not block the next one.  Every "*_u8" / upload call works in buffers of its own (a "lane":
 *    device staging, pinned bounce buffers, events), so concurrent calls on one context never
 *    share a result area; kernels of all calls are ordered on the context's one stream.  The lane of
 *    a killed caller is taken back -- drained and emptied -- at the next lane check-out or
 *    gj_debug_counters call after its thread has ended (see "Diagnostics" below for exactly what);
 *  - "*_dev" functions take DEVICE pointers, enqueue on the context's stream and return
 *    without synchronising (results land in device memory; no host synchronisation, no
 *    allocation when the workspace has been reserved) -- this is what bench.py times;
 *  - "*_u8" functions take HOST buffers (numpy arrays), stage them to HBM, run the same
 *    kernels, copy the small results back and return synchronously, reporting the
 *    kernel-only time measured with HIP events on the context's stream.  Their input may also
 *    be a DEVICE pointer (a resident capture from gj_upload / gj_upload_file / gj_malloc, or
 *    an address inside one): it is then used in place and nothing is staged;
 *  - samples are interleaved unsigned 8-bit I,Q,I,Q,... (RTL-SDR, README.md:95); one
 *    "sample" = one I/Q pair = 2 bytes.
 */
#ifndef GPSJAM_H
#define GPSJAM_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GJ_VERSION 150 /* 0.1.5: no new entry points.  gj_amp_stats_* / gj_onset_* run the fused pass + tail (any alignment accepted; same bits as gj_capture_scan_dev); gj_set_stream orders the new stream behind the old one; gj_ingest_files leaves the context's fill-thread setting alone */

typedef struct gj_ctx gj_ctx;

typedef enum gj_status {
    GJ_OK = 0,
    GJ_ERR_INVALID = -1,     /* bad argument */
    GJ_ERR_HIP = -2,         /* a HIP runtime call failed (see gj_last_error) */
    GJ_ERR_NOMEM = -3,       /* device or host allocation failed */
    GJ_ERR_NODEVICE = -4,    /* no such GPU */
    GJ_ERR_UNSUPPORTED = -5, /* size / parameter outside what the kernels implement */
    GJ_ERR_CAPACITY = -6     /* caller's output buffer too small */
} gj_status;

/* ---------------------------------------------------------------- context ---------- */
int gj_version(void);
const char* gj_strerror(int status);
const char* gj_last_error(gj_ctx* ctx); /* of the CALLING THREAD's last failure (kept per thread) */
int gj_device_count(int* count);
int gj_create(int device_id, gj_ctx** out);
/* gj_destroy: idempotent on NULL.  Takes down the communicators made on the context (their handles stay valid for
 * gj_comm_destroy); a collective another thread is still enqueueing on one of them returns first.  No OTHER call on
 * the context may be in progress or start while it is being destroyed. */
int gj_destroy(gj_ctx* ctx);
/* external != 0: run on the caller's HIP stream `hip_stream` (e.g.
 * torch.cuda.current_stream().cuda_stream; NULL then means the legacy default stream);
 * external == 0: back to the context's own non-blocking stream.
 * ONE stream at a time per context: a context's kernels hand results between workgroups through arrival counters that
 * belong to the context, and its workspace is one arena, so two launches of one context must never be in flight on two
 * streams at once.  The switch itself sees to that -- the new stream is ordered behind everything the context has queued
 * on the old one (an event + a wait; skipped while either stream is being captured into a graph) -- and a caller who
 * wants two streams to run side by side uses two contexts (gpsjam/sharded.py, split.py, local.py do). */
int gj_set_stream(gj_ctx* ctx, void* hip_stream, int external);
int gj_synchronize(gj_ctx* ctx);
/* Unpack convention of the uint8 samples on this context: sample = (u8 - offset) * scale.  The
 * reference uses three: u - 127.5 (worker.py:222, checkIfJamming.py:15,
 * triangulateTDOA.py:34), (u - 127.5)/127.5 (triangulateRSSI.py:30, widmo_plot.py:39-40) and
 * (int8)(u - 128) (GpsJammerApp/backend/sdrrcv.c:104-106).  The default (127.5, 1/127.5)
 * reproduces the first two: `offset` is honoured by every kernel, `scale` by the kernels whose
 * reference normalises (K2 Welch, K3 amplitude); K1, K4, K5 work in LSB units like their
 * references.  offset must be a multiple of 0.5 in [0, 255] (sums stay exact integers).
 * gj_set_unpack(ctx, 128, 1/128.) gives the gnssdec convention; the acquisition search always
 * uses 128 as its reference does. */
int gj_set_unpack(gj_ctx* ctx, double offset, double scale);
int gj_get_unpack(gj_ctx* ctx, double* offset, double* scale);
int gj_device_info(gj_ctx* ctx, char* name, size_t name_cap, int* compute_units,
                   uint64_t* hbm_bytes);
/* Which physical GPU this context runs on, as text: "pci=<domain:bus:device.function> uuid=<hex> hip=<index>".
 * Ranks of a multi-GPU run exchange these so that the collecting rank can tell N GPUs from N ranks on one
 * (bench.py's `devices` list). */
int gj_device_identity(gj_ctx* ctx, char* out, size_t cap);
/* Pre-size the internal workspace so that later *_dev calls allocate nothing. */
int gj_reserve(gj_ctx* ctx, size_t workspace_bytes);
/* Diagnostics.  The hook is called, with NO internal lock held, right before every host-side wait of
 * the library (site: 1 event, 2 stream, 3 staged copy, 4 waiting for a free lane); tests park or end
 * a thread there to show that an abandoned caller blocks nobody.  Site 5 is the exception: it is
 * inside gj_debug_counters WITH the lock held, so that a test can end a thread as the mutex's owner
 * and watch the next caller recover it.  NULL removes the hook.  The counters:
 * lanes made, lanes in use, lanes taken back from callers whose thread had ended, and how often the
 * context mutex was found with a dead owner.
 *
 * What happens to a caller that is killed inside a call, and when: the calling thread holds its lane's
 * owner token -- a robust mutex -- from check-out to check-in.  The kernel marks that token "owner
 * died" while the thread exits, before the thread can be joined, so there is no interval in which a dead
 * caller looks alive and no thread id a later thread could inherit.  EVERY later lane check-out (any
 * "*_u8" / upload / ingest call, from any thread) and every gj_debug_counters call first sweeps all
 * lanes in use; a lane whose owner has ended is taken over by the sweeping thread, which -- holding
 * no lock -- waits for whatever the dead caller had queued (the context's streams, the lane's copy
 * stream and bounce-buffer events), frees everything the lane had grown to (device staging, pinned
 * buffers, events, copy stream, ingest workspace) and returns it to the pool.  Not covered: the
 * resident capture a killed gj_upload / gj_ingest had allocated but not yet handed out (that device
 * memory stays allocated until gj_destroy of the process' GPU context), and a kill in the middle of a
 * staged copy whose helper threads are still running. */
int gj_debug_set_wait_hook(gj_ctx* ctx, void (*hook)(void* arg, int site), void* arg);
int gj_debug_counters(gj_ctx* ctx, int* lanes, int* lanes_busy, int* lanes_reclaimed, int* owner_deaths);
/* Fault injection for tests.  GJ_INJECT_OWNER_ALIVE: the next `count` owner probes of a lane in use are
 * skipped, i.e. answer "its caller is alive" whatever the truth -- the wrong answer that round 4's
 * sampled probe gave inside the thread-exit window.  A test uses it to show that a lane missed once is
 * still taken back at the next check-out or counters call. */
#define GJ_INJECT_OWNER_ALIVE 1
int gj_debug_inject(gj_ctx* ctx, int what, int count);

/* Stream-overlap probe.  Keeps the context's stream busy for `milliseconds` (0..100; more is refused) with ONE wave
 * that spins on the 100-MHz real-time counter: the chip stays free, the stream -- and the hardware queue the runtime
 * mapped it to -- does not.  A host that needs two streams to run side by side uses it to find out whether they share
 * a hardware queue (the HIP runtime deals streams over GPU_MAX_HW_QUEUES queues, four by default, and two streams on
 * one queue run one after the other): gpsjam/streams.py, which the pipelines of gpsjam/sharded.py, split.py and
 * local.py call at construction.  Enqueues only; never synchronises. */
int gj_probe_busy_dev(gj_ctx* ctx, float milliseconds);

/* device memory for callers that do not bring their own allocator (torch) */
int gj_malloc(gj_ctx* ctx, size_t bytes, void** dptr);
int gj_free(gj_ctx* ctx, void* dptr);
int gj_memcpy_h2d(gj_ctx* ctx, void* dst_dev, const void* src_host, size_t bytes);
int gj_memcpy_d2h(gj_ctx* ctx, void* dst_host, const void* src_dev, size_t bytes);

/* Resident captures: one upload, then any number of *_dev calls on the returned device pointer
 * (free with gj_free).  The "*_u8" entry points below stage their input on EVERY call; a caller
 * that runs scan + PSD + RSSI on one file uploads it once with these.  gj_upload_file is the
 * reference's ingest (np.fromfile / f.read: GpsJammerApp/app/worker.py:209-217,
 * skrypty/triangulateRSSI.py:29, skrypty/triangulateTDOA.py:33) as file -> pinned bounce buffers
 * -> HBM; max_bytes = 0 reads to the end of the file. */
int gj_upload(gj_ctx* ctx, const uint8_t* host, size_t nbytes, void** dptr);
int gj_upload_file(gj_ctx* ctx, const char* path, size_t offset, size_t max_bytes, void** dptr,
                   size_t* nbytes_out);

/* Host threads that fill the pinned bounce buffers of ONE staged copy (gj_upload*, gj_ingest_*): 0 = by capture size (two
 * to eight: four for the reference's 10-s captures, eight from 64 MiB up -- starting and joining eight threads costs a
 * 41-MB capture more than they carry), 1..16 = that many.  For a host that brings in several files at once from threads
 * of its own (one lane per call) and wants them to share the cores.  GPSJAM_FILL_THREADS in the environment overrides
 * both. */
int gj_set_fill_threads(gj_ctx* ctx, int n);

/* HIP-event stopwatch on the context's stream (what bench.py's roofline uses) */
int gj_timer_start(gj_ctx* ctx);
int gj_timer_stop(gj_ctx* ctx, float* elapsed_ms); /* synchronises on the stop event */

/* ------------------------------------------------- K1: per-chunk power scan --------- */
/* Replaces the loop of GPSAnalysisThread.precalculate_power_profile
 * (GpsJammerApp/app/worker.py:216-230) and analyze_chunk_power
 * (GpsJammerApp/app/checkIfJamming.py:7-20, with GJ_CP_ODD_CHUNK_ZERO, eps = 0).
 * power[c] = mean over the I/Q pairs of chunk c of (I-127.5)^2+(Q-127.5)^2, + eps.
 * The ragged tail chunk is included; a trailing odd byte is dropped (worker.py:226);
 * a chunk without a complete pair gives NaN (numpy mean of empty) unless the flag below
 * is set.  The chunk sum is accumulated as an exact integer and rounded once. */
#define GJ_CP_ODD_CHUNK_ZERO 1 /* odd-sized or empty chunk -> 0.0 (checkIfJamming.py:12-13) */
size_t gj_chunk_count(size_t nbytes, size_t chunk_bytes);
int gj_chunk_power_dev(gj_ctx* ctx, const uint8_t* d_iq, size_t nbytes, size_t chunk_bytes,
                       float eps, int flags, float* d_power /* [gj_chunk_count] */);
int gj_chunk_power_u8(gj_ctx* ctx, const uint8_t* iq, size_t nbytes, size_t chunk_bytes,
                      float eps, int flags, float* power, size_t power_cap, size_t* n_out,
                      float* kernel_ms);

/* Noise floor + threshold (worker.py:241-248): baseline = numpy.percentile(power, pct)
 * with numpy's float32 'linear' rule, 1.0 if <= 0; threshold = baseline * 10^(rise_db/10);
 * mask[c] = power[c] > threshold.  d_stats = {baseline, threshold, count_above}. */
int gj_power_threshold_dev(gj_ctx* ctx, const float* d_power, size_t n, float pct,
                           float rise_db, float* d_stats /* [3] */, uint8_t* d_mask /* [n] or NULL */);

/* ------------------------------------------------- K2: Welch PSD waterfall ---------- */
/* Replaces the chunk body of analyze_full_file (skrypty/widmo_plot.py:26-54) including
 * its scipy.signal.welch(x, fs, nperseg=N, return_onesided=False) call (:48):
 * per chunk of chunk_samples I/Q pairs: x = ((I-127.5) + j(Q-127.5))/127.5, periodic Hann,
 * 50 % overlap, per-segment mean removal, |FFT|^2 averaged over the segments,
 * scale 1/(fs*sum(w^2)); a trailing partial chunk is kept when it holds at least
 * nperseg samples (widmo_plot.py:31).  nperseg: power of two, 16..4096;
 * chunk_samples >= nperseg and chunk_samples + nperseg < 2^31 (a chunk is addressed with 32-bit
 * byte offsets; GJ_ERR_UNSUPPORTED otherwise -- the capture itself may be any length).
 * Output rows are float32[nperseg]; GJ_WELCH_SHIFT applies numpy.fft.fftshift (:51);
 * d_psd_db (optional) receives 10*log10(psd + 1e-15) (:52). */
#define GJ_WELCH_SHIFT 1
size_t gj_welch_rows(size_t nbytes, size_t chunk_samples, int nperseg);
int gj_welch_dev(gj_ctx* ctx, const uint8_t* d_iq, size_t nbytes, size_t chunk_samples,
                 int nperseg, double fs, int flags, float* d_psd /* [rows*nperseg] */,
                 float* d_psd_db /* [rows*nperseg] or NULL */);
int gj_welch_u8(gj_ctx* ctx, const uint8_t* iq, size_t nbytes, size_t chunk_samples,
                int nperseg, double fs, int flags, float* psd, float* psd_db, size_t cap_floats,
                size_t* rows_out, float* kernel_ms);
/* n captures of ONE length (nbytes_each) in one transform launch + one finalize launch: the reference's deployment is
 * three antenna recordings of one length (GpsJammerApp/app/worker.py:97-101), and at 10-s captures a step's time is its
 * launch gaps and grid tails, not its bytes.  Each capture is planned, cut into workgroups and summed exactly as by
 * gj_welch_dev on its own, so d_psd[a] receives the same bits; d_psd[a] must be 16-byte aligned. */
int gj_welch_batch_dev(gj_ctx* ctx, const uint8_t* const* d_iq, int n_captures, size_t nbytes_each, size_t chunk_samples,
                       int nperseg, double fs, int flags, float* const* d_psd /* [n_captures] x [rows*nperseg] */);
/* Measurement: gj_welch_dev with HIP events around the transform launch and around the finalize launch (the sum over
 * the per-workgroup spectra, scaling, fftshift, dB), on the context's stream; synchronises and reports both.  bench.py
 * times K2 at nperseg 4096 and 1024 with it, interleaved, so that the two figures of one line come from the same
 * minute of the same box. */
int gj_welch_timed_dev(gj_ctx* ctx, const uint8_t* d_iq, size_t nbytes, size_t chunk_samples, int nperseg, double fs,
                       int flags, float* d_psd, float* d_psd_db, float* kernel_ms, float* finalize_ms);
/* workspace bytes gj_welch_dev needs for this input (for gj_reserve) */
size_t gj_welch_workspace(gj_ctx* ctx, size_t nbytes, size_t chunk_samples, int nperseg);

/* raw-byte histogram of every `stride`-th byte (widmo_plot.py:35,85: stride 100,
 * 256 bins).  Strided per chunk exactly like raw_chunk[::100]. */
int gj_byte_histogram_dev(gj_ctx* ctx, const uint8_t* d_iq, size_t nbytes, size_t chunk_samples,
                          int nperseg, int stride, uint64_t* d_hist /* [256] */);

/* ------------------------------------------------- K3: amplitude statistics --------- */
/* Replaces read_iq_data + np.abs + find_change_point + np.mean(amp[idx:])
 * (skrypty/triangulateRSSI.py:29-31,37-40,65-68): amp = |((I-127.5) + j(Q-127.5))/127.5|,
 * first index with amp > threshold, mean of amp from that index to the end. */
typedef struct gj_amp_stats {
    int64_t first_index; /* -1: nothing above the threshold (or empty input) */
    uint64_t count;      /* samples from first_index to the end */
    double sum;          /* sum of amp over those samples */
    float mean;          /* (float)(sum / count) -- the reference's avg_amplitude */
    float reserved;
} gj_amp_stats;
int gj_amp_stats_dev(gj_ctx* ctx, const uint8_t* d_iq, size_t nbytes, float threshold,
                     gj_amp_stats* d_out);
int gj_amp_stats_u8(gj_ctx* ctx, const uint8_t* iq, size_t nbytes, float threshold,
                    gj_amp_stats* out, float* kernel_ms);

/* ------------------------------------------------- K4: interference onset ----------- */
/* Replaces find_interference_start (skrypty/triangulateTDOA.py:37-49) on
 * z = (I-127.5) + j(Q-127.5): noise = mean |z|^2 over the first noise_samples
 * (1e-9 if 0), moving average of |z|^2 over `window` samples ('valid'), first index
 * above noise*factor, + window/2; -1 when none or the stream is shorter than
 * noise_samples + window.  Window sums are exact integers. window <= 8192. */
typedef struct gj_onset {
    int64_t start_index; /* -1 = not found */
    float noise_power;
    float threshold;
    /* Decision margins, relative to the threshold.  The window sums here are exact integers and
     * the noise mean is rounded once; the reference sums float32 |z|^2 (pairwise for the noise
     * mean, float64 inside np.convolve), so its threshold and moving averages differ from these in
     * the last ulps (~1e-7 relative).  See guard_index below for the exact statement of when the
     * index is the reference's by construction. */
    float margin_hit;    /* (moving average at the crossing - threshold) / threshold; 0 if not found */
    float margin_before; /* (threshold - largest moving average in front of the crossing) / threshold;
                          * positions the screening pass proved quiet enter with their upper bound, so
                          * this can under-state the true gap, never over-state it.  Not found: over the
                          * whole capture. */
    /* First index (+ window/2, like start_index) whose moving average exceeds threshold * (1 - 1e-6),
     * -1 if none.  In front of it every moving average is below the reference's threshold whatever
     * the reference's float32 rounding did, so the reference's own first crossing cannot lie before
     * it.  guard_index == start_index and margin_hit >= 1e-6: the index is the reference's by
     * construction.  Otherwise only the positions from guard_index on need the reference's
     * arithmetic (skrypty/triangulateTDOA.py evaluates it there, on the host). */
    int64_t guard_index;
} gj_onset;
int gj_onset_dev(gj_ctx* ctx, const uint8_t* d_iq, size_t nbytes, int noise_samples, int window,
                 float factor, gj_onset* d_out);
int gj_onset_u8(gj_ctx* ctx, const uint8_t* iq, size_t nbytes, int noise_samples, int window,
                float factor, gj_onset* out, float* kernel_ms);

/* ------------------------------------------------- fused stream scan ---------------- */
/* K1 + K3 + K4 in ONE pass over the capture (the three are pure streaming reductions over
 * the same bytes): identical results to gj_chunk_power_dev, gj_amp_stats_dev and gj_onset_dev
 * called one after the other.  The single pass is used when chunk_bytes is a multiple of
 * 65536 and the capture is 16-byte aligned; otherwise K1 runs alone and the pass delivers K3 + K4.
 * gj_amp_stats_dev and gj_onset_dev are themselves this pass (with the outputs nobody asked for
 * dropped): neither depends on the chunk size, so a capture gives the same K3 / K4 bits through
 * every entry point.  A capture that is not 16-byte aligned is first copied into the workspace
 * (one device-to-device copy; any alignment is accepted). */
int gj_stream_scan_dev(gj_ctx* ctx, const uint8_t* d_iq, size_t nbytes,
                       size_t chunk_bytes, float eps, int flags, float* d_power,
                       float rssi_threshold, gj_amp_stats* d_amp,
                       int noise_samples, int window, float factor, gj_onset* d_onset);

/* The whole per-capture side chain in TWO launches: the fused pass above, then one "tail" launch whose workgroups take
 * roles -- noise-floor threshold of the power map (gj_power_threshold_dev: worker.py:241-248), amplitude totals
 * (triangulateRSSI.py:65-68), K4's screening + exact scan + record (triangulateTDOA.py:37-49) and the TDOA slot cut at that
 * onset (gj_tdoa_slot_dev with d_start = &d_onset->start_index) -- where gj_stream_scan_dev + gj_power_threshold_dev +
 * gj_tdoa_slot_dev were eight dependent launches.  At the reference's capture sizes (10 s = 41 MB) that chain, not the
 * bytes, was the step time.  Every result is the same bits as from the separate calls, except gj_onset.margin_before,
 * which is a bound (see gj_onset) and is now a function of the capture alone.  `extra` may be NULL (then this IS
 * gj_stream_scan_dev); a NULL d_stats / d_slot leaves that role out. */
typedef struct gj_scan_extra {
    float pct, rise_db;   /* noise floor: numpy.percentile(power, pct) * 10^(rise_db/10) */
    float* d_stats;       /* [3] {baseline, threshold, count_above} or NULL */
    uint8_t* d_mask;      /* [n_chunks] or NULL */
    size_t slice_samples; /* TDOA slot of slice_samples I/Q pairs ... */
    uint8_t* d_slot;      /* ... written here (gj_tdoa_slot_bytes, 16-byte aligned) or NULL */
} gj_scan_extra;
int gj_capture_scan_dev(gj_ctx* ctx, const uint8_t* d_iq, size_t nbytes, size_t chunk_bytes, float eps, int flags,
                        float* d_power, float rssi_threshold, gj_amp_stats* d_amp, int noise_samples, int window,
                        float factor, gj_onset* d_onset, const gj_scan_extra* extra);

/* ------------------------------------------------- overlapped ingest ----------------- */
/* Overlapped ingest: upload a capture AND analyse it, with the kernels running on the pieces (1-16 MiB, by capture size) that have
 * landed in HBM while the rest is still on its way (the reference reads, then computes: worker.py:209-230,
 * triangulateRSSI.py:29-31, widmo_plot.py:26-54).  What is computed: with chunk_bytes != 0 the fused scan (K1 power
 * map, K3 amplitude statistics, K4 onset -- gj_stream_scan_dev), with nperseg != 0 the Welch waterfall (K2 --
 * gj_welch_dev).  Results are bit-identical to gj_upload* followed by those calls.  The capture stays resident:
 * *dptr is a device pointer like gj_upload's (free with gj_free), usable with every "*_dev" and "*_u8" entry point. */
typedef struct gj_ingest_plan {
    size_t chunk_bytes; /* K1 power chunks; 0: no scan (K3 and K4 are skipped too) */
    float eps;
    int power_flags; /* GJ_CP_* */
    float rssi_threshold; /* K3 */
    int noise_samples, window; /* K4 */
    float factor;
    size_t chunk_samples; /* K2 */
    int nperseg; /* 0: no PSD */
    int welch_flags; /* GJ_WELCH_* */
    double fs;
} gj_ingest_plan;
typedef struct gj_ingest_result {
    size_t nbytes, n_chunks, rows;
    gj_amp_stats amp;
    gj_onset onset;
    float upload_ms; /* wall clock until the last piece had been queued and its bounce buffer released */
    float total_ms;  /* wall clock of the whole call: results on the host */
} gj_ingest_result;
int gj_ingest_u8(gj_ctx* ctx, const uint8_t* host, size_t nbytes, const gj_ingest_plan* plan, float* power,
                 size_t power_cap, float* psd, float* psd_db, size_t psd_cap_floats, gj_ingest_result* result,
                 void** dptr);
int gj_ingest_file(gj_ctx* ctx, const char* path, size_t offset, size_t max_bytes, const gj_ingest_plan* plan,
                   float* power, size_t power_cap, float* psd, float* psd_db, size_t psd_cap_floats,
                   gj_ingest_result* result, void** dptr);

/* The recordings of a deployment brought in together: gj_ingest_file for every job, side by side (one host thread of the
 * library's own and one lane per file; the fill threads per file are lowered for the duration unless gj_set_fill_threads
 * fixed them).  The reference reads them one after the other (skrypty/triangulateRSSI.py:160-174, worker.py:586-600).
 * Every job gets its own status, result and device pointer (free each with gj_free); the return value is the first
 * non-zero status, with that file's message in gj_last_error.  Results are those of gj_ingest_file per file. */
typedef struct gj_ingest_job {
    const char* path;
    size_t offset, max_bytes; /* as gj_ingest_file */
    float* power;
    size_t power_cap;
    float* psd;
    float* psd_db; /* or NULL */
    size_t psd_cap_floats;
    gj_ingest_result result; /* out */
    void* dptr;              /* out */
    int status;              /* out */
} gj_ingest_job;
int gj_ingest_files(gj_ctx* ctx, gj_ingest_job* jobs, int n_jobs, const gj_ingest_plan* plan);

/* ------------------------------------------------- K5: TDOA cross-correlation ------- */
/* Replaces signal.correlate(sig1, sig0, 'full') + argmax|.| - (N-1)
 * (skrypty/triangulateTDOA.py:80-89) for every requested antenna pair.
 * d_iq[a] is antenna a's capture in HBM, nbytes[a] its length; the slice of antenna a is
 * the n_samples I/Q pairs starting at d_starts[a] (DEVICE array, e.g. written by
 * gj_onset_dev; a negative or out-of-range start marks the antenna invalid).
 * pairs = {i0,j0,i1,j1,...}: lag of antenna j relative to antenna i
 * (= correlate(slice_j, slice_i)), positive when j is delayed.  FFT length is the next
 * power of two >= 2*n_samples-1 (four-step, in HBM/L2).  Outputs (device):
 * lags[p] (INT32_MIN if an antenna of the pair is invalid), peaks[p] = max |c|. */
#define GJ_MAX_ANTENNAS 16
#define GJ_LAG_INVALID INT32_MIN
/* margins[p] (optional, may be NULL) = 1 - |c|_runner-up / |c|_peak, the relative gap between the
 * peak and the largest |c| at any other lag: the arg-max is taken over values that carry the
 * rounding of a complex64 FFT (~1e-6 of the peak, here as in scipy), so a margin of that order
 * means the reference's own choice between the two lags is decided by its rounding. */
int gj_xcorr_lags_dev(gj_ctx* ctx, const uint8_t* const* d_iq, const size_t* nbytes, int n_ant,
                      const int64_t* d_starts, size_t n_samples, const int32_t* pairs,
                      int n_pairs, int32_t* d_lags, float* d_peaks, float* d_margins);
int gj_xcorr_lags_u8(gj_ctx* ctx, const uint8_t* const* slices, int n_ant, size_t n_samples,
                     const int32_t* pairs, int n_pairs, int32_t* lags, float* peaks, float* margins,
                     float* kernel_ms);
size_t gj_xcorr_workspace(gj_ctx* ctx, int n_ant, size_t n_samples, int n_pairs);

/* TDOA slot = what one capture contributes to a multi-antenna solve, as ONE message:
 *   [int64 flag: 0 valid / -1 invalid][int64 start sample][2*n_samples bytes of I/Q], padded to a
 * multiple of 256 bytes (gj_tdoa_slot_bytes).  gj_tdoa_slot_dev cuts the n_samples I/Q pairs that
 * start at *d_start (DEVICE scalar, e.g. &gj_onset.start_index) out of a capture; an un-found
 * onset or a slice that runs off the end marks the slot invalid (the reference aborts there,
 * skrypty/triangulateTDOA.py:67-77).  gj_xcorr_slots_dev solves pairs over an array of n_ant
 * slots (slot a at d_slots + a*slot_stride), e.g. the receive buffer of gj_comm_allgather_dev:
 * pairs (i, j) of skrypty/triangulateTDOA.py:80-89 generalised to n_ant antennas.  Only the
 * slots the pairs name are transformed (at most GJ_MAX_ANTENNAS per call), so the pairs of a
 * many-antenna solve can be dealt over the ranks (gpsjam/sharded.py pairs_of_rank). */
#define GJ_SLOT_HEADER 16
size_t gj_tdoa_slot_bytes(size_t n_samples);
int gj_tdoa_slot_dev(gj_ctx* ctx, const uint8_t* d_iq, size_t nbytes, const int64_t* d_start,
                     size_t n_samples, uint8_t* d_slot);
int gj_xcorr_slots_dev(gj_ctx* ctx, const uint8_t* d_slots, size_t slot_stride, int n_ant,
                       size_t n_samples, const int32_t* pairs, int n_pairs, int32_t* d_lags,
                       float* d_peaks, float* d_margins);

/* ------------------------------------------------- per-stream result vector ---------- */
/* What one rank sends to rank 0 (gpsjam/sharded.py):
 * double[40 + n_chunks + nperseg + 5*pair_capacity] =
 * header {  0 n_chunks, 1 baseline, 2 threshold, 3 n_above, 4 amp.first_index, 5 amp.count, 6 amp.mean,
 *           7 onset.start_index, 8 lag vs antenna 0 (0 on rank 0, GJ_LAG_INVALID elsewhere: the receiver
 *           fills it in from the pair table), 9 0, 10 onset.noise_power, 11 rows, 12 nperseg, 13 rank,
 *          14 n_pairs, 15 pair_capacity,
 *          16 onset.margin_hit, 17 onset.margin_before, 18 onset.guard_index, 19 onset.threshold -- the
 *             decision margins of K4 travel with the result, so the receiver knows when an onset was
 *             decided inside the rounding band (skrypty/triangulateTDOA.py:37-49),
 *          20 antenna, 21 part, 22 parts (1 = the stream is a whole capture), 23 first chunk, 24 first row,
 *          25 first sample of the part, 26 amp.sum, 27 amp tail, 28 tiles, 29 first tile, 30-31 reserved,
 *          32-35 the gj_onset record as it is (32 bytes), 36-39 the gj_amp_stats (gj_amp_part for a
 *          part) record as it is },
 * the float32 power map, the mean over the rows of the PSD waterfall, and the pairs THIS stream
 * solved as {i, j, lag, peak, margin} each (d_pairs = {i0,j0,i1,j1,...} and the outputs of
 * gj_xcorr_slots_dev, all DEVICE arrays), zero-padded to pair_capacity -- packed by one kernel
 * from device-resident outputs (no host synchronisation). */
#define GJ_RESULT_HEADER 40
#define GJ_RESULT_PAIR_FIELDS 5
int gj_pack_result_dev(gj_ctx* ctx, size_t n_chunks, const float* d_power, const float* d_stats,
                       const gj_amp_stats* d_amp, const gj_onset* d_onset, const float* d_psd, size_t rows,
                       int nperseg, int rank, int n_pairs, int pair_capacity, const int32_t* d_pairs,
                       const int32_t* d_lags, const float* d_peaks, const float* d_margins, double* d_out);

/* gj_pack_result_dev for up to GJ_MAX_ANTENNAS captures in ONE launch.  `captures` is a HOST array of gj_combine_capture
 * (declared below) of which the fields n_chunks, rows, antenna (-> header field 13), n_pairs, pair_cap, d_power, d_stats,
 * d_amp, d_onset, d_psd and d_out are read; the pair arrays are shared (the capture with n_pairs > 0 carries them). */
struct gj_combine_capture;
int gj_pack_results_dev(gj_ctx* ctx, const struct gj_combine_capture* captures, int n_captures, int nperseg,
                        const int32_t* d_pairs, const int32_t* d_lags, const float* d_peaks, const float* d_margins);

/* ------------------------------------------------- one capture over several GPUs ------ */
/* "fewer files than GPUs => split one file into contiguous chunk ranges aligned to
 * 65 536 B / 1-s chunks".  The reference's deployment has THREE antennas (GpsJammerApp/app/worker.py:97-101,
 * 586-600; skrypty/triangulateRSSI.py:147-154); on an 8-GPU node each capture is cut into parts, one per GPU.
 * A part OWNS the capture bytes [own_first_byte, own_first_byte + own_bytes): a whole number of power chunks and
 * of PSD chunks (only the capture's last part may end ragged).  Its device buffer additionally holds
 *   - a HALO in front (whole 64-KiB tiles, >= window - 1 samples; none for the first part), so that K4 evaluates
 *     every window that ends inside the own range: the parts' position ranges then tile the capture,
 *   - a TAIL behind (>= the TDOA slice), so that a slice that starts in the own range can be cut here,
 * and every part but the first brings the capture's first 2*noise_samples bytes (d_noise) for K4's threshold --
 * each rank reads those few hundred KB itself, so K1-K4 need NO collective.  Results are in CAPTURE coordinates
 * and bit-identical to the unsplit run by construction: chunk powers and PSD rows are per-chunk quantities (the
 * Welch workgroup split is planned for the whole capture), amplitude sums travel as per-tile sums and are added
 * by the same code in the same order (gj_amp_combine_dev), window sums are exact integers. */
typedef struct gj_part_view {
    const uint8_t* d_buf;  /* DEVICE: capture bytes [buf_first_byte, buf_first_byte + buf_bytes), 16-byte aligned */
    size_t buf_bytes;
    size_t buf_first_byte;
    size_t own_first_byte; /* multiple of chunk_bytes and of 2*chunk_samples; own_first_byte - buf_first_byte = halo */
    size_t own_bytes;
    size_t total_bytes;    /* of the whole capture */
    const uint8_t* d_noise; /* DEVICE: the capture's first 2*noise_samples bytes (may be NULL for a first part
                             * whose own range holds them) */
} gj_part_view;
typedef struct gj_amp_part {
    int64_t first_index; /* capture coordinates; -1: nothing above the threshold in this part */
    uint64_t count;      /* own samples from first_index to the end of the own range */
    double sum;          /* of the amplitudes over those samples (this part alone) */
    double tail;         /* of the amplitudes from first_index to the end of ITS 64-KiB tile */
} gj_amp_part;
/* 64-KiB amplitude tiles of a range of nbytes (16 bytes each: double sum, int64 first) */
size_t gj_amp_tile_count(size_t nbytes);
/* K1 + K3 + K4 of one part in one pass (gj_stream_scan_dev for a part): d_power[own chunks], d_tiles[own tiles],
 * *d_amp, *d_onset (capture coordinates; duplicates of a crossing inside the halo are harmless: the combining
 * rank takes the smallest index). */
int gj_part_scan_dev(gj_ctx* ctx, const gj_part_view* part, size_t chunk_bytes, float eps, int flags, float* d_power,
                     float rssi_threshold, void* d_tiles, gj_amp_part* d_amp, int noise_samples, int window,
                     float factor, gj_onset* d_onset);
/* gj_part_scan_dev + gj_part_slot_dev (at &d_onset->start_index) in the same two launches as gj_capture_scan_dev */
int gj_part_capture_scan_dev(gj_ctx* ctx, const gj_part_view* part, size_t chunk_bytes, float eps, int flags, float* d_power,
                             float rssi_threshold, void* d_tiles, gj_amp_part* d_amp, int noise_samples, int window,
                             float factor, gj_onset* d_onset, const gj_scan_extra* extra);
/* K2 of the own range: rows [own_first_byte / (2 chunk_samples), ...) of the capture's waterfall */
int gj_part_welch_dev(gj_ctx* ctx, const gj_part_view* part, size_t chunk_samples, int nperseg, double fs, int flags,
                      float* d_psd, float* d_psd_db);
size_t gj_part_welch_workspace(gj_ctx* ctx, const gj_part_view* part, size_t chunk_samples, int nperseg);
/* TDOA slot cut from a part's buffer at the CAPTURE index *d_start (flag -2: valid in the capture but not held here) */
int gj_part_slot_dev(gj_ctx* ctx, const gj_part_view* part, const int64_t* d_start, size_t n_samples, uint8_t* d_slot);
/* Slots of the parts -> one slot per capture: group g = the slots d_members[d_offsets[g]] ...
 * d_members[d_offsets[g+1] - 1] (DEVICE int arrays, d_offsets has n_groups + 1 entries); its slot is the one cut
 * at the smallest start >= 0 (none: an invalid slot with start -1). */
int gj_slots_pick_dev(gj_ctx* ctx, const uint8_t* d_slots, size_t slot_stride, const int32_t* d_offsets,
                      const int32_t* d_members, int n_groups, uint8_t* d_out);
/* On the combining rank: the capture's amplitude statistics from ALL its tile sums (in tile order) and the parts'
 * first hits; its onset from the parts' onsets. */
int gj_amp_combine_dev(gj_ctx* ctx, const void* d_tiles, size_t n_tiles, const gj_amp_part* d_parts, int n_parts,
                       size_t total_bytes, gj_amp_stats* d_out);
int gj_onset_combine_dev(gj_ctx* ctx, const gj_onset* d_parts, int n_parts, gj_onset* d_out);
/* What a part sends to the combining rank: double[gj_part_result_len] = the 40-field header of
 * gj_pack_result_dev (part fields filled, records 32-39 = gj_onset, gj_amp_part), chunk_cap chunk powers,
 * tile_cap x (sum, first) tile records, pair_cap x {i, j, lag, peak, margin}, rows_cap x nperseg PSD values as
 * float32 (two per double slot).  The capacities are the largest over all parts, so every vector has one length. */
typedef struct gj_part_pack {
    int32_t rank, antenna, part, parts;
    uint64_t first_chunk, n_chunks, chunk_cap;
    uint64_t first_row, rows, rows_cap;
    uint64_t first_tile, n_tiles, tile_cap;
    int64_t first_sample;
    int32_t nperseg, n_pairs, pair_cap, reserved;
    const float* d_power;
    const gj_amp_part* d_amp;
    const gj_onset* d_onset;
    const void* d_tiles;
    const float* d_psd;
    const int32_t* d_pairs;
    const int32_t* d_lags;
    const float* d_peaks;
    const float* d_margins;
} gj_part_pack;
size_t gj_part_result_len(size_t chunk_cap, size_t tile_cap, size_t rows_cap, int nperseg, int pair_cap);
int gj_pack_part_dev(gj_ctx* ctx, const gj_part_pack* args, double* d_out);

/* The combining rank, every capture at once.  gj_amp_combine_dev / gj_onset_combine_dev / gj_power_threshold_dev /
 * gj_pack_result_dev finish ONE capture and want its arrays contiguous; with three antennas cut into ten parts that is
 * a chain of some forty small launches and copies on the combining rank -- the critical path once a rank's own K2 has
 * shrunk to 1/8.  gj_split_combine_dev does the same work in THREE launches whatever the number of antennas:
 *   1. assemble: every copy of `copies` (part vector -> capture-order array) in one grid;
 *   2. statistics: one workgroup per capture for the noise-floor threshold, one for amplitude totals + onset;
 *   3. pack: one grid row per capture (the layout of gj_pack_result_dev).
 * The kernels' bodies are those of the one-capture entry points (same code, same order, same bits).  The copy list and
 * the capture descriptors are static for a deployment: gj_combine_plan_create checks on the HOST that every copy
 * reads inside the gathered vectors (rows_bytes) and that every destination / array lies inside the caller's arena
 * (one device allocation holding all assembled arrays and result vectors), then keeps device copies of both. */
typedef struct gj_combine_copy {
    uint64_t src_byte;   /* offset of the first source element in the gathered part vectors */
    uint64_t dst;        /* DEVICE address of the first destination element (inside the arena) */
    uint64_t count;      /* elements */
    uint32_t src_stride; /* bytes between source elements (8 for a run of doubles, 40 for one field of the pair block) */
    uint32_t kind;       /* GJ_COPY_* */
} gj_combine_copy;
#define GJ_COPY_F64_F32 0 /* double -> float32 (chunk powers, pair peaks / margins) */
#define GJ_COPY_F64 1     /* 8-byte copy (amplitude tile records, gj_onset / gj_amp_part records) */
#define GJ_COPY_F32 2     /* 4-byte copy (PSD rows) */
#define GJ_COPY_F64_I32 3 /* double -> int32 (pair lags) */
typedef struct gj_combine_capture {
    uint64_t n_chunks, rows, n_tiles, total_bytes;
    int32_t n_parts, antenna;
    int32_t n_pairs, pair_cap; /* pairs this capture's vector carries (all of them on antenna 0, none elsewhere) */
    float* d_power;            /* [n_chunks]        assembled */
    float* d_stats;            /* [3]               written by step 2 */
    void* d_tiles;             /* [n_tiles] x 16 B  assembled */
    gj_amp_part* d_amp_parts;  /* [n_parts]         assembled */
    gj_onset* d_onset_parts;   /* [n_parts]         assembled */
    gj_amp_stats* d_amp;       /* written by step 2 */
    gj_onset* d_onset;         /* written by step 2 */
    float* d_psd;              /* [max(rows,1)][nperseg] assembled */
    double* d_out;             /* [40 + n_chunks + nperseg + 5 pair_cap] written by step 3 */
} gj_combine_capture;
typedef struct gj_combine_plan gj_combine_plan;
int gj_combine_plan_create(gj_ctx* ctx, const gj_combine_copy* copies, int n_copies, const gj_combine_capture* captures,
                           int n_captures, size_t rows_bytes, const void* d_arena, size_t arena_bytes, int nperseg,
                           float pct, float rise_db, const int32_t* d_pairs /* every solved pair, static */,
                           const int32_t* d_lags, const float* d_peaks, const float* d_margins /* assembled */,
                           gj_combine_plan** out);
/* The validation of gj_combine_plan_create alone (host arithmetic only; no context, no GPU): GJ_OK, or the status the
 * create call would return for these lists.  Bounds are taken by division, so a count or stride chosen to wrap a 64-bit
 * product is refused like any other out-of-range value. */
int gj_combine_plan_check(const gj_combine_copy* copies, int n_copies, const gj_combine_capture* captures, int n_captures,
                          size_t rows_bytes, const void* d_arena, size_t arena_bytes, int nperseg, int have_pairs);
int gj_split_combine_dev(gj_ctx* ctx, const gj_combine_plan* plan, const double* d_rows);
int gj_combine_plan_destroy(gj_ctx* ctx, gj_combine_plan* plan); /* idempotent on NULL; synchronises the context's stream */

/* ------------------------------------------------- GNSS acquisition search ----------- */
/* the reference receiver's parallel code-phase search, batched over every
 * PRN and Doppler bin.  Replaces, per call, what each of its channel threads does on its own:
 *   sdraqcuisition    GpsJammerApp/backend/sdracq.c:3-50    loop over `intg` steps, early stop
 *   pcorrelator       GpsJammerApp/backend/sdrcmn.c:742-773 per bin: mixcarr -> cpxcpx -> cpxconv
 *   mixcarr           sdrcmn.c:618-705 (the SSE2 form the reference's makefile builds)
 *   cpxconv           sdrcmn.c:124-147 FFT . conj(code FFT) . IFFT -> |.|^2 / m^2, accumulated
 *   checkacquisition  sdracq.c:52-84   peak, +-2 chip exclusion zone, peak ratio > threshold
 * Inputs (device memory): the capture as int8 = u8 - 128 (sdrrcv.c:104-106); step s of the search
 * reads the 2*nsamp samples from first_sample + s*nsamp; d_codes[p][nsamp] = the code of PRN p
 * resampled to the sampling rate (+-1, int16: rescode, sdrinit.c:439); d_phase[f][2*nsamp] = the
 * mixer's 4-bit phase index per sample for Doppler bin f, built on the host the way mixcarr
 * builds it (gpsjam/gnss.py).  nsamp: 512, 1024 or 2048 (FFT length 2*nsamp, sdrinit.c:402).
 * A PRN stops integrating at the first step whose peak ratio passes (results of that step are
 * kept); the others run all `intg` steps.  d_power (optional) receives the accumulated
 * correlation power double[n_prn][n_freq][nsamp] (the reference's `power` array). */
typedef struct gj_acq_result {
    double max_power;    /* maxP */
    double second_power; /* maxP2: largest power of the peak's Doppler row outside the exclusion zone */
    double mean_power;   /* meanP of that row outside the exclusion zone */
    double peak_ratio;   /* maxP / maxP2 (acq.peakr) */
    double cn0;          /* 10 log10(maxP / meanP / ctime) */
    int32_t code_index;  /* acq.acqcodei */
    int32_t freq_index;  /* acq.freqi */
    int32_t steps;       /* integration steps used, 1..intg */
    int32_t acquired;    /* peak_ratio > threshold */
} gj_acq_result;
int gj_acq_search_dev(gj_ctx* ctx, const uint8_t* d_iq, size_t nbytes, size_t first_sample, int nsamp,
                      int intg, const int16_t* d_codes, int n_prn, const uint8_t* d_phase, int n_freq,
                      int nsampchip, double ctime, float threshold, gj_acq_result* d_out /* [n_prn] */,
                      double* d_power /* [n_prn*n_freq*nsamp] or NULL */);
size_t gj_acq_workspace(gj_ctx* ctx, int nsamp, int n_freq, int n_prn, int intg, int with_power);

/* ------------------------------------------------- collectives (RCCL over xGMI) ------- */
/* One communicator rank per GPU / process, for hosts without torch.distributed.  The path
 * shards by capture (file k -> GPU k) and has ONE exchange: TDOA slots and per-stream result
 * vectors travel to the solving rank (ncclGather, rccl.h:745).  The reference has no
 * equivalent (one process, numpy).  Rank 0 calls gj_comm_unique_id and ships the 128 bytes to
 * the other ranks by whatever means the host has (gpsjam/comm.py: one TCP socket on
 * MASTER_ADDR:MASTER_PORT); every rank then calls gj_comm_init_rank on ITS context.
 * Collectives are enqueued on the context's current stream (gj_set_stream) and do not
 * synchronise the host; buffers are device memory.  The context lock is NOT held across the RCCL
 * call (a communicator's first collective connects its peers inside the call and may wait for a
 * late rank): other threads keep using the context meanwhile; the order of collectives on one
 * communicator is the caller's.  gj_comm_destroy -- and gj_destroy of the communicator's context --
 * detach the communicator first (a call that starts afterwards answers GJ_ERR_INVALID) and wait for the
 * calls already in progress on it to return before RCCL's handle is destroyed; the two must not be
 * called concurrently with each other for the same context.  librccl is bound at run time
 * (GJ_ERR_UNSUPPORTED when it cannot be loaded). */
typedef struct gj_comm gj_comm;
#define GJ_COMM_ID_BYTES 128
int gj_comm_unique_id(void* id /* [GJ_COMM_ID_BYTES] */);
int gj_comm_init_rank(gj_ctx* ctx, const void* id, int rank, int n_ranks, gj_comm** out);
/* rank and size as the LIVE communicator reports them (ncclCommUserRank / ncclCommCount), not as passed at init;
 * -1 of 0 for a communicator whose context has been destroyed.  gj_comm_device: the HIP device it is bound to. */
int gj_comm_rank(gj_comm* comm, int* rank, int* n_ranks);
int gj_comm_device(gj_comm* comm, int* hip_device);
/* every rank sends `bytes` bytes; root receives n_ranks*bytes in rank order (d_recv may be NULL
 * elsewhere) */
int gj_comm_gather_dev(gj_comm* comm, const void* d_send, size_t bytes, void* d_recv, int root);
/* every rank sends `bytes` bytes and receives all n_ranks*bytes in rank order (ncclAllGather) */
int gj_comm_allgather_dev(gj_comm* comm, const void* d_send, size_t bytes, void* d_recv);
int gj_comm_bcast_dev(gj_comm* comm, void* d_buf, size_t bytes, int root);
int gj_comm_destroy(gj_comm* comm); /* idempotent on NULL; synchronises the context's stream */

/* ------------------------------------------------- synthetic captures --------------- */
/* Bit-identical to gpsjam/synth.py (integer-only counter-based generator that mirrors
 * the value distribution of simulate/frontend/weaken_gps.py + add_jammer_and_mix.py). */
typedef struct gj_synth_params {
    uint64_t key_noise;  /* per-antenna hash key */
    uint64_t key_common; /* common-source hash key */
    int64_t delay;       /* samples by which this antenna sees the source late */
    int64_t jam_start;   /* source-time sample range [start, end) of the burst */
    int64_t jam_end;
    int32_t noise_k; /* fixed-point gains, see synth.gain_k */
    int32_t jam_k;
    int32_t dc_i_q8; /* DC offsets in 1/256 LSB */
    int32_t dc_q_q8;
} gj_synth_params;
int gj_synth_u8_dev(gj_ctx* ctx, const gj_synth_params* params, int64_t first_sample,
                    size_t n_samples, uint8_t* d_out /* [2*n_samples] */);

#ifdef __cplusplus
}
#endif
#endif /* GPSJAM_H */
