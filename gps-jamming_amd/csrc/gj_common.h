// Internal definitions shared by the HIP translation units of libgpsjam_hip.so.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <mutex>

#include "../../include/gpsjam.h"
#include "fft_core.h"

struct gj_ctx {
    int device = 0;
    int num_cus = 256;
    hipStream_t own_stream = nullptr;
    hipStream_t stream = nullptr;
    hipEvent_t ev_start = nullptr, ev_stop = nullptr;
    std::recursive_mutex mu;
    gj::cf* d_twiddle = nullptr;   // W_4096^m
    unsigned char* ws = nullptr;   // grow-only workspace for partial results
    size_t ws_bytes = 0;
    unsigned char* stage = nullptr;   // grow-only device staging for the host-buffer entry points
    size_t stage_bytes = 0;
    char last_error[512] = {0};
};

namespace gj {

inline int fail(gj_ctx* ctx, int code, const char* fmt, ...) {
    if (ctx) {
        va_list ap;
        va_start(ap, fmt);
        vsnprintf(ctx->last_error, sizeof(ctx->last_error), fmt, ap);
        va_end(ap);
    }
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

struct Guard {
    gj_ctx* c;
    explicit Guard(gj_ctx* ctx) : c(ctx) {
        c->mu.lock();
        (void)hipSetDevice(c->device);
    }
    ~Guard() { c->mu.unlock(); }
};

// grow-only device arenas
int ensure_workspace(gj_ctx* ctx, size_t bytes);
int ensure_stage(gj_ctx* ctx, size_t bytes);

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

}   // namespace gj

// entry points implemented per translation unit (called from api.hip)
namespace gj {
int launch_chunk_power(gj_ctx*, const uint8_t*, size_t, size_t, float, int, float*);
int launch_power_threshold(gj_ctx*, const float*, size_t, float, float, float*, uint8_t*);
int launch_amp_stats(gj_ctx*, const uint8_t*, size_t, float, gj_amp_stats*);
int launch_onset(gj_ctx*, const uint8_t*, size_t, int, int, float, gj_onset*);
int launch_histogram(gj_ctx*, const uint8_t*, size_t, size_t, int, int, unsigned long long*);
int launch_welch(gj_ctx*, const uint8_t*, size_t, size_t, int, double, int, float*, float*);
size_t welch_workspace(gj_ctx*, size_t, size_t, int);
int launch_xcorr(gj_ctx*, const uint8_t* const*, const size_t*, int, const int64_t*, size_t,
                 const int32_t*, int, int32_t*, float*);
size_t xcorr_workspace(gj_ctx*, int, size_t, int);
int launch_synth(gj_ctx*, const gj_synth_params&, int64_t, size_t, uint8_t*);
}   // namespace gj
