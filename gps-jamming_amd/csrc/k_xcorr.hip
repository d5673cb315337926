// K5: pairwise TDOA cross-correlation lag (gfx950).
// Replaces scipy.signal.correlate(sig1, sig0, 'full') + argmax|.| - (N-1)
// (skrypty/triangulateTDOA.py:80-89; scipy/signal/_signaltools.py fftconvolve).
//
// Linear correlation through a zero-padded circular one of length L = 2^p >= 2N-1 (any
// length >= 2N-1 gives the same correlation values; scipy picks next_fast_len, we pick the
// next power of two).  L = L1 x 4096 four-step FFT out of HBM/L2 built from the block FFT
// of fft_core.h:
//   forward  : columns (length L1, stride 4096; unpack + zero-pad fused) -> twiddle W_L^(k1 n2)
//              -> rows (length 4096, contiguous)            spectrum stored as Z[k1][k2], k = k1 + L1 k2
//   inverse  : conj(C) = conj(Z_j) Z_i fused into the row pass -> twiddle -> column pass with
//              |.|^2 + arg-max fused (the correlation itself is never written).
// The inverse uses ifft(C) = conj(fft(conj C))/L, and |.| is conjugation-invariant.
// Tie rule of numpy.argmax (first maximum in 'full' order m = lag + N-1) is kept.
#include "gj_common.h"

// ---- build-time experiment knobs (defaults = the shipped configuration; tools/ab_build.sh flips them) ----
#ifndef GJ_XC_NT
#define GJ_XC_NT 0        // 1: the spectra between the launches are stored / loaded non-temporally (each is used once): -2 % at 2^19-sample slices, +2.5 % at the reference's 50 000 (profiles/r06_k5_variants.txt): off
#endif
#ifndef GJ_XC_SWIZZLE
#define GJ_XC_SWIZZLE 0   // 1: column tiles are dealt so that one XCD walks a contiguous range of columns: no effect (same file): off
#endif

namespace gj {

constexpr int kRow = 4096;   // L2: contiguous row length

// the spectra that travel between K5's launches (Y after the forward columns, D after the rows)
__device__ __forceinline__ void xc_store(cf* p, c2 v) {
#if GJ_XC_NT
    typedef float f2 __attribute__((ext_vector_type(2)));
    __builtin_nontemporal_store(f2{v.x, v.y}, reinterpret_cast<f2*>(p));
#else
    *p = to_cf(v);
#endif
}
__device__ __forceinline__ c2 xc_load(const cf* p) {
#if GJ_XC_NT
    typedef float f2 __attribute__((ext_vector_type(2)));
    const f2 t = __builtin_nontemporal_load(reinterpret_cast<const f2*>(p));
    return make_c2(t.x, t.y);
#else
    return to_c2(*p);
#endif
}
// column tile of workgroup x out of `tiles` (a multiple of 8 for every L >= 2^16): workgroups are dealt round-robin over
// the eight XCDs, so with the swizzle XCD k walks tiles [k tiles/8, (k+1) tiles/8) in order
__device__ __forceinline__ unsigned xc_tile(unsigned x, unsigned tiles) {
#if GJ_XC_SWIZZLE
    return (tiles & 7u) ? x : (x & 7u) * (tiles >> 3) + (x >> 3);
#else
    (void)tiles;
    return x;
#endif
}

struct XcParams {
    int off2;   // unpack convention: 2 * offset (255)
    const uint8_t* iq[GJ_MAX_ANTENNAS];
    const long long* start_ptr[GJ_MAX_ANTENNAS];   // where antenna a's start sample (or slot flag word) lives
    unsigned long long nsamples[GJ_MAX_ANTENNAS];
    int pair_i[GJ_MAX_ANTENNAS * GJ_MAX_ANTENNAS / 2 + 8];
    int pair_j[GJ_MAX_ANTENNAS * GJ_MAX_ANTENNAS / 2 + 8];
    unsigned long long n;   // slice length in samples
    unsigned long long L;   // FFT length
    int L1;
    int n_ant, n_pairs;
};
constexpr int kMaxPairs = GJ_MAX_ANTENNAS * GJ_MAX_ANTENNAS / 2 + 8;

struct XcCand {
    float val;    // largest |c|^2
    int m;        // its 'full'-mode index (first one on a tie: numpy.argmax)
    float val2;   // largest |c|^2 at any OTHER index (decision margin, gj_xcorr_lags_dev d_margins)
    int pad;
};

// winner of two candidates; the loser's peak becomes a runner-up of the winner
__device__ __forceinline__ XcCand xc_merge(XcCand a, XcCand b) {
    const bool take_b = b.val > a.val || (b.val == a.val && b.m < a.m);
    XcCand w = take_b ? b : a;
    const XcCand l = take_b ? a : b;
    w.val2 = fmaxf(w.val2, l.val);
    return w;
}

__device__ __forceinline__ c2 twiddle_big(unsigned long long m, unsigned long long L) {
    // exp(-2 pi i m / L), m < L <= 2^24: the ratio is exact in float
    float s, c;
    sincospif(-2.0f * ((float)m / (float)L), &s, &c);
    return make_c2(c, s);
}

// One transform.  The twiddles of pass P + 1 are loaded right behind the scatter of pass P -- the points are in LDS, their
// registers are free, and the fifteen loads are in flight across the exchange's barriers.  Round 4: until then every pass
// loaded its twiddles right in front of its butterflies and the compiler issued them one at a time, each with a full
// wait (90 serialised L2 round trips per workgroup in xc_rows_pair_kernel: visible as L[0]L[0]L[0]... in the ISA).
// With many workgroups in flight other waves covered that; the reference's own slice size (50 000 samples, FFT length
// 131 072: 96 row workgroups) did not.  Same arithmetic, same bits.
template <int N, int PASS>
__device__ __forceinline__ void xc_passes_tw(c2 (&v)[16], cf* lds, int base, int jl, const cf* twtab, const c2 (&tw)[15]) {
    constexpr int NP = fft_npass(N);
    fft_pass<N, PASS>(v, tw, inner_twiddles());
    if constexpr (PASS + 1 < NP) {
        lds_scatter<N, PASS>(v, lds, base, jl);
        c2 nxt[15];
        load_twiddles<N, PASS + 1>(nxt, twtab, jl);
        __builtin_amdgcn_sched_barrier(0);   // all fifteen issued here, not drip-fed behind the gather
        __syncthreads();
        lds_gather<N>(v, lds, base, jl);
        __syncthreads();
        xc_passes_tw<N, PASS + 1>(v, lds, base, jl, twtab, nxt);
    }
}

template <int N, int PASS>
__device__ __forceinline__ void xc_passes(c2 (&v)[16], cf* lds, int base, int jl, const cf* twtab) {
    static_assert(PASS == 0, "transforms start at pass 0 (which has no twiddles)");
    c2 none[15];
#pragma unroll
    for (int k = 0; k < 15; ++k) none[k] = make_c2(1.f, 0.f);
    xc_passes_tw<N, 0>(v, lds, base, jl, twtab, none);
}

// start word of antenna a: an element of the caller's start array, or the flag word of a TDOA slot.  Read by every
// workgroup that needs it (a wave-uniform 8-byte load) -- until round 5 a one-wave kernel of its own in front of the
// transforms, i.e. one more dependent launch on a chain of six.
__device__ __forceinline__ bool xc_start(const XcParams& P, int a, long long& eff) {
    const long long s = *P.start_ptr[a];
    const bool ok = s >= 0 && (unsigned long long)s + P.n <= P.nsamples[a];
    eff = ok ? s : 0;
    return ok;
}

// ---- column pass: transforms of length L1 over rows, a tile of 4096/L1 adjacent columns ----
// MODE 0: forward, input = uint8 slice (zero-padded), output Y[a][k1][n2] * W_L^(k1 n2)
// MODE 1: inverse tail, input = D[p][k1][n2], output = per-workgroup arg-max candidate
template <int L1, int MODE>
__global__ __launch_bounds__(kBlockThreads) void xc_cols_kernel(XcParams P, const cf* __restrict__ twtab, cf* __restrict__ buf,
                                                                XcCand* __restrict__ cand, unsigned* __restrict__ arrive,
                                                                int* __restrict__ lags, float* __restrict__ peaks,
                                                                float* __restrict__ margins) {
    constexpr int TF = L1 / 16, B = kBlockPoints / L1;
    constexpr int RS = lds_span(L1) + 1;   // LDS region stride: b-fastest lanes land on different banks
    __shared__ cf lds[B * RS + 16];
    __shared__ XcCand red[kBlockThreads / 64];
    __shared__ int last_s;
    const int tid = threadIdx.x;
    const int b = tid % B, jl = tid / B;
    const int t = blockIdx.y;   // antenna (MODE 0) or pair (MODE 1)
    const unsigned tile = xc_tile(blockIdx.x, gridDim.x);
    const int n2 = (int)tile * B + b;
    c2 v[16];
    if constexpr (MODE == 0) {
        long long eff;
        const bool ok = xc_start(P, t, eff);
        const uint16_t* src = reinterpret_cast<const uint16_t*>(P.iq[t]) + eff;
        // L <= 2^24: every index fits 32 bits.  Branch-free: out-of-slice points read the slice's
        // first sample and are zeroed by a select (per-element branches made hipcc shuffle the
        // whole register array through AGPRs: 4800 instructions, 236 VGPRs)
        const unsigned nlim = ok ? (unsigned)P.n : 0u;
        unsigned raw[16];
#pragma unroll
        for (int s = 0; s < 16; ++s) raw[s] = 0u;
        if (nlim) {   // workgroup-uniform: an invalid antenna's pointer is never dereferenced
#pragma unroll
            for (int s = 0; s < 16; ++s) {
                const unsigned n = (unsigned)(jl + TF * s) * (unsigned)kRow + (unsigned)n2;
                raw[s] = src[n < nlim ? n : 0u];
            }
        }
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            const unsigned n = (unsigned)(jl + TF * s) * (unsigned)kRow + (unsigned)n2;
            const unsigned u = raw[s];
            const c2 x = make_c2((float)(2 * (int)(u & 255u) - P.off2), (float)(2 * (int)(u >> 8) - P.off2));
            v[s] = (n < nlim) ? x : make_c2(0.f, 0.f);
        }
    } else {
        const cf* src = buf + (size_t)t * P.L;
#pragma unroll
        for (int s = 0; s < 16; ++s) v[s] = xc_load(src + (unsigned)(jl + TF * s) * (unsigned)kRow + (unsigned)n2);
    }
    xc_passes<L1, 0>(v, lds, b * RS, jl, twtab);
    if constexpr (MODE == 0) {
        cf* dst = buf + (size_t)t * P.L;
        // W_L^(k1 n2), k1 = jl + TF s: two sincospi and a geometric recurrence instead of sixteen
        c2 w = twiddle_big((unsigned long long)jl * n2, P.L);
        const c2 step = twiddle_big(((unsigned long long)TF * n2) & (P.L - 1), P.L);
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            const unsigned k1 = (unsigned)(jl + TF * s);
            xc_store(dst + k1 * (unsigned)kRow + (unsigned)n2, cmul(v[s], w));
            w = cmul(w, step);
        }
    } else {
        XcCand c{-1.f, 0x7fffffff, -1.f, 0};
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            const unsigned long long n = (unsigned long long)(jl + TF * s) * kRow + n2;
            long long m;
            if (n < P.n) m = (long long)n + (long long)P.n - 1;                 // lag = n >= 0
            else if (n > P.L - P.n) m = (long long)n - (long long)P.L + (long long)P.n - 1;   // lag = n - L < 0
            else continue;
            const float val = v[s].x * v[s].x + v[s].y * v[s].y;
            c = xc_merge(c, XcCand{val, (int)m, -1.f, 0});
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            XcCand o;
            o.val = __shfl_xor(c.val, off, 64);
            o.m = __shfl_xor(c.m, off, 64);
            o.val2 = __shfl_xor(c.val2, off, 64);
            o.pad = 0;
            c = xc_merge(c, o);
        }
        if ((tid & 63) == 0) red[tid >> 6] = c;
        __syncthreads();
        // The pair's candidates meet in its LAST workgroup to finish (arrival counter of the pair, agent-scope release /
        // acquire: gj_common.h), which picks the winner in candidate order -- the reduction a one-workgroup-per-pair
        // kernel used to do in a launch of its own.  xc_merge is a total order with the runner-up carried along, so the
        // result does not depend on who arrives last.
        if (tid == 0) {
            XcCand r = red[0];
            for (int k = 1; k < kBlockThreads / 64; ++k) r = xc_merge(r, red[k]);
            cand[(size_t)t * gridDim.x + blockIdx.x] = r;
            const unsigned ticket = arrive_release(arrive + t);
            const int last = ticket == gridDim.x - 1;
            if (last) last_arriver_acquire(arrive + t);
            last_s = last;
        }
        __syncthreads();
        if (!last_s) return;
        XcCand r{-1.f, 0x7fffffff, -1.f, 0};
        for (unsigned k = tid; k < gridDim.x; k += kBlockThreads) r = xc_merge(r, cand[(size_t)t * gridDim.x + k]);
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            XcCand o;
            o.val = __shfl_xor(r.val, off, 64);
            o.m = __shfl_xor(r.m, off, 64);
            o.val2 = __shfl_xor(r.val2, off, 64);
            o.pad = 0;
            r = xc_merge(r, o);
        }
        if ((tid & 63) == 0) red[tid >> 6] = r;
        __syncthreads();
        if (tid == 0) {
            r = red[0];
            for (int k = 1; k < kBlockThreads / 64; ++k) r = xc_merge(r, red[k]);
            long long e;
            const bool ok = xc_start(P, P.pair_i[t], e) && xc_start(P, P.pair_j[t], e);
            lags[t] = ok ? r.m - (int)(P.n - 1) : GJ_LAG_INVALID;
            // inputs were 2(u-127.5): |c| = sqrt(val) / L / 4
            peaks[t] = ok ? sqrtf(r.val) * (0.25f / (float)P.L) : 0.f;
            // relative gap between the peak and the largest |c| at any other lag
            if (margins) margins[t] = (ok && r.val > 0.f) ? 1.0f - sqrtf(fmaxf(r.val2, 0.f) / r.val) : 0.f;
        }
    }
}

// ---- row pass: contiguous 4096-point transforms --------------------------------------------
// MODE 0: forward, in place on Y[a] -> Z[a]
// MODE 1: inverse head, input conj(Z_j) * Z_i, output D[p][k1][n2] * W_L^(k1 n2)
template <int MODE>
__global__ __launch_bounds__(kBlockThreads) void xc_rows_kernel(XcParams P, const cf* __restrict__ twtab,
                                                                cf* __restrict__ spec, cf* __restrict__ dbuf) {
    constexpr int N = kRow, TF = N / 16;
    __shared__ cf lds[lds_span(kBlockPoints)];
    const int jl = threadIdx.x;
    const int r = blockIdx.x;   // k1
    const int t = blockIdx.y;
    c2 v[16];
    if constexpr (MODE == 0) {
        const cf* src = spec + (size_t)t * P.L + (size_t)r * N;
#pragma unroll
        for (int s = 0; s < 16; ++s) v[s] = to_c2(src[jl + TF * s]);
    } else {
        const cf* zi = spec + (size_t)P.pair_i[t] * P.L + (size_t)r * N;
        const cf* zj = spec + (size_t)P.pair_j[t] * P.L + (size_t)r * N;
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            const cf a = zj[jl + TF * s], bb = zi[jl + TF * s];
            v[s] = make_c2(a.x * bb.x + a.y * bb.y, a.x * bb.y - a.y * bb.x);   // conj(a) * b
        }
    }
    xc_passes<N, 0>(v, lds, 0, jl, twtab);
    if constexpr (MODE == 0) {
        cf* dst = spec + (size_t)t * P.L + (size_t)r * N;
#pragma unroll
        for (int s = 0; s < 16; ++s) dst[jl + TF * s] = to_cf(v[s]);
    } else {
        cf* dst = dbuf + (size_t)t * P.L + (size_t)r * N;
        c2 w = twiddle_big((unsigned long long)r * jl, P.L);
        const c2 step = twiddle_big(((unsigned long long)r * TF) & (P.L - 1), P.L);
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            const int n2 = jl + TF * s;
            dst[n2] = to_cf(cmul(v[s], w));
            w = cmul(w, step);
        }
    }
}

// Few pairs (the per-stream pipeline: a rank's share of the antenna pairs): per pair the two forward row
// transforms, the product conj(Z_j) Z_i and the inverse row transform of row k1 in ONE kernel (grid.y = pair) --
// the row spectra never go to memory and there is one launch less on the chain that runs beside K2, where
// every kernel waits for K2 workgroups to leave.  An antenna in several pairs is transformed once per pair:
// used while 3 P <= 2 (A + P), i.e. P <= 2 A.
__global__ __launch_bounds__(kBlockThreads) void xc_rows_pair_kernel(XcParams P, const cf* __restrict__ twtab,
                                                                     const cf* __restrict__ spec, cf* __restrict__ dbuf) {
    constexpr int N = kRow, TF = N / 16;
    __shared__ cf lds[lds_span(kBlockPoints)];
    const int jl = threadIdx.x;
    const int r = blockIdx.x;   // k1
    const int p = blockIdx.y;   // pair
    const cf* si = spec + (size_t)P.pair_i[p] * P.L + (size_t)r * N;
    const cf* sj = spec + (size_t)P.pair_j[p] * P.L + (size_t)r * N;
    c2 vi[16], vj[16];
#pragma unroll
    for (int s = 0; s < 16; ++s) vi[s] = xc_load(si + jl + TF * s);
#pragma unroll
    for (int s = 0; s < 16; ++s) vj[s] = xc_load(sj + jl + TF * s);
    xc_passes<N, 0>(vi, lds, 0, jl, twtab);
    xc_passes<N, 0>(vj, lds, 0, jl, twtab);
#pragma unroll
    for (int s = 0; s < 16; ++s) {   // conj(Z_j) * Z_i, same expression as xc_rows_kernel<1>
        const c2 a = vj[s], bb = vi[s];
        vi[s] = make_c2(a.x * bb.x + a.y * bb.y, a.x * bb.y - a.y * bb.x);
    }
    xc_passes<N, 0>(vi, lds, 0, jl, twtab);
    cf* dst = dbuf + (size_t)p * P.L + (size_t)r * N;
    c2 w = twiddle_big((unsigned long long)r * jl, P.L);
    const c2 step = twiddle_big(((unsigned long long)r * TF) & (P.L - 1), P.L);
#pragma unroll
    for (int s = 0; s < 16; ++s) {
        xc_store(dst + jl + TF * s, cmul(vi[s], w));
        w = cmul(w, step);
    }
}

static unsigned long long xc_fft_len(size_t n) {
    unsigned long long L = 65536;
    while (L < 2ull * n - 1) L <<= 1;
    return L;
}

size_t xcorr_workspace(gj_ctx*, int n_ant, size_t n_samples, int n_pairs) {
    if (n_samples == 0) return 0;
    const unsigned long long L = xc_fft_len(n_samples);
    const size_t ncand = (size_t)(L / kBlockPoints);
    return (size_t)(n_ant + n_pairs) * L * sizeof(cf) + (size_t)n_pairs * ncand * sizeof(XcCand) + 4096;
}

template <int L1>
static void xc_launch_cols(gj_ctx* ctx, int mode, const XcParams& P, int count, cf* buf, XcCand* cand, int* lags, float* peaks,
                           float* margins) {
    const dim3 grid((unsigned)(P.L / kBlockPoints), (unsigned)count);
    unsigned* arrive = ctx->d_sync + kSyncXcorr;
    if (mode == 0)
        hipLaunchKernelGGL((xc_cols_kernel<L1, 0>), grid, dim3(kBlockThreads), 0, ctx->stream, P, ctx->d_twiddle, buf, cand, arrive,
                           lags, peaks, margins);
    else
        hipLaunchKernelGGL((xc_cols_kernel<L1, 1>), grid, dim3(kBlockThreads), 0, ctx->stream, P, ctx->d_twiddle, buf, cand, arrive,
                           lags, peaks, margins);
}

static void xc_cols(gj_ctx* ctx, int mode, const XcParams& P, int count, cf* buf, XcCand* cand, int* lags = nullptr,
                    float* peaks = nullptr, float* margins = nullptr) {
    switch (P.L1) {
        case 16: xc_launch_cols<16>(ctx, mode, P, count, buf, cand, lags, peaks, margins); break;
        case 32: xc_launch_cols<32>(ctx, mode, P, count, buf, cand, lags, peaks, margins); break;
        case 64: xc_launch_cols<64>(ctx, mode, P, count, buf, cand, lags, peaks, margins); break;
        case 128: xc_launch_cols<128>(ctx, mode, P, count, buf, cand, lags, peaks, margins); break;
        case 256: xc_launch_cols<256>(ctx, mode, P, count, buf, cand, lags, peaks, margins); break;
        case 512: xc_launch_cols<512>(ctx, mode, P, count, buf, cand, lags, peaks, margins); break;
        case 1024: xc_launch_cols<1024>(ctx, mode, P, count, buf, cand, lags, peaks, margins); break;
        case 2048: xc_launch_cols<2048>(ctx, mode, P, count, buf, cand, lags, peaks, margins); break;
        default: xc_launch_cols<4096>(ctx, mode, P, count, buf, cand, lags, peaks, margins); break;
    }
}

int launch_xcorr(gj_ctx* ctx, const uint8_t* const* d_iq, const size_t* nbytes, int n_ant,
                 const int64_t* const* start_ptrs, size_t n_samples, const int32_t* pairs, int n_pairs, int32_t* d_lags,
                 float* d_peaks, float* d_margins) {
    if (n_ant < 1 || n_ant > GJ_MAX_ANTENNAS) return fail(ctx, GJ_ERR_INVALID, "n_ant must be 1..%d", GJ_MAX_ANTENNAS);
    if (n_pairs < 1 || n_pairs > kMaxPairs) return fail(ctx, GJ_ERR_INVALID, "n_pairs must be 1..%d", kMaxPairs);
    static_assert(kSyncXcorr + kMaxPairs <= kSyncWords, "one arrival counter per pair");
    if (n_samples < 1) return fail(ctx, GJ_ERR_INVALID, "n_samples must be >= 1");
    if (n_samples > (1ull << 23)) return fail(ctx, GJ_ERR_UNSUPPORTED, "slice longer than 2^23 samples");
    XcParams P;
    memset(&P, 0, sizeof(P));
    P.off2 = ctx->off2;
    for (int a = 0; a < n_ant; ++a) {
        if (reinterpret_cast<uintptr_t>(d_iq[a]) & 1) return fail(ctx, GJ_ERR_INVALID, "capture must be 2-byte aligned");
        P.iq[a] = d_iq[a];
        P.nsamples[a] = nbytes[a] / 2;
        P.start_ptr[a] = reinterpret_cast<const long long*>(start_ptrs[a]);
    }
    for (int p = 0; p < n_pairs; ++p) {
        const int i = pairs[2 * p], j = pairs[2 * p + 1];
        if (i < 0 || j < 0 || i >= n_ant || j >= n_ant) return fail(ctx, GJ_ERR_INVALID, "pair %d out of range", p);
        P.pair_i[p] = i;
        P.pair_j[p] = j;
    }
    P.n = n_samples;
    P.L = xc_fft_len(n_samples);
    P.L1 = (int)(P.L / kRow);
    P.n_ant = n_ant;
    P.n_pairs = n_pairs;
    const size_t need = xcorr_workspace(ctx, n_ant, n_samples, n_pairs);
    int rc = ensure_workspace(ctx, need);
    if (rc) return rc;
    cf* spec = reinterpret_cast<cf*>(ctx->ws);
    cf* dbuf = spec + (size_t)n_ant * P.L;
    XcCand* cand = reinterpret_cast<XcCand*>(dbuf + (size_t)n_pairs * P.L);

    // three launches: forward columns (unpack + zero-pad fused), rows (both forward row transforms, the product and the
    // inverse row transform per pair), inverse columns with |.|^2, arg-max and the pair's final pick fused
    xc_cols(ctx, 0, P, n_ant, spec, nullptr);
    GJ_LAUNCH_CHECK(ctx);
    if (n_pairs <= 2 * n_ant) {
        hipLaunchKernelGGL(xc_rows_pair_kernel, dim3((unsigned)P.L1, (unsigned)n_pairs), dim3(kBlockThreads), 0, ctx->stream,
                           P, ctx->d_twiddle, spec, dbuf);
        GJ_LAUNCH_CHECK(ctx);
    } else {
        hipLaunchKernelGGL((xc_rows_kernel<0>), dim3((unsigned)P.L1, (unsigned)n_ant), dim3(kBlockThreads), 0, ctx->stream,
                           P, ctx->d_twiddle, spec, dbuf);
        GJ_LAUNCH_CHECK(ctx);
        hipLaunchKernelGGL((xc_rows_kernel<1>), dim3((unsigned)P.L1, (unsigned)n_pairs), dim3(kBlockThreads), 0,
                           ctx->stream, P, ctx->d_twiddle, spec, dbuf);
        GJ_LAUNCH_CHECK(ctx);
    }
    xc_cols(ctx, 1, P, n_pairs, dbuf, cand, d_lags, d_peaks, d_margins);
    GJ_LAUNCH_CHECK(ctx);
    return GJ_OK;
}

// ---- TDOA slot (layout and rules: tdoa_slot_body, gj_common.h) ----
__global__ __launch_bounds__(256) void tdoa_slot_kernel(const uint8_t* __restrict__ iq, size_t nsamples,
                                                        const long long* __restrict__ start, size_t n,
                                                        uint8_t* __restrict__ slot, long long sample0, size_t total) {
    tdoa_slot_body(iq, nsamples, *start, n, slot, sample0, total, blockIdx.x * (size_t)blockDim.x + threadIdx.x,
                   (size_t)gridDim.x * blockDim.x, blockIdx.x == 0 && threadIdx.x == 0);
}

int launch_tdoa_slot(gj_ctx* ctx, const uint8_t* d_iq, size_t nbytes, const int64_t* d_start, size_t n_samples,
                     uint8_t* d_slot, long long sample0, size_t total_samples) {
    if (total_samples == 0 && sample0 == 0) total_samples = nbytes / 2;   // a whole capture
    if (n_samples < 1) return fail(ctx, GJ_ERR_INVALID, "n_samples must be >= 1");
    if ((reinterpret_cast<uintptr_t>(d_iq) & 1) != 0) return fail(ctx, GJ_ERR_INVALID, "capture must be 2-byte aligned");
    if ((reinterpret_cast<uintptr_t>(d_slot) & 15) != 0) return fail(ctx, GJ_ERR_INVALID, "slot must be 16-byte aligned");
    size_t blocks = ((n_samples + 7) / 8 + 16 + 255) / 256;
    if (blocks > 1024) blocks = 1024;
    hipLaunchKernelGGL(tdoa_slot_kernel, dim3((unsigned)blocks), dim3(256), 0, ctx->stream, d_iq, nbytes / 2,
                       (const long long*)d_start, n_samples, d_slot, sample0, total_samples);
    GJ_LAUNCH_CHECK(ctx);
    return GJ_OK;
}

// A capture split over GPUs contributes one slot per part; the capture's slot is the one cut at the SMALLEST
// onset >= 0 among its parts (the whole capture's first crossing), whatever its flag says -- exactly the slot the
// unsplit capture would have produced.  No part found an onset: an invalid slot with start -1.
// grid (copy blocks, groups); group g = the slots members[offsets[g]] .. members[offsets[g + 1] - 1].
__global__ __launch_bounds__(256) void slots_pick_kernel(const uint8_t* __restrict__ slots, size_t stride,
                                                         const int* __restrict__ offsets, const int* __restrict__ members,
                                                         uint8_t* __restrict__ out) {
    const int g = blockIdx.y;
    int best = -1;
    long long best_s = 0x7fffffffffffffffll;
    for (int k = offsets[g]; k < offsets[g + 1]; ++k) {
        const long long* h = reinterpret_cast<const long long*>(slots + (size_t)members[k] * stride);
        const long long st = h[1];
        if (st >= 0 && st < best_s) { best_s = st; best = members[k]; }
    }
    uint4* dst = reinterpret_cast<uint4*>(out + (size_t)g * stride);
    const size_t ngroups = stride / 16;
    if (best < 0) {
        for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < ngroups; i += (size_t)gridDim.x * blockDim.x)
            dst[i] = (i == 0) ? uint4{0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu} : uint4{0u, 0u, 0u, 0u};
        return;
    }
    const uint4* src = reinterpret_cast<const uint4*>(slots + (size_t)best * stride);
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < ngroups; i += (size_t)gridDim.x * blockDim.x) dst[i] = src[i];
}

int launch_slots_pick(gj_ctx* ctx, const uint8_t* d_slots, size_t stride, const int* d_offsets, const int* d_members, int n_groups,
                      uint8_t* d_out) {
    if (n_groups < 1) return fail(ctx, GJ_ERR_INVALID, "n_groups must be >= 1");
    if ((stride & 15) || ((reinterpret_cast<uintptr_t>(d_slots) | reinterpret_cast<uintptr_t>(d_out)) & 15))
        return fail(ctx, GJ_ERR_INVALID, "slots must be 16-byte aligned");
    size_t blocks = (stride / 16 + 255) / 256;
    if (blocks > 256) blocks = 256;
    hipLaunchKernelGGL(slots_pick_kernel, dim3((unsigned)blocks, (unsigned)n_groups), dim3(256), 0, ctx->stream, d_slots,
                       stride, d_offsets, d_members, d_out);
    GJ_LAUNCH_CHECK(ctx);
    return GJ_OK;
}

}   // namespace gj
