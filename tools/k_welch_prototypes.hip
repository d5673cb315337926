// FROZEN READING COPY (round 2): K2 with its measured-and-rejected variants (GJ_W_PRIO, GJ_W_RAWREUSE, GJ_W_SCANSUMS,
// GJ_STAMPS).  Not part of the build: the production kernel is gps-jamming_amd/csrc/k_welch.hip; DESIGN.md section 4-5
// records what each variant measured.

// K2: fused uint8 unpack + periodic-Hann Welch PSD (gfx950).
// Replaces skrypty/widmo_plot.py:38-52 including scipy.signal.welch(..., nperseg=N,
// return_onesided=False) (widmo_plot.py:48; scipy/signal/_spectral_py.py _spectral_helper).
//
// One 256-thread workgroup owns a run of consecutive segments of one 1-s chunk.  Per step it
// transforms 4096 points = 4096/N overlapping segments: every thread pulls its 16 samples
// straight from the uint8 stream (2-byte loads, 128 B per wave instruction; the 50 % overlap
// re-read is served by L2), applies unpack and window (one FMA + one multiply per component), runs the
// register-resident Stockham passes of fft_core.h with LDS exchanges, and accumulates
// |X[k]|^2 in 16 VGPRs for the bins it ends up holding.  Per-segment mean removal
// (detrend='constant') is applied in the frequency domain: the periodic Hann window has
// only three non-zero DFT bins (N/2 at 0, -N/4 at +-1), so
//     FFT(w (v - m)) = FFT(w v) - m W   touches bins 0, 1, N-1 only,
// with m = (exact integer sum of the segment)/N from a wave-shuffle reduction.
// Work in integer LSB units v = 2u-255 (= 255 x); 1/255^2, 1/(fs sum w^2) and 1/nseg are
// folded into the finalize kernel, which also sums the per-workgroup partial spectra in a
// fixed order (deterministic), applies fftshift and writes the optional dB row.
#include "gj_common.h"

// ---- build-time tuning knobs (defaults = the shipped configuration; tools/ab_build.sh flips them)
#ifndef GJ_LB
#define GJ_LB 2          // min waves per SIMD asked of the register allocator
#endif
#ifndef GJ_W_TWOSTEP
#define GJ_W_TWOSTEP 0   // 1: six twiddles per radix-16 pass (dft16_twiddled) instead of fifteen
#endif
#ifndef GJ_W_PREFETCH
#define GJ_W_PREFETCH 1  // 1: next step's raw samples are loaded while the current one is transformed
#endif
#ifndef GJ_W_FMA
#define GJ_W_FMA 1       // 1: FMA-form radix-4 butterflies (fft_core.h dft16_fma*): ~9 % fewer packed ops
#endif
#ifndef GJ_W_PKACC
#define GJ_W_PKACC 1     // 1: |X|^2 accumulated as (re^2, im^2) pairs with one v_pk_fma_f32 per bin
#endif
#ifndef GJ_W_DBUF
#define GJ_W_DBUF 1      // 1: two LDS exchange buffers, one barrier per exchange; 0: one buffer, two barriers
#endif
#ifndef GJ_W_OCC3_MASK
// Bit k set: transform size 2^k runs in the "three workgroups per CU" shape (<= 168 VGPRs, one
// LDS buffer with two barriers per exchange, window kept as 16 floats, scalar |X|^2 accumulators:
// +32 VALU instructions per step, but a third wave per SIMD to fill the issue slots -- a wave
// issues at most one instruction every ~5 cycles, whatever its kind).  Measured on MI355X, 1 GiB,
// same box, two-workgroup shape -> three-workgroup shape:
//   N = 4096  1.328 -> 1.257 ms   2048  1.348 -> 1.320   1024  1.284 -> 1.211   512  1.242 -> 1.142
//   N = 256   1.001 -> 0.960      128   1.055 -> 1.042   64    1.278 -> 1.207   32   2.160 -> 2.119
//   N = 16    1.669 -> 1.707 (stays in the two-workgroup shape)
// (512..2048 fit 168 VGPRs only since the exchange addresses are written as base + constant,
// fft_core.h lds_scatter/lds_gather: 236-246 -> 192-206 VGPRs in the two-workgroup shape.)
#define GJ_W_OCC3_MASK 0x1FE0u
#endif
#ifndef GJ_W_HALFSUM
#define GJ_W_HALFSUM 1   // 1: (one transform per workgroup) half-segment sums carried from step to step
#endif
#ifndef GJ_W_XPOSE
#define GJ_W_XPOSE 1     // 1: N = 4096 uses the bank-conflict-free exchange schedule (fft_core.h X4096)
#endif
#ifndef GJ_W_PRIO
// s_setprio level of K2's waves (0 = the hardware default).  The SIMD's instruction arbiter serves the
// higher-priority wave first: K2 is bound by VALU issue, and the waves of the HBM-bound scan that the pipeline
// runs beside it on the second stream otherwise take issue slots away from it one for one.
#define GJ_W_PRIO 0
#endif
#ifndef GJ_W_RAWREUSE
// 1: (one transform per workgroup) the raw samples of a segment's second half are kept as the next segment's first
// half -- eight register moves instead of eight of the sixteen 2-byte loads per step
#define GJ_W_RAWREUSE 0
#endif
#ifndef GJ_W_SCANSUMS
// feasibility prototype (timing only, tools/ab_build.sh): per step also sum |z|^2 and |z| over the NEW half segment
// (what K1 / K3 / K4 need), reduce per wave and store -- "K2 as the one reader of the capture", DESIGN section 4
#define GJ_W_SCANSUMS 0
#endif
#define GJ_LOAD_RAW(x) (x)
// GJ_STAMPS (diagnostic builds only, tools/ab_build.sh): s_memtime stamps around the phases of
// a step, summed per wave and added to g_welch_stamps; read with gj_debug_welch_stamps().
#ifdef GJ_STAMPS
#define GJ_STAMP(var)                                                                        \
    unsigned long long var;                                                                  \
    __builtin_amdgcn_sched_barrier(0);                                                       \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(var)::"memory");              \
    __builtin_amdgcn_sched_barrier(0)
#define GJ_STAMP_ADD(slot, a, b) stamps[slot] += (b) - (a)
#else
#define GJ_STAMP(var)
#define GJ_STAMP_ADD(slot, a, b)
#endif

namespace gj {

constexpr int welch_log2(int n) { return n <= 1 ? 0 : 1 + welch_log2(n / 2); }
constexpr bool welch_occ3(int n) { return ((GJ_W_OCC3_MASK >> welch_log2(n)) & 1u) != 0; }
template <int N>
struct WelchCfg {
    static constexpr bool occ3 = welch_occ3(N);
    static constexpr int min_waves = occ3 ? 3 : GJ_LB;          // per SIMD, asked of the register allocator
    static constexpr bool pkacc = (GJ_W_PKACC != 0) && !occ3;   // |X|^2 as (re^2, im^2) pairs
    static constexpr bool dbuf = (GJ_W_DBUF != 0) && !occ3;     // two LDS exchange buffers
    static constexpr bool win16 = occ3;                         // window as 16 floats instead of 16 pairs
};

#ifdef GJ_STAMPS
__device__ unsigned long long g_welch_stamps[8];
#endif

struct WelchGeom {
    float neg_off;        // -offset of the unpack convention (default -127.5)
    float off2;           // 2 * offset
    unsigned long long chunk_samples;
    unsigned nchunks;     // rows kept
    unsigned splits;      // workgroups per chunk
    unsigned nseg_full;   // segments in a full chunk
    unsigned nseg_last;   // segments in the last kept chunk
};

template <int N>
struct WelchBins {   // (thread, slot) that ends up holding bin k
    static constexpr int TF = N / 16;
    static constexpr int jl(int k) { return k % TF; }
    static constexpr int slot(int k) { return k / TF; }
};

// Two LDS buffers, used alternately by consecutive exchanges: a thread may scatter into one
// while slower threads of the workgroup still gather from the other, so ONE barrier per
// exchange (between scatter and gather) is enough.
template <int N, int PASS>
__device__ __forceinline__ void welch_passes(c2 (&v)[16], cf* lds0, cf* lds1, unsigned it, int base, int jl,
                                             const c2 (&tw)[3][15], const InnerTw& ktw,
                                             unsigned long long (&stamps)[8]) {
    constexpr int NP = fft_npass(N);
    GJ_STAMP(t0);
    fft_pass<N, PASS, GJ_W_TWOSTEP != 0, GJ_W_FMA != 0>(v, tw[PASS], ktw);
    GJ_STAMP(t1);
    GJ_STAMP_ADD(0, t0, t1);   // butterflies
    if constexpr (PASS + 1 < NP) {
        // exchanges per segment: NP-1.  Even count -> parity of PASS; odd count -> parity of (it + PASS)
        const bool second = ((NP - 1) % 2 == 0) ? (PASS & 1) : ((it + PASS) & 1);
        cf* lds = (WelchCfg<N>::dbuf && second) ? lds1 : lds0;
        lds_scatter<N, PASS>(v, lds, base, jl);
        GJ_STAMP(t2);
        GJ_STAMP_ADD(1, t1, t2);   // scatter issued and landed (the stamp waits lgkmcnt(0))
        __syncthreads();
        GJ_STAMP(t3);
        GJ_STAMP_ADD(2, t2, t3);   // barrier wait
        lds_gather<N>(v, lds, base, jl);
        GJ_STAMP(t4);
        GJ_STAMP_ADD(3, t3, t4);   // gather
        if (!WelchCfg<N>::dbuf) __syncthreads();
        welch_passes<N, PASS + 1>(v, lds0, lds1, it, base, jl, tw, ktw, stamps);
    }
}

// N = 4096 with the conflict-free exchange schedule of fft_core.h (X4096): pass 0 in role
// jl0 = tid, passes 1 and 2 in role jl1; buffer 0 carries exchange 0, buffer 1 exchange 1.
__device__ __forceinline__ void welch_passes_x4096(c2 (&v)[16], cf* lds0, cf* lds1, int tid, const c2 (&tw)[3][15],
                                                   const InnerTw& ktw, unsigned long long (&stamps)[8]) {
    GJ_STAMP(t0);
    fft_pass<4096, 0, false, GJ_W_FMA != 0>(v, tw[0], ktw);
    GJ_STAMP(t1);
    GJ_STAMP_ADD(0, t0, t1);
    x4096_scatter<0>(v, lds0, tid);
    GJ_STAMP(t2);
    GJ_STAMP_ADD(1, t1, t2);
    __syncthreads();
    GJ_STAMP(t3);
    GJ_STAMP_ADD(2, t2, t3);
    x4096_gather<0>(v, lds0, tid);
    if (!WelchCfg<4096>::dbuf) __syncthreads();
    GJ_STAMP(t4);
    GJ_STAMP_ADD(3, t3, t4);
    fft_pass<4096, 1, GJ_W_TWOSTEP != 0, GJ_W_FMA != 0>(v, tw[1], ktw);
    GJ_STAMP(t5);
    GJ_STAMP_ADD(0, t4, t5);
    cf* ldsx = WelchCfg<4096>::dbuf ? lds1 : lds0;
    x4096_scatter<1>(v, ldsx, tid);
    GJ_STAMP(t6);
    GJ_STAMP_ADD(1, t5, t6);
    __syncthreads();
    GJ_STAMP(t7);
    GJ_STAMP_ADD(2, t6, t7);
    x4096_gather<1>(v, ldsx, tid);
    if (!WelchCfg<4096>::dbuf) __syncthreads();
    GJ_STAMP(t8);
    GJ_STAMP_ADD(3, t7, t8);
    fft_pass<4096, 2, GJ_W_TWOSTEP != 0, GJ_W_FMA != 0>(v, tw[2], ktw);
    GJ_STAMP(t9);
    GJ_STAMP_ADD(0, t8, t9);
}

template <int N>
__global__ __launch_bounds__(kBlockThreads, WelchCfg<N>::min_waves) void welch_kernel(const uint8_t* __restrict__ iq, WelchGeom g,
                                                              const cf* __restrict__ twtab,
                                                              const float* __restrict__ wintab,
                                                              float* __restrict__ partial) {
    constexpr int TF = N / 16, B = kBlockPoints / N, NP = fft_npass(N);
    constexpr int WPF = (TF >= 64) ? TF / 64 : 1;   // waves per transform
    constexpr bool XP = (N == 4096) && GJ_W_XPOSE;
    constexpr int SPAN = XP ? X4096::kSpan : lds_span(kBlockPoints);
    __shared__ cf lds0[SPAN];
    using Cfg = WelchCfg<N>;
    constexpr bool HS = (B == 1) && (GJ_W_HALFSUM != 0);
    __shared__ cf lds1[Cfg::dbuf ? SPAN : 1];
    // (sum I, sum Q) per wave; three slots when half-segment sums are carried over (see HS below)
    __shared__ float wsum[3][B][WPF][2];
    if constexpr (GJ_W_PRIO != 0) __builtin_amdgcn_s_setprio(GJ_W_PRIO);
    const int tid = threadIdx.x;
    const int b = tid / TF, jl0 = tid % TF;   // jl0: butterfly of pass 0 (input index jl0 + TF s)
    const int jl = XP ? X4096::jl1(tid) : jl0;   // butterfly of the later passes = bins held at the end
    const unsigned c = blockIdx.x / g.splits, part = blockIdx.x % g.splits;
    const unsigned nseg = (c + 1 == g.nchunks) ? g.nseg_last : g.nseg_full;
    const unsigned seg_lo = (unsigned)((unsigned long long)part * nseg / g.splits);
    const unsigned seg_hi = (unsigned)((unsigned long long)(part + 1) * nseg / g.splits);

    const InnerTw ktw = inner_twiddles();
    c2 tw[3][15];
#pragma unroll
    for (int p = 0; p < 3; ++p)
#pragma unroll
        for (int k = 0; k < 15; ++k) tw[p][k] = make_c2(1.f, 0.f);
    if constexpr (NP > 1) {
        if constexpr (GJ_W_TWOSTEP && fft_radix(N, 1) == 16) load_twiddles6<N, 1>(tw[1], twtab, jl);
        else load_twiddles<N, 1>(tw[1], twtab, jl);
    }
    if constexpr (NP > 2) {
        if constexpr (GJ_W_TWOSTEP && fft_radix(N, 2) == 16) load_twiddles6<N, 2>(tw[2], twtab, jl);
        else load_twiddles<N, 2>(tw[2], twtab, jl);
    }

    // window folded into the unpack: w (2u - 255) = u (2w) + (-255 w); (w[2i], w[2i+1]) share a
    // VGPR pair and op_sel picks the half, so 16 points cost 16 register pairs
    c2 w2p[8], wcp[Cfg::win16 ? 1 : 8];
#pragma unroll
    for (int s = 0; s < 8; ++s) {
        const float wa = wintab[jl0 + TF * (2 * s)], wb = wintab[jl0 + TF * (2 * s + 1)];
        w2p[s] = make_c2(2.0f * wa, 2.0f * wb);
        if constexpr (!Cfg::win16) wcp[s] = make_c2(-g.off2 * wa, -g.off2 * wb);
    }
    [[maybe_unused]] const c2 khalf = make_c2(g.neg_off, g.neg_off);
    c2 accp[Cfg::pkacc ? 16 : 1];      // (sum re^2, sum im^2): one packed FMA per bin and step
    float accs[Cfg::pkacc ? 1 : 16];   // or scalar sums (two FMAs per bin and step, 16 VGPRs fewer)
#pragma unroll
    for (int s = 0; s < 16; ++s) {
        if constexpr (Cfg::pkacc) accp[s] = make_c2(0.f, 0.f);
        else accs[s] = 0.f;
    }

    // wave-uniform chunk base + 32-bit per-lane byte offsets: the loads keep their addresses in
    // one SGPR pair + one VGPR + immediates
    const uint8_t* chunk8 = iq + (size_t)c * g.chunk_samples * 2;
    auto load_step = [&](unsigned (&dst)[16], unsigned seg_idx) {
        const unsigned byte0 = (seg_idx * (unsigned)(N / 2) + (unsigned)jl0) * 2u;
#pragma unroll
        for (int s = 0; s < 16; ++s)
            dst[s] = GJ_LOAD_RAW(*reinterpret_cast<const uint16_t*>(chunk8 + (byte0 + 2u * TF * s)));
    };
    const unsigned nsteps = (seg_hi - seg_lo + B - 1) / B;
    // raw samples of the NEXT step are fetched while the current one is transformed
    unsigned raw[16];
    load_step(raw, (seg_lo + b < seg_hi) ? seg_lo + b : seg_lo);
    unsigned long long stamps[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    [[maybe_unused]] unsigned ws_cur = 0, ws_prv = 2;   // HS: slot of this step's half-sum / of the previous step's
    GJ_STAMP(t_begin);
    for (unsigned it = 0; it < nsteps; ++it) {
        GJ_STAMP(t_it0);
        const unsigned seg = seg_lo + it * B + b;
        const bool active = seg < seg_hi;
        c2 v[16];
        if (!GJ_W_PREFETCH && it > 0) load_step(raw, active ? seg : seg_lo);
        // HS (one transform per workgroup = consecutive segments per step): the first half of a
        // segment is the second half of the previous one, so only the second half is summed each
        // step and the previous step's half-sum is read back from its slot (eight packed adds less)
        const unsigned cur = HS ? ws_cur : (it & 1), prv = HS ? ws_prv : 0;
        if (HS && it == 0) {   // first step of the workgroup: the first half has no predecessor
            c2 flo = make_c2(0.f, 0.f);
#pragma unroll
            for (int s = 0; s < 8; ++s) flo = cadd(flo, make_c2((float)(raw[s] & 255u), (float)((raw[s] >> 8) & 255u)));
            const float li = group_sum_dpp_f<64>(flo.x), lq = group_sum_dpp_f<64>(flo.y);
            if ((tid & 63) == 0) {
                wsum[prv][b][(tid >> 6) % WPF][0] = li;
                wsum[prv][b][(tid >> 6) % WPF][1] = lq;
            }
        }
        c2 fsum = make_c2(0.f, 0.f);   // (sum I, sum Q) of the raw bytes: integers < 2^24, exact in f32
        [[maybe_unused]] float scan_m = 0.f, scan_a = 0.f;
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            const unsigned u = raw[s];
            const c2 f = make_c2((float)(u & 255u), (float)((u >> 8) & 255u));
            if constexpr (GJ_W_SCANSUMS != 0 && HS) {
                if (s >= 8) {
                    const c2 t = cadd(f, khalf);
                    const float r2 = fmaf(t.x, t.x, t.y * t.y);   // |z|^2, exact
                    scan_m += r2;
                    scan_a += __fsqrt_rn(r2);
                }
            }
            if constexpr (Cfg::win16)   // w (2u - 255) = (u - 127.5) (2w)
                v[s] = (s & 1) ? scale_hi(cadd(f, khalf), w2p[s >> 1]) : scale_lo(cadd(f, khalf), w2p[s >> 1]);
            else
                v[s] = (s & 1) ? fma_hi(f, w2p[s >> 1], wcp[s >> 1]) : fma_lo(f, w2p[s >> 1], wcp[s >> 1]);
            if (!HS || s >= 8) fsum = cadd(fsum, f);
        }
        if constexpr (GJ_W_PREFETCH && GJ_W_RAWREUSE && HS) {
            const unsigned nseg_idx = (seg + B < seg_hi) ? seg + B : seg_lo;
            const bool consecutive = seg + B < seg_hi;
            const unsigned byte0 = (nseg_idx * (unsigned)(N / 2) + (unsigned)jl0) * 2u;
#pragma unroll
            for (int s = 0; s < 8; ++s) raw[s] = raw[s + 8];
            if (!consecutive) {   // wrap to the dummy segment: its first half must be loaded after all (values unused)
#pragma unroll
                for (int s = 0; s < 8; ++s)
                    raw[s] = GJ_LOAD_RAW(*reinterpret_cast<const uint16_t*>(chunk8 + (byte0 + 2u * TF * s)));
            }
#pragma unroll
            for (int s = 8; s < 16; ++s)
                raw[s] = GJ_LOAD_RAW(*reinterpret_cast<const uint16_t*>(chunk8 + (byte0 + 2u * TF * s)));
        } else if (GJ_W_PREFETCH) {
            load_step(raw, (seg + B < seg_hi) ? seg + B : seg_lo);
        }
        float si, sq;
        if constexpr (TF >= 64) {
            si = group_sum_dpp_f<64>(fsum.x);   // wave-uniform
            sq = group_sum_dpp_f<64>(fsum.y);
            if ((tid & 63) == 0) {
                wsum[cur][b][(tid >> 6) % WPF][0] = si;
                wsum[cur][b][(tid >> 6) % WPF][1] = sq;
            }
        } else {
            si = group_sum_dpp_f<TF>(fsum.x);
            sq = group_sum_dpp_f<TF>(fsum.y);
        }

        if constexpr (GJ_W_SCANSUMS != 0 && HS) {
            const int mi = group_sum_dpp<64>((int)(4.0f * scan_m));
            const float ai = group_sum_dpp_f<64>(scan_a);
            if ((tid & 63) == 0) {
                float* dst = partial + (size_t)gridDim.x * N + ((size_t)blockIdx.x * 64 + (it & 63)) * 8 + (tid >> 6) * 2;
                dst[0] = __int_as_float(mi);
                dst[1] = ai;
            }
        }
        GJ_STAMP(t_it1);
        GJ_STAMP_ADD(4, t_it0, t_it1);   // unpack + window + sums (+ waiting for the prefetched loads)
        if constexpr (XP) welch_passes_x4096(v, lds0, lds1, tid, tw, ktw, stamps);
        else welch_passes<N, 0>(v, lds0, lds1, it, b * lds_span(N), jl, tw, ktw, stamps);
        GJ_STAMP(t_it2);

        // detrend in the frequency domain on bins 0, 1, N-1
        if constexpr (TF >= 64) {
            if (jl <= 1 || jl == TF - 1) {
                si = 0.f; sq = 0.f;
#pragma unroll
                for (int k = 0; k < WPF; ++k) { si += wsum[cur][b][k][0]; sq += wsum[cur][b][k][1]; }
                if constexpr (HS) {
#pragma unroll
                    for (int k = 0; k < WPF; ++k) { si += wsum[prv][b][k][0]; sq += wsum[prv][b][k][1]; }
                }
            }
        }
        const float Sx = fmaf(2.0f, si, -g.off2 * N), Sy = fmaf(2.0f, sq, -g.off2 * N);   // sum of (2u - off2)
        if (jl == WelchBins<N>::jl(0)) {
            v[WelchBins<N>::slot(0)].x -= 0.5f * Sx;
            v[WelchBins<N>::slot(0)].y -= 0.5f * Sy;
        }
        if (jl == WelchBins<N>::jl(1)) {
            v[WelchBins<N>::slot(1)].x += 0.25f * Sx;
            v[WelchBins<N>::slot(1)].y += 0.25f * Sy;
        }
        if (jl == WelchBins<N>::jl(N - 1)) {
            v[WelchBins<N>::slot(N - 1)].x += 0.25f * Sx;
            v[WelchBins<N>::slot(N - 1)].y += 0.25f * Sy;
        }
        if (active) {
#pragma unroll
            for (int s = 0; s < 16; ++s) {
                if constexpr (Cfg::pkacc) acc_sq(accp[s], v[s]);
                else accs[s] = fmaf(v[s].x, v[s].x, fmaf(v[s].y, v[s].y, accs[s]));
            }
        }
        GJ_STAMP(t_it3);
        GJ_STAMP_ADD(5, t_it2, t_it3);   // detrend fix + |X|^2
        if constexpr (HS) { ws_prv = ws_cur; ws_cur = (ws_cur == 2) ? 0 : ws_cur + 1; }
    }
#ifdef GJ_STAMPS
    {
        GJ_STAMP(t_end);
        stamps[6] = t_end - t_begin;
        stamps[7] = nsteps;
        if ((tid & 63) == 0)
            for (int k = 0; k < 8; ++k) atomicAdd(&g_welch_stamps[k], stamps[k]);
    }
#endif
    float* out = partial + ((size_t)blockIdx.x * B + b) * N + jl;
#pragma unroll
    for (int s = 0; s < 16; ++s) {
        if constexpr (Cfg::pkacc) out[TF * s] = accp[s].x + accp[s].y;
        else out[TF * s] = accs[s];
    }
}

// one bin per thread: used when the caller's output arrays are not 16-byte aligned
__global__ __launch_bounds__(256) void welch_finalize_scalar_kernel(const float* __restrict__ partial, int n,
                                                                    unsigned per_chunk, unsigned nchunks, float scale_full,
                                                                    float scale_last, int shift, float* __restrict__ psd,
                                                                    float* __restrict__ psd_db) {
    const unsigned c = blockIdx.y;
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    const float* p = partial + (size_t)c * per_chunk * n + k;
    float s = 0.f;
    for (unsigned i = 0; i < per_chunk; ++i) s += p[(size_t)i * n];
    const float val = s * ((c + 1 == nchunks) ? scale_last : scale_full);
    const int o = shift ? ((k + n / 2) & (n - 1)) : k;
    psd[(size_t)c * n + o] = val;
    if (psd_db) psd_db[(size_t)c * n + o] = 10.0f * log10f(val + 1e-15f);
}

// four consecutive bins per thread: 16-byte loads of the partial rows, 16-byte stores
__global__ __launch_bounds__(256) void welch_finalize_kernel(const float* __restrict__ partial, int n, unsigned per_chunk,
                                                             unsigned nchunks, float scale_full, float scale_last,
                                                             int shift, float* __restrict__ psd,
                                                             float* __restrict__ psd_db) {
    const unsigned c = blockIdx.y;
    const int k = 4 * (blockIdx.x * blockDim.x + threadIdx.x);
    if (k >= n) return;
    const float4* p = reinterpret_cast<const float4*>(partial + (size_t)c * per_chunk * n + k);
    const size_t stride = (size_t)n / 4;
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    for (unsigned i = 0; i < per_chunk; ++i) {   // fixed order: bit-identical from run to run
        const float4 q = p[(size_t)i * stride];
        s.x += q.x; s.y += q.y; s.z += q.z; s.w += q.w;
    }
    const float sc = (c + 1 == nchunks) ? scale_last : scale_full;
    const float4 val = make_float4(s.x * sc, s.y * sc, s.z * sc, s.w * sc);
    const int o = shift ? ((k + n / 2) & (n - 1)) : k;   // n/2 is a multiple of 4: the group stays contiguous
    *reinterpret_cast<float4*>(psd + (size_t)c * n + o) = val;
    if (psd_db)
        *reinterpret_cast<float4*>(psd_db + (size_t)c * n + o) =
            make_float4(10.0f * log10f(val.x + 1e-15f), 10.0f * log10f(val.y + 1e-15f), 10.0f * log10f(val.z + 1e-15f),
                        10.0f * log10f(val.w + 1e-15f));
}

struct WelchPlan {
    WelchGeom g;
    size_t rows;
    int batch;
    size_t ws_bytes;
    double scale_full, scale_last;
};

static bool welch_plan(gj_ctx* ctx, size_t nbytes, size_t chunk_samples, int nperseg, double fs, WelchPlan& pl) {
    if (nperseg < 16 || nperseg > 4096 || (nperseg & (nperseg - 1))) return false;
    if (chunk_samples < (size_t)nperseg) return false;
    // the kernel addresses a chunk with 32-bit byte offsets from a 64-bit chunk base
    if (2ull * chunk_samples + 2ull * (unsigned long long)nperseg > (1ull << 32)) return false;
    pl.rows = gj_welch_rows(nbytes, chunk_samples, nperseg);
    pl.batch = kBlockPoints / nperseg;
    pl.g.chunk_samples = chunk_samples;
    pl.g.neg_off = -0.5f * (float)ctx->off2;
    pl.g.off2 = (float)ctx->off2;
    pl.g.nchunks = (unsigned)pl.rows;
    const size_t step = nperseg / 2;
    pl.g.nseg_full = (unsigned)((chunk_samples - nperseg) / step + 1);
    size_t last_len = chunk_samples;
    if (pl.rows) {
        const size_t total = nbytes / 2;
        const size_t rem = total - (pl.rows - 1) * chunk_samples;
        last_len = rem < chunk_samples ? rem : chunk_samples;
    }
    pl.g.nseg_last = (unsigned)((last_len - nperseg) / step + 1);
    // Workgroups per chunk: 2 workgroups are resident per CU (VGPR-limited), the grid runs in
    // ceil(workgroups / slots) rounds of about (steps per workgroup + start-up) each, and a
    // nearly empty last round is pure loss -- pick the split that minimises rounds x length.
    const size_t slots = (size_t)ctx->num_cus * (welch_occ3(nperseg) ? 3 : 2);
    size_t cap = pl.g.nseg_full / (2 * (size_t)pl.batch);   // at least ~2 steps per workgroup
    if (cap < 1) cap = 1;
    if (cap > 256) cap = 256;
    size_t want = 1;
    double best = 1e300;
    for (size_t sp = 1; sp <= cap; ++sp) {
        const size_t wgs = (pl.rows ? pl.rows : 1) * sp;
        const double rounds = (double)((wgs + slots - 1) / slots);
        const double steps = (double)pl.g.nseg_full / (double)(sp * pl.batch) + 1.5;   // 1.5: twiddle/window set-up
        const double cost = rounds * steps;
        if (cost < best * 0.999) { best = cost; want = sp; }
    }
    pl.g.splits = (unsigned)want;
    pl.ws_bytes = pl.rows * want * (size_t)kBlockPoints * sizeof(float) + (GJ_W_SCANSUMS ? pl.rows * want * 64 * 8 * sizeof(float) : 0);
    const double sw2 = 0.375 * nperseg;   // sum of the squared periodic Hann window
    const double norm2 = unpack_norm2(ctx);   // the kernel works on 2u - off2 = sample * (2 / scale): 65025 by default
    pl.scale_full = 1.0 / (fs * sw2 * norm2 * (double)pl.g.nseg_full);
    pl.scale_last = 1.0 / (fs * sw2 * norm2 * (double)pl.g.nseg_last);
    return true;
}

#ifdef GJ_STAMPS
extern "C" int gj_debug_welch_stamps(unsigned long long* out, int reset) {
    if (out && hipMemcpyFromSymbol(out, HIP_SYMBOL(g_welch_stamps), sizeof(unsigned long long) * 8) != hipSuccess) return -1;
    if (reset) {
        unsigned long long z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        if (hipMemcpyToSymbol(HIP_SYMBOL(g_welch_stamps), z, sizeof(z)) != hipSuccess) return -1;
    }
    return 0;
}
#endif

size_t welch_workspace(gj_ctx* ctx, size_t nbytes, size_t chunk_samples, int nperseg) {
    WelchPlan pl;
    if (!welch_plan(ctx, nbytes, chunk_samples, nperseg, 1.0, pl)) return 0;
    return pl.ws_bytes;
}

extern const float* window_table(gj_ctx* ctx, int n);

template <int N>
static void welch_launch(gj_ctx* ctx, const uint8_t* d_iq, const WelchPlan& pl, float* partial) {
    hipLaunchKernelGGL(welch_kernel<N>, dim3(pl.g.nchunks * pl.g.splits), dim3(kBlockThreads), 0, ctx->stream, d_iq,
                       pl.g, ctx->d_twiddle, window_table(ctx, N), partial);
}

int launch_welch(gj_ctx* ctx, const uint8_t* d_iq, size_t nbytes, size_t chunk_samples, int nperseg, double fs,
                 int flags, float* d_psd, float* d_psd_db) {
    WelchPlan pl;
    if (!(fs > 0.0)) return fail(ctx, GJ_ERR_INVALID, "fs must be > 0");
    if (!welch_plan(ctx, nbytes, chunk_samples, nperseg, fs, pl))
        return fail(ctx, GJ_ERR_UNSUPPORTED, "nperseg must be a power of two in [16, 4096] and <= chunk_samples, chunk_samples < 2^31 - nperseg");
    if ((reinterpret_cast<uintptr_t>(d_iq) & 1) != 0) return fail(ctx, GJ_ERR_INVALID, "capture must be 2-byte aligned");
    if (pl.rows == 0) return GJ_OK;
    if ((unsigned long long)pl.g.nchunks * pl.g.splits > 0x7fffffffull) return fail(ctx, GJ_ERR_UNSUPPORTED, "too many chunks");
    int rc = ensure_workspace(ctx, pl.ws_bytes);
    if (rc) return rc;
    float* partial = reinterpret_cast<float*>(ctx->ws);
    switch (nperseg) {
        case 16: welch_launch<16>(ctx, d_iq, pl, partial); break;
        case 32: welch_launch<32>(ctx, d_iq, pl, partial); break;
        case 64: welch_launch<64>(ctx, d_iq, pl, partial); break;
        case 128: welch_launch<128>(ctx, d_iq, pl, partial); break;
        case 256: welch_launch<256>(ctx, d_iq, pl, partial); break;
        case 512: welch_launch<512>(ctx, d_iq, pl, partial); break;
        case 1024: welch_launch<1024>(ctx, d_iq, pl, partial); break;
        case 2048: welch_launch<2048>(ctx, d_iq, pl, partial); break;
        default: welch_launch<4096>(ctx, d_iq, pl, partial); break;
    }
    GJ_LAUNCH_CHECK(ctx);
    const bool aligned = ((reinterpret_cast<uintptr_t>(d_psd) | reinterpret_cast<uintptr_t>(d_psd_db)) & 15) == 0;
    if (aligned)
        hipLaunchKernelGGL(welch_finalize_kernel, dim3((nperseg / 4 + 255) / 256, pl.g.nchunks), dim3(256), 0, ctx->stream,
                           partial, nperseg, pl.g.splits * (unsigned)pl.batch, pl.g.nchunks, (float)pl.scale_full,
                           (float)pl.scale_last, (flags & GJ_WELCH_SHIFT) ? 1 : 0, d_psd, d_psd_db);
    else
        hipLaunchKernelGGL(welch_finalize_scalar_kernel, dim3((nperseg + 255) / 256, pl.g.nchunks), dim3(256), 0,
                           ctx->stream, partial, nperseg, pl.g.splits * (unsigned)pl.batch, pl.g.nchunks,
                           (float)pl.scale_full, (float)pl.scale_last, (flags & GJ_WELCH_SHIFT) ? 1 : 0, d_psd, d_psd_db);
    GJ_LAUNCH_CHECK(ctx);
    return GJ_OK;
}

}   // namespace gj
