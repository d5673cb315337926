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
// (Figures of rounds 2-5.  Below 4096 they INCLUDED a finalize that added B = 4096 / N partial rows per workgroup -- at 16
// points 256 of them, a launch longer than the transform.  With one partial row per workgroup (round 6) K2 + finalize
// per GiB is 0.43 / 0.63 / 0.68 / 0.73 / 0.76 / 0.94 / 0.99 / 1.15 / 1.09 ms at 16 / 32 / ... / 4096 points
// (profiles/r06_k2_all_sizes.txt); the two-against-three-workgroup comparison itself has not been repeated.)
// (512..2048 fit 168 VGPRs only since the exchange addresses are written as base + constant,
// fft_core.h lds_scatter/lds_gather: 236-246 -> 192-206 VGPRs in the two-workgroup shape.)
#define GJ_W_OCC3_MASK 0x1FE0u
#endif
#ifndef GJ_W_HALFSUM
#define GJ_W_HALFSUM 1   // 1: (one transform per workgroup) half-segment sums carried from step to step
#endif
#ifndef GJ_W_CARRY
// 1: (one transform per workgroup, i.e. N = 4096) the unpacked second half of a segment is CARRIED in registers as
// the first half of the next one (50 % overlap), and the periodic Hann window's symmetry w[n + N/2] = 1 - w[n]
// halves the window registers that pays for it: per step 16 conversions, 8 offset FMAs and 8 loads fewer
// (72 -> 48 front-end instructions per thread), same VGPR count.
#define GJ_W_CARRY 1
#endif
#ifndef GJ_W_CARRY_MASK
#define GJ_W_CARRY_MASK 0x1400u   // bit k set: transform size 2^k carries.  1024 and 4096; 2048 spills with it (+2 %, measured)
#endif
#ifndef GJ_W_WAVEFENCE
// 1: for N <= 1024 the LDS exchange is ordered by a wavefront fence (a transform lies inside one wave); 0: by the
// workgroup barrier, as for the larger sizes.  The second form is built as libgpsjam_hip_barrier.so (Makefile) and
// tests/test_round5_gpu.py compares the two byte for byte: the fence path must never depend on a barrier it removed.
#define GJ_W_WAVEFENCE 1
#endif
#ifndef GJ_W_WIDELOAD
// 1: transforms of 16 and 32 points fetch their segment with 16-byte loads (2 or 4 of them) and pick the thread's sixteen
// samples out of the registers, instead of sixteen 2-byte loads at a 2 TF-byte stride.  Measured per GiB, interleaved on
// one box (profiles/r06_k2_small_sizes.txt): N = 32 0.95 -> 0.64 ms, N = 16 0.449 -> 0.441; the same for 64 points needs a
// run-time choice between register pairs, spills, and is three times SLOWER (0.70 -> 2.29 ms): 64 keeps the narrow loads.
#define GJ_W_WIDELOAD 1
#endif
#ifndef GJ_W_XPOSE
#define GJ_W_XPOSE 1     // 1: N = 4096 uses the bank-conflict-free exchange schedule (fft_core.h X4096)
#endif

namespace gj {

// 2 f + k  (k wave-uniform, in SGPRs): the unpack (2u - off2) of two components at once
__device__ __forceinline__ c2 twice_plus_k(c2 f, c2 k) {
    c2 r;
    asm("v_pk_fma_f32 %0, %1, 2.0, %2" : "=v"(r) : "v"(f), "s"(k));
    return r;
}
// a - a * p.x  /  a - a * p.y : the second-half window 1 - w applied without forming it
__device__ __forceinline__ c2 one_minus_lo(c2 a, c2 p) {
    c2 r;
    asm("v_pk_fma_f32 %0, %1, %2, %1 op_sel:[0,0,0] op_sel_hi:[1,0,1] neg_lo:[0,1,0] neg_hi:[0,1,0]" : "=v"(r) : "v"(a), "v"(p));
    return r;
}
__device__ __forceinline__ c2 one_minus_hi(c2 a, c2 p) {
    c2 r;
    asm("v_pk_fma_f32 %0, %1, %2, %1 op_sel:[0,1,0] op_sel_hi:[1,1,1] neg_lo:[0,1,0] neg_hi:[0,1,0]" : "=v"(r) : "v"(a), "v"(p));
    return r;
}

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
// `after_scatter0()` runs right behind the first scatter (see welch_passes_x4096)
// The exchange barrier.  For N <= 1024 a transform's N / 16 threads lie inside ONE wave, whose LDS instructions execute
// in issue order: the writes of a scatter are seen by the gather behind them without any workgroup barrier.  What is
// needed is only that the COMPILER keeps the two in order (per lane they touch different addresses): a wavefront-scope
// fence.  The four (or more) transforms a workgroup works on then run independently instead of waiting for each other
// four times per segment (round 4: welch_kernel<1024> is the reference's own FFT size, skrypty/widmo_plot.py:10).
template <int N>
__device__ __forceinline__ void welch_exchange_sync() {
    if constexpr (N / 16 <= 64 && GJ_W_WAVEFENCE != 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    } else {
        __syncthreads();
    }
}

template <int N, int PASS, typename Mid>
__device__ __forceinline__ void welch_passes(c2 (&v)[16], cf* lds0, cf* lds1, unsigned it, int base, int jl,
                                             const c2 (&tw)[3][15], const InnerTw& ktw, Mid&& after_scatter0) {
    constexpr int NP = fft_npass(N);
    fft_pass<N, PASS, GJ_W_TWOSTEP != 0, GJ_W_FMA != 0>(v, tw[PASS], ktw);
    if constexpr (PASS + 1 < NP) {
        // exchanges per segment: NP-1.  Even count -> parity of PASS; odd count -> parity of (it + PASS)
        const bool second = ((NP - 1) % 2 == 0) ? (PASS & 1) : ((it + PASS) & 1);
        cf* lds = (WelchCfg<N>::dbuf && second) ? lds1 : lds0;
        lds_scatter<N, PASS>(v, lds, base, jl);
        if constexpr (PASS == 0) after_scatter0();
        welch_exchange_sync<N>();
        lds_gather<N>(v, lds, base, jl);
        if (!WelchCfg<N>::dbuf) welch_exchange_sync<N>();
        welch_passes<N, PASS + 1>(v, lds0, lds1, it, base, jl, tw, ktw, after_scatter0);
    }
}

// N = 4096 with the conflict-free exchange schedule of fft_core.h (X4096): pass 0 in role
// jl0 = tid, passes 1 and 2 in role jl1; buffer 0 carries exchange 0, buffer 1 exchange 1.
// `after_scatter0()` runs where the fewest registers are live (the points are in LDS, the next pass has not
// gathered them yet): the caller issues its prefetch loads there.
template <typename Mid>
__device__ __forceinline__ void welch_passes_x4096(c2 (&v)[16], cf* lds0, cf* lds1, int tid, const c2 (&tw)[3][15],
                                                   const InnerTw& ktw, Mid&& after_scatter0) {
    fft_pass<4096, 0, false, GJ_W_FMA != 0>(v, tw[0], ktw);
    x4096_scatter<0>(v, lds0, tid);
    after_scatter0();
    __syncthreads();
    x4096_gather<0>(v, lds0, tid);
    if (!WelchCfg<4096>::dbuf) __syncthreads();
    fft_pass<4096, 1, GJ_W_TWOSTEP != 0, GJ_W_FMA != 0>(v, tw[1], ktw);
    cf* ldsx = WelchCfg<4096>::dbuf ? lds1 : lds0;
    x4096_scatter<1>(v, ldsx, tid);
    __syncthreads();
    x4096_gather<1>(v, ldsx, tid);
    if (!WelchCfg<4096>::dbuf) __syncthreads();
    fft_pass<4096, 2, GJ_W_TWOSTEP != 0, GJ_W_FMA != 0>(v, tw[2], ktw);
}

// Several captures of ONE size in one launch (gj_welch_batch_dev: the reference's deployment is three antenna files of one
// length, worker.py:97-101): chunk c of the launch is chunk c % rows_each of capture c / rows_each.  The single-capture
// instantiation (BATCHED = false) is the kernel as it was: nothing of this reaches its code.
struct WelchBatch {
    const uint8_t* iq[GJ_MAX_ANTENNAS];
    unsigned rows_each;
};

template <int N, bool BATCHED = false>
__global__ __launch_bounds__(kBlockThreads, WelchCfg<N>::min_waves) void welch_kernel(const uint8_t* __restrict__ iq, WelchGeom g,
                                                              const cf* __restrict__ twtab,
                                                              const float* __restrict__ wintab,
                                                              float* __restrict__ partial, unsigned wg_base,
                                                              WelchBatch batch = WelchBatch()) {
    // wg_base: index of this launch's first workgroup in the whole capture's grid (0 unless the capture is
    // transformed piece by piece while it is still being uploaded, gj_ingest_*)
    constexpr int TF = N / 16, B = kBlockPoints / N, NP = fft_npass(N);
    constexpr int WPF = (TF >= 64) ? TF / 64 : 1;   // waves per transform
    constexpr bool XP = (N == 4096) && GJ_W_XPOSE;
    constexpr int SPAN = XP ? X4096::kSpan : lds_span(kBlockPoints);
    __shared__ cf lds0[SPAN];
    using Cfg = WelchCfg<N>;
    // RUNS (transform groups of one wave or more, N >= 1024): group b of the workgroup walks a run of CONSECUTIVE
    // segments (b-th slice of the workgroup's segments) instead of taking every B-th one, so that every step's first
    // half segment is the previous step's second half for ALL these sizes, not only for N = 4096 (B = 1): the
    // half-segment sums (HS) and the unpacked half (CARRY) are then carried from step to step.  Smaller transforms keep
    // the interleaved order: their groups are fractions of a wave and consecutive groups read consecutive memory.
    constexpr bool RUNS = (TF >= 64) && (GJ_W_HALFSUM != 0);
    constexpr bool HS = RUNS;
    __shared__ cf lds1[Cfg::dbuf ? SPAN : 1];
    // (sum I, sum Q) per wave; three slots when half-segment sums are carried over (see HS below)
    __shared__ float wsum[3][B][WPF][2];
    // What welch_exchange_sync<N>'s wave fence (N <= 1024) rests on: a transform group is an aligned
    // fraction of ONE wave -- group index tid / TF with TF | 64 -- so every LDS word a lane gathers was scattered by a
    // lane of its own wave; and the group's (sum I, sum Q) slot wsum[.][b][0] is written by a lane of that wave and read
    // only by lanes of it (WPF == 1).  Changing the block size, the thread-to-group mapping or adding a cross-wave LDS
    // use for these sizes must bring the workgroup barrier back.
    static_assert(TF > 64 || (64 % TF == 0 && WPF == 1 && kBlockThreads % 64 == 0 && B * TF == kBlockThreads),
                  "the wave-fence exchange needs a transform group inside one wave");
    const int tid = threadIdx.x;
    // a transform group of one wave or more: its index is wave-uniform, so the run bookkeeping lives in SGPRs
    const int b = (TF >= 64) ? __builtin_amdgcn_readfirstlane(tid / TF) : tid / TF;
    const int jl0 = tid % TF;   // jl0: butterfly of pass 0 (input index jl0 + TF s)
    const int jl = XP ? X4096::jl1(tid) : jl0;   // butterfly of the later passes = bins held at the end
    const unsigned wg = blockIdx.x + wg_base;
    unsigned c = wg / g.splits;
    const unsigned part = wg % g.splits;
    if constexpr (BATCHED) {
        const unsigned cap = c / batch.rows_each;
        iq = batch.iq[cap];
        c -= cap * batch.rows_each;
    }
    const unsigned nseg = (c + 1 == (BATCHED ? batch.rows_each : g.nchunks)) ? g.nseg_last : g.nseg_full;
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

    // CARRY (every group walks consecutive segments, N >= 1024): see GJ_W_CARRY above
    constexpr bool CARRY = HS && Cfg::win16 && (GJ_W_CARRY != 0) && (GJ_W_PREFETCH != 0) &&
                           (((GJ_W_CARRY_MASK >> welch_log2(N)) & 1u) != 0);
    // window folded into the unpack: w (2u - 255) = u (2w) + (-255 w); (w[2i], w[2i+1]) share a
    // VGPR pair and op_sel picks the half, so 16 points cost 16 register pairs.
    // CARRY: w[n + N/2] = 1 - w[n] (periodic Hann), so the eight values of the first half do for both.
    constexpr int NW = CARRY ? 4 : 8;
    c2 w2p[NW], wcp[Cfg::win16 ? 1 : 8];
#pragma unroll
    for (int s = 0; s < NW; ++s) {
        const float wa = wintab[jl0 + TF * (2 * s)], wb = wintab[jl0 + TF * (2 * s + 1)];
        w2p[s] = CARRY ? make_c2(wa, wb) : make_c2(2.0f * wa, 2.0f * wb);
        if constexpr (!Cfg::win16) wcp[s] = make_c2(-g.off2 * wa, -g.off2 * wb);
    }
    [[maybe_unused]] const c2 khalf = make_c2(g.neg_off, g.neg_off);
    [[maybe_unused]] const c2 kmoff = make_c2(-g.off2, -g.off2);
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
        if constexpr (GJ_W_WIDELOAD != 0 && TF <= 2) {
            // the whole segment (2 N bytes) in 16-byte loads; sample jl0 + TF s is one half of dword (jl0 + TF s) / 2.
            // The segment is only 2-byte aligned in general: global memory takes unaligned vector loads.
            struct __attribute__((packed, aligned(2))) Vec16 { unsigned x, y, z, w; };
            constexpr int NV = 2 * N / 16;
            const Vec16* src = reinterpret_cast<const Vec16*>(chunk8 + seg_idx * (unsigned)N);
            unsigned w[4 * NV];
#pragma unroll
            for (int v = 0; v < NV; ++v) {
                const Vec16 g = src[v];
                w[4 * v] = g.x; w[4 * v + 1] = g.y; w[4 * v + 2] = g.z; w[4 * v + 3] = g.w;
            }
#pragma unroll
            for (int s = 0; s < 16; ++s) {
                if constexpr (TF == 1) dst[s] = (s & 1) ? (w[s >> 1] >> 16) : (w[s >> 1] & 0xffffu);
                else dst[s] = (w[s] >> (16u * (unsigned)jl0)) & 0xffffu;
            }
        } else {
            const unsigned byte0 = (seg_idx * (unsigned)(N / 2) + (unsigned)jl0) * 2u;
#pragma unroll
            for (int s = 0; s < 16; ++s)
                dst[s] = (*reinterpret_cast<const uint16_t*>(chunk8 + (byte0 + 2u * TF * s)));
        }
    };
    // samples s0 .. s0+7 of segment seg_idx (one half segment)
    auto load_half = [&](unsigned (&dst)[8], unsigned seg_idx, int s0) {
        const unsigned byte0 = (seg_idx * (unsigned)(N / 2) + (unsigned)jl0) * 2u;
#pragma unroll
        for (int s = 0; s < 8; ++s)
            dst[s] = (*reinterpret_cast<const uint16_t*>(chunk8 + (byte0 + 2u * TF * (s0 + s))));
    };
    const unsigned nsteps = (seg_hi - seg_lo + B - 1) / B;
    // this transform group's segments: [my_lo, my_hi) one per step (RUNS), or seg_lo + b, + B, ... (interleaved)
    const unsigned my_lo = RUNS ? ((seg_lo + b * nsteps < seg_hi) ? seg_lo + b * nsteps : seg_hi) : seg_lo;
    const unsigned my_hi = RUNS ? ((my_lo + nsteps < seg_hi) ? my_lo + nsteps : seg_hi) : seg_hi;
    const unsigned seg_idle = (my_lo < seg_hi) ? my_lo : seg_lo;   // what a group without work loads (values unused)
    // raw samples of the NEXT step are fetched while the current one is transformed
    unsigned raw[CARRY ? 1 : 16];
    unsigned rawh[CARRY ? 8 : 1];      // CARRY: only the new half segment is ever loaded
    c2 carry[CARRY ? 8 : 1];           // CARRY: 2u - off2 of the previous step's second half = this step's first half
    [[maybe_unused]] unsigned ws_cur = 0, ws_prv = 2;   // HS: slot of this step's half-sum / of the previous step's
    if constexpr (CARRY) {
        // first step of the group's run: the first half has no predecessor -- unpack it here, once
        load_half(rawh, seg_idle, 0);
        c2 flo = make_c2(0.f, 0.f);
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            const c2 f = make_c2((float)(rawh[s] & 255u), (float)((rawh[s] >> 8) & 255u));
            flo = cadd(flo, f);
            carry[s] = twice_plus_k(f, kmoff);
        }
        float li, lq;
        wave_sum_pair_u16(flo.x, flo.y, li, lq);
        if ((tid & 63) == 0) {
            wsum[ws_prv][b][(tid >> 6) % WPF][0] = li;
            wsum[ws_prv][b][(tid >> 6) % WPF][1] = lq;
        }
        load_half(rawh, seg_idle, 8);
    } else {
        load_step(raw, RUNS ? seg_idle : ((seg_lo + b < seg_hi) ? seg_lo + b : seg_lo));
    }
    for (unsigned it = 0; it < nsteps; ++it) {
        const unsigned seg = RUNS ? my_lo + it : seg_lo + it * B + b;
        const bool active = seg < my_hi;
        const unsigned seg_next = RUNS ? ((seg + 1 < my_hi) ? seg + 1 : seg_idle) : ((seg + B < seg_hi) ? seg + B : seg_lo);
        c2 v[16];
        c2 fsum = make_c2(0.f, 0.f);   // (sum I, sum Q) of the raw bytes: integers < 2^24, exact in f32
        // HS (one transform per workgroup = consecutive segments per step): the first half of a
        // segment is the second half of the previous one, so only the second half is summed each
        // step and the previous step's half-sum is read back from its slot (eight packed adds less)
        const unsigned cur = HS ? ws_cur : (it & 1), prv = HS ? ws_prv : 0;
        if constexpr (CARRY) {
#pragma unroll
            for (int s = 0; s < 8; ++s)   // first half: carried, already 2u - off2
                v[s] = (s & 1) ? scale_hi(carry[s], w2p[s >> 1]) : scale_lo(carry[s], w2p[s >> 1]);
#pragma unroll
            for (int s = 0; s < 8; ++s) {   // second half: unpack, keep for the next step, window 1 - w
                const unsigned u = rawh[s];
                const c2 f = make_c2((float)(u & 255u), (float)((u >> 8) & 255u));
                fsum = cadd(fsum, f);
                const c2 u2 = twice_plus_k(f, kmoff);
                carry[s] = u2;
                v[8 + s] = (s & 1) ? one_minus_hi(u2, w2p[s >> 1]) : one_minus_lo(u2, w2p[s >> 1]);
            }
        } else {
            if (!GJ_W_PREFETCH && it > 0) load_step(raw, active ? seg : seg_idle);
            if (HS && it == 0) {   // first step of the workgroup: the first half has no predecessor
                c2 flo = make_c2(0.f, 0.f);
#pragma unroll
                for (int s = 0; s < 8; ++s) flo = cadd(flo, make_c2((float)(raw[s] & 255u), (float)((raw[s] >> 8) & 255u)));
                float li, lq;
                wave_sum_pair_u16(flo.x, flo.y, li, lq);
                if ((tid & 63) == 0) {
                    wsum[prv][b][(tid >> 6) % WPF][0] = li;
                    wsum[prv][b][(tid >> 6) % WPF][1] = lq;
                }
            }
#pragma unroll
            for (int s = 0; s < 16; ++s) {
                const unsigned u = raw[s];
                const c2 f = make_c2((float)(u & 255u), (float)((u >> 8) & 255u));
                if constexpr (Cfg::win16)   // w (2u - 255) = (u - 127.5) (2w)
                    v[s] = (s & 1) ? scale_hi(cadd(f, khalf), w2p[s >> 1]) : scale_lo(cadd(f, khalf), w2p[s >> 1]);
                else
                    v[s] = (s & 1) ? fma_hi(f, w2p[s >> 1], wcp[s >> 1]) : fma_lo(f, w2p[s >> 1], wcp[s >> 1]);
                if (!HS || s >= 8) fsum = cadd(fsum, f);
            }
            if (GJ_W_PREFETCH) load_step(raw, seg_next);
        }
        float si, sq;
        if constexpr (TF >= 64) {
            wave_sum_pair_u16(fsum.x, fsum.y, si, sq);   // wave-uniform, exact (at most 16 x 255 per lane)
            if ((tid & 63) == 0) {
                wsum[cur][b][(tid >> 6) % WPF][0] = si;
                wsum[cur][b][(tid >> 6) % WPF][1] = sq;
            }
        } else {
            si = group_sum_dpp_f<TF>(fsum.x);
            sq = group_sum_dpp_f<TF>(fsum.y);
        }

        // CARRY: the next step's new half, asked for while this step's points sit in LDS (about two thirds of a step
        // = 2 us ahead of their use, and where the fewest registers are live)
        auto prefetch_half = [&] {
            if constexpr (CARRY) load_half(rawh, seg_next, 8);
        };
        if constexpr (XP) welch_passes_x4096(v, lds0, lds1, tid, tw, ktw, prefetch_half);
        else welch_passes<N, 0>(v, lds0, lds1, it, b * lds_span(N), jl, tw, ktw, prefetch_half);

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
        if constexpr (HS) { ws_prv = ws_cur; ws_cur = (ws_cur == 2) ? 0 : ws_cur + 1; }
    }
    if constexpr (B == 1) {
        float* out = partial + (size_t)wg * N + jl;
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            if constexpr (Cfg::pkacc) out[TF * s] = accp[s].x + accp[s].y;
            else out[TF * s] = accs[s];
        }
    } else {
        // N < 4096: the workgroup's B transform groups each hold a spectrum.  They are added HERE, in group order, through
        // the exchange buffer (its last use is behind every group: all of them run the same number of steps), and the
        // workgroup writes ONE partial row (round 6).  Until then it wrote B rows: at nperseg 1024 four times the partial
        // traffic of the 4096 kernel per sample and four times the rows for the finalize to add -- 304 per chunk on the
        // reference's 10-s captures, a 12-us finalize behind a 160-us K2 (profiles/r06_deployment_timeline_graph.txt).
        __syncthreads();
        float* red = reinterpret_cast<float*>(lds0);
        static_assert((size_t)SPAN * sizeof(cf) >= (size_t)kBlockPoints * sizeof(float), "the exchange buffer holds one spectrum per group");
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            if constexpr (Cfg::pkacc) red[b * N + jl + TF * s] = accp[s].x + accp[s].y;
            else red[b * N + jl + TF * s] = accs[s];
        }
        __syncthreads();
        float* out = partial + (size_t)wg * N;
        for (int k = tid; k < N; k += kBlockThreads) {
            float t = red[k];
#pragma unroll
            for (int bb = 1; bb < B; ++bb) t += red[bb * N + k];
            out[k] = t;
        }
    }
}

// one bin per thread: used when the caller's output arrays are not 16-byte aligned
// Sum of a chunk's partial spectra, in ONE fixed order whatever the launch shape and whoever computes it (a part of a
// split capture sums the same rows the same way): four partial sums over the rows i = r, r + 4, r + 8, ... (ascending),
// r = 0..3, combined as (s0 + s1) + (s2 + s3).  Round 4: until then every thread added its column's rows one after
// the other -- 23 rows for the 1-GiB capture (19 us), but 300 for a 10-s capture, whose few chunks are cut into many
// runs to fill the chip: 80-108 us of dependent loads per launch, three times the K2 launch it finishes
// (profiles/r04_deployment_timeline.txt).  Now a workgroup is 64 columns x 4 row lanes (one wave per row lane: coalesced
// rows), eight loads in flight per thread.
__global__ __launch_bounds__(256) void welch_finalize_scalar_kernel(const float* __restrict__ partial, int n,
                                                                    unsigned per_chunk, unsigned nchunks, float scale_full,
                                                                    float scale_last, int shift, float* __restrict__ psd,
                                                                    float* __restrict__ psd_db) {
    __shared__ float sh[4][64];
    const unsigned c = blockIdx.y;
    const int cx = threadIdx.x & 63, ry = threadIdx.x >> 6;
    const int k = blockIdx.x * 64 + cx;
    float s = 0.f;
    if (k < n) {
        const float* p = partial + (size_t)c * per_chunk * n + k;
#pragma unroll 8
        for (unsigned i = ry; i < per_chunk; i += 4) s += p[(size_t)i * n];
    }
    sh[ry][cx] = s;
    __syncthreads();
    if (ry != 0 || k >= n) return;
    const float t = (sh[0][cx] + sh[1][cx]) + (sh[2][cx] + sh[3][cx]);
    const float val = t * ((c + 1 == nchunks) ? scale_last : scale_full);
    const int o = shift ? ((k + n / 2) & (n - 1)) : k;
    psd[(size_t)c * n + o] = val;
    if (psd_db) psd_db[(size_t)c * n + o] = 10.0f * log10f(val + 1e-15f);
}

// four consecutive bins per thread: 16-byte loads of the partial rows, 16-byte stores; the same order of summation
struct WelchOut {
    float* psd[GJ_MAX_ANTENNAS];
    unsigned rows_each;    // 0: one capture, rows contiguous in `psd` / `psd_db`
};

__global__ __launch_bounds__(256) void welch_finalize_kernel(const float* __restrict__ partial, int n, unsigned per_chunk,
                                                             unsigned nchunks, float scale_full, float scale_last,
                                                             int shift, float* __restrict__ psd,
                                                             float* __restrict__ psd_db, WelchOut outs = WelchOut()) {
    __shared__ float4 sh[4][64];
    unsigned c = blockIdx.y;
    const int cx = threadIdx.x & 63, ry = threadIdx.x >> 6;
    const int k = 4 * (blockIdx.x * 64 + cx);
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    if (k < n) {
        const float4* p = reinterpret_cast<const float4*>(partial + (size_t)c * per_chunk * n + k);
        const size_t stride = (size_t)n / 4;
#pragma unroll 8
        for (unsigned i = ry; i < per_chunk; i += 4) {
            const float4 q = p[(size_t)i * stride];
            s.x += q.x; s.y += q.y; s.z += q.z; s.w += q.w;
        }
    }
    sh[ry][cx] = s;
    __syncthreads();
    if (ry != 0 || k >= n) return;
    const float4 a0 = sh[0][cx], a1 = sh[1][cx], a2 = sh[2][cx], a3 = sh[3][cx];
    if (outs.rows_each) {               // batched: row c of the launch is row c % rows_each of capture c / rows_each
        const unsigned cap = c / outs.rows_each;
        psd = outs.psd[cap];
        c -= cap * outs.rows_each;
        nchunks = outs.rows_each;
    }
    const float sc = (c + 1 == nchunks) ? scale_last : scale_full;
    const float4 val = make_float4(((a0.x + a1.x) + (a2.x + a3.x)) * sc, ((a0.y + a1.y) + (a2.y + a3.y)) * sc,
                                   ((a0.z + a1.z) + (a2.z + a3.z)) * sc, ((a0.w + a1.w) + (a2.w + a3.w)) * sc);
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

// plan_bytes (a part of a split capture): the workgroups-per-chunk split is chosen as for a capture of plan_bytes,
// so that a chunk's partial spectra -- and with them the float sum behind its PSD row -- are cut the same way
// whether the chunk is processed as part of the whole capture or as part of a piece of it.
static bool welch_plan(gj_ctx* ctx, size_t nbytes, size_t chunk_samples, int nperseg, double fs, WelchPlan& pl,
                       size_t plan_bytes = 0) {
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
    const size_t plan_rows = plan_bytes ? gj_welch_rows(plan_bytes, chunk_samples, nperseg) : pl.rows;
    for (size_t sp = 1; sp <= cap; ++sp) {
        const size_t wgs = (plan_rows ? plan_rows : 1) * sp;
        const double rounds = (double)((wgs + slots - 1) / slots);
        const double steps = (double)pl.g.nseg_full / (double)(sp * pl.batch) + 1.5;   // 1.5: twiddle/window set-up
        const double cost = rounds * steps;
        if (cost < best * 0.999) { best = cost; want = sp; }
    }
    pl.g.splits = (unsigned)want;
    pl.ws_bytes = pl.rows * want * (size_t)nperseg * sizeof(float);   // one partial row per workgroup
    const double sw2 = 0.375 * nperseg;   // sum of the squared periodic Hann window
    const double norm2 = unpack_norm2(ctx);   // the kernel works on 2u - off2 = sample * (2 / scale): 65025 by default
    pl.scale_full = 1.0 / (fs * sw2 * norm2 * (double)pl.g.nseg_full);
    pl.scale_last = 1.0 / (fs * sw2 * norm2 * (double)pl.g.nseg_last);
    return true;
}

size_t welch_workspace(gj_ctx* ctx, size_t nbytes, size_t chunk_samples, int nperseg, size_t plan_bytes) {
    WelchPlan pl;
    if (!welch_plan(ctx, nbytes, chunk_samples, nperseg, 1.0, pl, plan_bytes)) return 0;
    return pl.ws_bytes;
}

extern const float* window_table(gj_ctx* ctx, int n);

template <int N>
static void welch_launch(gj_ctx* ctx, const uint8_t* d_iq, const WelchPlan& pl, float* partial, unsigned c0, unsigned c1) {
    hipLaunchKernelGGL(welch_kernel<N>, dim3((c1 - c0) * pl.g.splits), dim3(kBlockThreads), 0, ctx->stream, d_iq, pl.g,
                       ctx->d_twiddle, window_table(ctx, N), partial, c0 * pl.g.splits);
}

template <int N>
static void welch_launch_batch(gj_ctx* ctx, const WelchPlan& pl, float* partial, const WelchBatch& batch, unsigned n_captures) {
    hipLaunchKernelGGL((welch_kernel<N, true>), dim3(n_captures * batch.rows_each * pl.g.splits), dim3(kBlockThreads), 0, ctx->stream,
                       (const uint8_t*)nullptr, pl.g, ctx->d_twiddle, window_table(ctx, N), partial, 0u, batch);
}

// n captures of ONE length in one K2 launch + one finalize launch.  Every capture is planned, cut into workgroups and
// summed exactly as by launch_welch on its own (same plan, same per-chunk partial spectra, same fixed order), so each PSD
// is the same bits; what goes away is n - 1 launch gaps, n - 1 grid tails and n - 1 finalize launches -- which is what a
// step over the reference's 10-s captures consists of (profiles/r05_deployment_timeline_*.txt).
int launch_welch_batch(gj_ctx* ctx, const uint8_t* const* d_iq, int n_captures, size_t nbytes, size_t chunk_samples, int nperseg,
                       double fs, int flags, float* const* d_psd) {
    if (n_captures < 1 || n_captures > GJ_MAX_ANTENNAS) return fail(ctx, GJ_ERR_INVALID, "n_captures must be 1..%d", GJ_MAX_ANTENNAS);
    WelchJob job;
    int rc = welch_begin(ctx, nbytes, chunk_samples, nperseg, fs, 0, job);
    if (rc) return rc;
    if (job.rows == 0) return GJ_OK;
    WelchPlan pl;
    memcpy(&pl, job.plan, sizeof(pl));
    {
        // Measurement only (tools/k2_batch_sweep.py, profiles/r06_k2_batch_sweep.txt): force the workgroups per chunk of a
        // batched launch.  Round 6 tried planning the split over ALL the batch's chunks (three 10-s captures: 25 per chunk
        // = 750 workgroups of 40 steps instead of 76 per chunk = 2 280 of 13): K2 alone 4 % faster (122 against 127 us),
        // the deployment step it is part of 11 % SLOWER (0.231 against 0.206 ms, interleaved on one box) -- few long
        // workgroups keep the scan / tail / K5 kernels of the side chains waiting for a slot, many short ones let them in.
        // The per-capture plan stays, and with it the byte-equality of a PSD across every arrangement.
        static const long forced = [] { const char* e = getenv("GPSJAM_W_BATCH_SPLITS"); return e ? atol(e) : 0l; }();
        if (forced > 0 && forced <= 256 && (size_t)forced * 2 * (size_t)pl.batch <= pl.g.nseg_full) {
            pl.g.splits = (unsigned)forced;
            job.ws_bytes = pl.rows * (size_t)forced * (size_t)nperseg * sizeof(float);
        }
    }
    if ((unsigned long long)n_captures * pl.g.nchunks * pl.g.splits > 0x7fffffffull || (unsigned long long)n_captures * pl.g.nchunks > 65535ull)
        return fail(ctx, GJ_ERR_UNSUPPORTED, "too many chunks for one launch");
    WelchBatch batch;
    WelchOut outs;
    memset(&batch, 0, sizeof(batch));
    memset(&outs, 0, sizeof(outs));
    for (int a = 0; a < n_captures; ++a) {
        if (!d_iq[a] || !d_psd[a]) return fail(ctx, GJ_ERR_INVALID, "null buffer %d", a);
        if (reinterpret_cast<uintptr_t>(d_iq[a]) & 1) return fail(ctx, GJ_ERR_INVALID, "capture must be 2-byte aligned");
        if (reinterpret_cast<uintptr_t>(d_psd[a]) & 15) return fail(ctx, GJ_ERR_INVALID, "PSD rows must be 16-byte aligned");
        batch.iq[a] = d_iq[a];
        outs.psd[a] = d_psd[a];
    }
    batch.rows_each = outs.rows_each = pl.g.nchunks;
    rc = ensure_workspace(ctx, (size_t)n_captures * job.ws_bytes);
    if (rc) return rc;
    float* partial = reinterpret_cast<float*>(ctx->ws);
    const unsigned n = (unsigned)n_captures;
    switch (nperseg) {
        case 16: welch_launch_batch<16>(ctx, pl, partial, batch, n); break;
        case 32: welch_launch_batch<32>(ctx, pl, partial, batch, n); break;
        case 64: welch_launch_batch<64>(ctx, pl, partial, batch, n); break;
        case 128: welch_launch_batch<128>(ctx, pl, partial, batch, n); break;
        case 256: welch_launch_batch<256>(ctx, pl, partial, batch, n); break;
        case 512: welch_launch_batch<512>(ctx, pl, partial, batch, n); break;
        case 1024: welch_launch_batch<1024>(ctx, pl, partial, batch, n); break;
        case 2048: welch_launch_batch<2048>(ctx, pl, partial, batch, n); break;
        default: welch_launch_batch<4096>(ctx, pl, partial, batch, n); break;
    }
    GJ_LAUNCH_CHECK(ctx);
    hipLaunchKernelGGL(welch_finalize_kernel, dim3((nperseg / 4 + 63) / 64, n * pl.g.nchunks), dim3(256), 0, ctx->stream, partial,
                       nperseg, pl.g.splits, n * pl.g.nchunks, (float)pl.scale_full, (float)pl.scale_last,
                       (flags & GJ_WELCH_SHIFT) ? 1 : 0, (float*)nullptr, (float*)nullptr, outs);
    GJ_LAUNCH_CHECK(ctx);
    return GJ_OK;
}

// The transform of a capture in three steps, so that chunk ranges can be launched while later chunks are still on
// their way to HBM (gj_ingest_*): begin (plan; `partial` = where the per-workgroup spectra go), range (chunks
// [c0, c1)), end (fixed-order sum, scale, fftshift, dB).  launch_welch = begin + one range + end.
int welch_begin(gj_ctx* ctx, size_t nbytes, size_t chunk_samples, int nperseg, double fs, size_t plan_bytes, WelchJob& job) {
    if (!(fs > 0.0)) return fail(ctx, GJ_ERR_INVALID, "fs must be > 0");
    WelchPlan pl;
    if (!welch_plan(ctx, nbytes, chunk_samples, nperseg, fs, pl, plan_bytes))
        return fail(ctx, GJ_ERR_UNSUPPORTED, "nperseg must be a power of two in [16, 4096] and <= chunk_samples, chunk_samples < 2^31 - nperseg");
    if ((unsigned long long)pl.g.nchunks * pl.g.splits > 0x7fffffffull) return fail(ctx, GJ_ERR_UNSUPPORTED, "too many chunks");
    static_assert(sizeof(WelchPlan) <= sizeof(job.plan), "WelchJob::plan too small");
    memcpy(job.plan, &pl, sizeof(pl));
    job.nperseg = nperseg;
    job.rows = pl.rows;
    job.ws_bytes = pl.ws_bytes;
    job.partial = nullptr;
    return GJ_OK;
}

int welch_range(gj_ctx* ctx, const WelchJob& job, const uint8_t* d_iq, size_t c0, size_t c1) {
    WelchPlan pl;
    memcpy(&pl, job.plan, sizeof(pl));
    if (c1 > pl.rows) c1 = pl.rows;
    if (c0 >= c1) return GJ_OK;
    if ((reinterpret_cast<uintptr_t>(d_iq) & 1) != 0) return fail(ctx, GJ_ERR_INVALID, "capture must be 2-byte aligned");
    float* partial = job.partial;
    const unsigned a = (unsigned)c0, b = (unsigned)c1;
    switch (job.nperseg) {
        case 16: welch_launch<16>(ctx, d_iq, pl, partial, a, b); break;
        case 32: welch_launch<32>(ctx, d_iq, pl, partial, a, b); break;
        case 64: welch_launch<64>(ctx, d_iq, pl, partial, a, b); break;
        case 128: welch_launch<128>(ctx, d_iq, pl, partial, a, b); break;
        case 256: welch_launch<256>(ctx, d_iq, pl, partial, a, b); break;
        case 512: welch_launch<512>(ctx, d_iq, pl, partial, a, b); break;
        case 1024: welch_launch<1024>(ctx, d_iq, pl, partial, a, b); break;
        case 2048: welch_launch<2048>(ctx, d_iq, pl, partial, a, b); break;
        default: welch_launch<4096>(ctx, d_iq, pl, partial, a, b); break;
    }
    GJ_LAUNCH_CHECK(ctx);
    return GJ_OK;
}

int welch_end(gj_ctx* ctx, const WelchJob& job, int flags, float* d_psd, float* d_psd_db) {
    WelchPlan pl;
    memcpy(&pl, job.plan, sizeof(pl));
    if (pl.rows == 0) return GJ_OK;
    const int nperseg = job.nperseg;
    float* partial = job.partial;
    const bool aligned = ((reinterpret_cast<uintptr_t>(d_psd) | reinterpret_cast<uintptr_t>(d_psd_db)) & 15) == 0;
    if (aligned)
        hipLaunchKernelGGL(welch_finalize_kernel, dim3((nperseg / 4 + 63) / 64, pl.g.nchunks), dim3(256), 0, ctx->stream,
                           partial, nperseg, pl.g.splits, pl.g.nchunks, (float)pl.scale_full,
                           (float)pl.scale_last, (flags & GJ_WELCH_SHIFT) ? 1 : 0, d_psd, d_psd_db);
    else
        hipLaunchKernelGGL(welch_finalize_scalar_kernel, dim3((nperseg + 63) / 64, pl.g.nchunks), dim3(256), 0,
                           ctx->stream, partial, nperseg, pl.g.splits, pl.g.nchunks,
                           (float)pl.scale_full, (float)pl.scale_last, (flags & GJ_WELCH_SHIFT) ? 1 : 0, d_psd, d_psd_db);
    GJ_LAUNCH_CHECK(ctx);
    return GJ_OK;
}

int launch_welch(gj_ctx* ctx, const uint8_t* d_iq, size_t nbytes, size_t chunk_samples, int nperseg, double fs,
                 int flags, float* d_psd, float* d_psd_db, size_t plan_bytes) {
    WelchJob job;
    int rc = welch_begin(ctx, nbytes, chunk_samples, nperseg, fs, plan_bytes, job);
    if (rc) return rc;
    if ((reinterpret_cast<uintptr_t>(d_iq) & 1) != 0) return fail(ctx, GJ_ERR_INVALID, "capture must be 2-byte aligned");
    if (job.rows == 0) return GJ_OK;
    rc = ensure_workspace(ctx, job.ws_bytes);
    if (rc) return rc;
    job.partial = reinterpret_cast<float*>(ctx->ws);
    rc = welch_range(ctx, job, d_iq, 0, job.rows);
    if (rc) return rc;
    return welch_end(ctx, job, flags, d_psd, d_psd_db);
}

}   // namespace gj
