// Streaming scans over a uint8 I/Q capture resident in HBM (gfx950).
//   K1  chunk power + noise-floor threshold   (GpsJammerApp/app/worker.py:216-264)
//   K3  amplitude statistics                  (skrypty/triangulateRSSI.py:29-31,65-68)
//   K4  interference onset                    (skrypty/triangulateTDOA.py:37-49)
//       byte histogram                        (skrypty/widmo_plot.py:35,85)
// All of them are HBM-bound: 2 bytes per I/Q sample read once with 16-byte-per-lane
// coalesced loads, integer arithmetic on the packed bytes (v_dot4_u32_u8), wave shuffles
// + one LDS hop for the block reduction.  Sums of squares are exact integers:
//   (I-127.5)^2 + (Q-127.5)^2 = ((2I-255)^2 + (2Q-255)^2) / 4.
#include <type_traits>

#include "gj_common.h"

namespace gj {

constexpr int kScanThreads = 256;
constexpr size_t kScanTile = 65536;   // bytes per workgroup step

__device__ __forceinline__ void acc_moments(const uint4& q, unsigned& s2, unsigned& s1) {
    s2 = __builtin_amdgcn_udot4(q.x, q.x, s2, false);
    s2 = __builtin_amdgcn_udot4(q.y, q.y, s2, false);
    s2 = __builtin_amdgcn_udot4(q.z, q.z, s2, false);
    s2 = __builtin_amdgcn_udot4(q.w, q.w, s2, false);
    s1 = __builtin_amdgcn_udot4(q.x, 0x01010101u, s1, false);
    s1 = __builtin_amdgcn_udot4(q.y, 0x01010101u, s1, false);
    s1 = __builtin_amdgcn_udot4(q.z, 0x01010101u, s1, false);
    s1 = __builtin_amdgcn_udot4(q.w, 0x01010101u, s1, false);
}

// sum u^2 and sum u over the bytes [begin, end) of `p`, whole workgroup cooperating;
// result valid in thread 0.
__device__ __forceinline__ void block_byte_moments(const uint8_t* __restrict__ p, size_t begin, size_t end,
                                                   unsigned long long& s2_out, unsigned long long& s1_out) {
    __shared__ unsigned long long red[2][kScanThreads / 64];
    const int tid = threadIdx.x;
    unsigned s2 = 0, s1 = 0;
    size_t a0 = (begin + 15) & ~size_t(15);
    size_t a1 = end & ~size_t(15);
    if (a0 > a1) { a0 = end; a1 = end; }
    // ragged head / tail: at most 15 bytes each
    if (begin + tid < a0) { unsigned u = p[begin + tid]; s2 += u * u; s1 += u; }
    if (a1 + tid < end && a1 >= a0) { unsigned u = p[a1 + tid]; s2 += u * u; s1 += u; }
    const uint4* v = reinterpret_cast<const uint4*>(p + a0);
    const size_t nvec = (a1 - a0) >> 4;
    size_t i = tid;
    for (; i + 3 * kScanThreads < nvec; i += 4 * kScanThreads) {
        uint4 w0 = v[i], w1 = v[i + kScanThreads], w2 = v[i + 2 * kScanThreads], w3 = v[i + 3 * kScanThreads];
        acc_moments(w0, s2, s1); acc_moments(w1, s2, s1); acc_moments(w2, s2, s1); acc_moments(w3, s2, s1);
    }
    for (; i < nvec; i += kScanThreads) {
        uint4 w0 = v[i];
        acc_moments(w0, s2, s1);
    }
    unsigned long long t2 = wave_sum_u64(s2), t1 = wave_sum_u64(s1);
    if ((tid & 63) == 0) { red[0][tid >> 6] = t2; red[1][tid >> 6] = t1; }
    __syncthreads();
    if (tid == 0) {
        s2_out = red[0][0] + red[0][1] + red[0][2] + red[0][3];
        s1_out = red[1][0] + red[1][1] + red[1][2] + red[1][3];
    }
    __syncthreads();
}

__device__ __forceinline__ float power_from_moments(unsigned long long s2, unsigned long long s1, size_t npairs,
                                                    float eps, int o2) {
    // sum (2u-off2)^2 over 2*npairs bytes
    const long long S = 4ll * (long long)s2 - 4ll * o2 * (long long)s1 + (long long)o2 * o2 * (long long)(2 * npairs);
    const float mean = (float)((double)S / (4.0 * (double)npairs));
    return mean + eps;
}

// the same from S = sum over the pairs of (2I-255)^2 + (2Q-255)^2 (what the fused scan accumulates)
__device__ __forceinline__ float power_from_msum(unsigned long long S, size_t npairs, float eps) {
    const float mean = (float)((double)S / (4.0 * (double)npairs));
    return mean + eps;
}

__global__ __launch_bounds__(kScanThreads) void chunk_power_kernel(const uint8_t* __restrict__ iq, size_t nbytes,
                                                                   size_t chunk_bytes, unsigned tiles_per_chunk,
                                                                   float eps, int flags, float* __restrict__ power,
                                                                   unsigned long long* __restrict__ acc, int o2) {
    const size_t c = blockIdx.x / tiles_per_chunk;
    const unsigned t = blockIdx.x % tiles_per_chunk;
    const size_t off = c * chunk_bytes;
    const size_t len = (nbytes - off < chunk_bytes) ? nbytes - off : chunk_bytes;
    const size_t npairs = len >> 1;
    if ((flags & GJ_CP_ODD_CHUNK_ZERO) && ((len & 1) || len == 0)) {
        if (t == 0 && threadIdx.x == 0) power[c] = 0.0f;
        return;
    }
    if (npairs == 0) {
        if (t == 0 && threadIdx.x == 0) power[c] = __builtin_nanf("");
        return;
    }
    const size_t use_end = off + 2 * npairs;
    size_t b = off + (size_t)t * kScanTile;
    size_t e = b + kScanTile;
    if (e > use_end) e = use_end;
    if (b > e) b = e;
    unsigned long long s2 = 0, s1 = 0;
    block_byte_moments(iq, b, e, s2, s1);
    if (threadIdx.x == 0) {
        if (tiles_per_chunk == 1) {
            power[c] = power_from_moments(s2, s1, npairs, eps, o2);
        } else {
            atomicAdd(&acc[2 * c], s2);
            atomicAdd(&acc[2 * c + 1], s1);
        }
    }
}

// msum: acc[2c] holds the sum of (2u-255)^2 itself (fused scan) instead of the byte moments
__global__ void chunk_power_finalize_kernel(const unsigned long long* __restrict__ acc, size_t nchunks, size_t nbytes,
                                            size_t chunk_bytes, float eps, int flags, float* __restrict__ power,
                                            int o2, bool msum = false) {
    const size_t c = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (c >= nchunks) return;
    const size_t off = c * chunk_bytes;
    const size_t len = (nbytes - off < chunk_bytes) ? nbytes - off : chunk_bytes;
    const size_t npairs = len >> 1;
    if ((flags & GJ_CP_ODD_CHUNK_ZERO) && ((len & 1) || len == 0)) return;   // already written
    if (npairs == 0) return;
    power[c] = msum ? power_from_msum(acc[2 * c], npairs, eps) : power_from_moments(acc[2 * c], acc[2 * c + 1], npairs, eps, o2);
}

int launch_chunk_power(gj_ctx* ctx, const uint8_t* d_iq, size_t nbytes, size_t chunk_bytes, float eps, int flags,
                       float* d_power) {
    if (chunk_bytes == 0) return fail(ctx, GJ_ERR_INVALID, "chunk_bytes must be > 0");
    const size_t nchunks = gj_chunk_count(nbytes, chunk_bytes);
    if (nchunks == 0) return GJ_OK;
    const size_t tiles = (chunk_bytes + kScanTile - 1) / kScanTile;
    if (nchunks * tiles > 0x7fffffffull) return fail(ctx, GJ_ERR_UNSUPPORTED, "too many tiles");
    unsigned long long* acc = nullptr;
    if (tiles > 1) {
        int rc = ensure_workspace(ctx, nchunks * 16);
        if (rc) return rc;
        acc = reinterpret_cast<unsigned long long*>(ctx->ws);
        GJ_HIP(ctx, hipMemsetAsync(acc, 0, nchunks * 16, ctx->stream));
    }
    hipLaunchKernelGGL(chunk_power_kernel, dim3((unsigned)(nchunks * tiles)), dim3(kScanThreads), 0, ctx->stream, d_iq,
                       nbytes, chunk_bytes, (unsigned)tiles, eps, flags, d_power, acc, ctx->off2);
    GJ_LAUNCH_CHECK(ctx);
    if (tiles > 1) {
        hipLaunchKernelGGL(chunk_power_finalize_kernel, dim3((unsigned)((nchunks + 255) / 256)), dim3(256), 0,
                           ctx->stream, acc, nchunks, nbytes, chunk_bytes, eps, flags, d_power, ctx->off2);
        GJ_LAUNCH_CHECK(ctx);
    }
    return GJ_OK;
}

// ---------------------------------------------------------------------------------------
// noise floor: numpy.percentile(power, pct) with the float32 'linear' rule, by radix select
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ unsigned float_key(float f) {
    unsigned b = __float_as_uint(f);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__device__ __forceinline__ float key_float(unsigned k) {
    return __uint_as_float((k & 0x80000000u) ? (k & 0x7fffffffu) : ~k);
}

// Radix select over monotone uint keys of the power map.  One 256-thread workgroup; every
// thread keeps its 64 keys in VGPRs (16 384 chunks = a 1-GiB capture are read from HBM once),
// longer maps re-read global memory (L2 hits).  256-bin histograms by LDS atomics, digit search
// by a parallel prefix scan.  The footprint (1 KiB of LDS, < 112 VGPRs, one wave per SIMD) is
// deliberate: the kernel must fit NEXT TO two resident Welch workgroups of a CU, because the
// per-stream pipeline runs it on a second stream while K2 owns the chip -- the earlier
// 128-KiB-LDS version waited ~0.75 ms for K2's grid to drain.
constexpr int kThrThreads = 256;
constexpr int kThrPerThread = 64;
constexpr int kThrCache = kThrThreads * kThrPerThread;   // keys held in registers

struct ThrShared {
    unsigned hist[256];
    unsigned wave_tot[4];
    unsigned digit, rest;
    unsigned nan_flag;
    unsigned count_le;
    unsigned next_key;
    unsigned long long above;
    float thr;
};

// f(key, index) over every element this thread owns (index = tid + 256 j, j < mine)
template <typename F>
__device__ __forceinline__ void thr_for_each(const unsigned (&reg)[kThrPerThread], const float* __restrict__ power,
                                             bool cached, size_t n, F&& f) {
    const unsigned tid = threadIdx.x;
    if (cached) {
        const unsigned n32 = (unsigned)n;
        const unsigned mine = (n32 > tid) ? (n32 - tid + kThrThreads - 1) / kThrThreads : 0u;
#pragma unroll
        for (unsigned j = 0; j < (unsigned)kThrPerThread; ++j)
            if (j < mine) f(reg[j], tid + kThrThreads * j);
    } else {
        for (size_t i = tid; i < n; i += kThrThreads) f(float_key(power[i]), i);
    }
}

// key of rank `rank` (0-based, ascending, NaNs last) -- whole block must call
__device__ unsigned block_select(const unsigned (&reg)[kThrPerThread], const float* __restrict__ power, bool cached,
                                 size_t n, size_t rank, ThrShared& sh) {
    unsigned prefix = 0, mask = 0;
    unsigned want = (unsigned)rank;
    const int tid = threadIdx.x;
    for (int shift = 24; shift >= 0; shift -= 8) {
        sh.hist[tid] = 0;
        __syncthreads();
        // equal digits are the rule (neighbouring powers share exponent and leading mantissa
        // bits): count runs in registers and flush one atomic per run
        unsigned run_digit = 0xffffffffu, run_len = 0;
        thr_for_each(reg, power, cached, n, [&](unsigned k, size_t) {
            if ((k & mask) == prefix) {
                const unsigned d = (k >> shift) & 255u;
                if (d != run_digit) {
                    if (run_len) atomicAdd(&sh.hist[run_digit], run_len);
                    run_digit = d;
                    run_len = 0;
                }
                ++run_len;
            }
        });
        if (run_len) atomicAdd(&sh.hist[run_digit], run_len);
        __syncthreads();
        // digit d with excl(d) <= want < excl(d) + hist[d]
        const unsigned h = sh.hist[tid];
        unsigned inc = h;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const unsigned o = __shfl_up(inc, off, 64);
            if ((tid & 63) >= off) inc += o;
        }
        if ((tid & 63) == 63) sh.wave_tot[tid >> 6] = inc;
        __syncthreads();
        unsigned base = 0;
        for (int w = 0; w < (tid >> 6); ++w) base += sh.wave_tot[w];
        const unsigned excl = base + inc - h;
        if (h && excl <= want && want < excl + h) { sh.digit = tid; sh.rest = want - excl; }
        __syncthreads();
        prefix |= sh.digit << shift;
        mask |= 255u << shift;
        want = sh.rest;
        __syncthreads();
    }
    return prefix;
}

// whole 256-thread workgroup; `sh` is the workgroup's scratch
__device__ __forceinline__ void power_threshold_body(const float* __restrict__ power, size_t n, float pct, float ratio,
                                                     float* __restrict__ stats, uint8_t* __restrict__ mask, ThrShared& sh) {
    const int tid = threadIdx.x;
    const bool cached = n <= (size_t)kThrCache;
    if (tid == 0) { sh.nan_flag = 0; sh.above = 0; sh.count_le = 0; sh.next_key = 0xffffffffu; }
    __syncthreads();
    unsigned reg[kThrPerThread];
    unsigned has_nan = 0;
    if (cached) {
#pragma unroll
        for (int j = 0; j < kThrPerThread; ++j) {
            const size_t i = (size_t)tid + (size_t)kThrThreads * j;
            const float p = (i < n) ? power[i] : 0.0f;
            has_nan |= (p != p);
            reg[j] = float_key(p);
        }
    } else {
#pragma unroll
        for (int j = 0; j < kThrPerThread; ++j) reg[j] = 0;
        for (size_t i = tid; i < n; i += kThrThreads) {
            const float p = power[i];
            has_nan |= (p != p);
        }
    }
    if (has_nan) atomicOr(&sh.nan_flag, 1u);
    // numpy 2.x: q = float32(pct)/float32(100); virtual index = float32(n-1) * q  (all float32)
    const float q = pct / 100.0f;
    const float vidx = (float)(n - 1) * q;
    size_t lo = (size_t)floorf(vidx);
    if (lo > n - 1) lo = n - 1;
    const size_t hi = (lo + 1 < n) ? lo + 1 : n - 1;
    const float g = vidx - (float)lo;
    const unsigned ka = block_select(reg, power, cached, n, lo, sh);
    // rank lo+1: ka again when it occurs often enough, else the smallest key above it
    unsigned kb = ka;
    if (hi != lo) {
        unsigned cnt = 0, nxt = 0xffffffffu;
        thr_for_each(reg, power, cached, n, [&](unsigned k, size_t) {
            cnt += (k <= ka);
            if (k > ka && k < nxt) nxt = k;
        });
        cnt = wave_sum_u32(cnt);
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            const unsigned o = __shfl_xor(nxt, off, 64);
            nxt = o < nxt ? o : nxt;
        }
        if ((tid & 63) == 0) { atomicAdd(&sh.count_le, cnt); atomicMin(&sh.next_key, nxt); }
        __syncthreads();
        kb = (sh.count_le >= lo + 2) ? ka : sh.next_key;
    }
    if (tid == 0) {
        const float a = key_float(ka), b = key_float(kb);
        const float d = b - a;
        float base = (g >= 0.5f) ? (b - d * (1.0f - g)) : (a + d * g);
        if (sh.nan_flag) base = __builtin_nanf("");
        if (base <= 0.0f) base = 1.0f;
        const float thr = base * ratio;
        stats[0] = base;
        stats[1] = thr;
        sh.thr = thr;
    }
    __syncthreads();
    const float thr = sh.thr;
    unsigned cnt = 0;
    thr_for_each(reg, power, cached, n, [&](unsigned k, size_t i) {
        const bool hot = key_float(k) > thr;
        cnt += hot;
        if (mask) mask[i] = hot;
    });
    cnt = wave_sum_u32(cnt);
    if ((tid & 63) == 0) atomicAdd(&sh.above, (unsigned long long)cnt);
    __syncthreads();
    if (tid == 0) stats[2] = (float)sh.above;
}

__global__ __launch_bounds__(kThrThreads) __attribute__((amdgpu_num_vgpr(104))) void power_threshold_kernel(const float* __restrict__ power, size_t n, float pct,
                                                                      float ratio, float* __restrict__ stats,
                                                                      uint8_t* __restrict__ mask) {
    __shared__ ThrShared sh;
    power_threshold_body(power, n, pct, ratio, stats, mask, sh);
}

int launch_power_threshold(gj_ctx* ctx, const float* d_power, size_t n, float pct, float rise_db, float* d_stats,
                           uint8_t* d_mask) {
    if (n == 0) return fail(ctx, GJ_ERR_INVALID, "empty power map");
    const float ratio = (float)pow(10.0, (double)rise_db / 10.0);
    hipLaunchKernelGGL(power_threshold_kernel, dim3(1), dim3(kThrThreads), 0, ctx->stream, d_power, n, pct, ratio, d_stats,
                       d_mask);
    GJ_LAUNCH_CHECK(ctx);
    return GJ_OK;
}

// ---------------------------------------------------------------------------------------
// K3 amplitude statistics
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ float amp_of(unsigned i8, unsigned q8, Unpack up) {
    const int vi = 2 * (int)i8 - up.off2, vq = 2 * (int)q8 - up.off2;
    return __fsqrt_rn((float)(vi * vi + vq * vq)) * up.half_scale;
}
// packed (-off2, -off2) for v_pk_mad_i16
__device__ __forceinline__ unsigned pk_minus_off2(int o2) { return ((unsigned)(-o2) & 0xffffu) * 0x10001u; }

// amp_of for sample `HI` (0: bytes 0-1, 1: bytes 2-3) of the dword w, in three integer
// instructions instead of seven: v_perm_b32 spreads (I, Q) into the two 16-bit halves,
// v_pk_mad_i16 forms (2I-255, 2Q-255), v_dot2_i32_i16 squares and adds.  k2 = 0x00020002,
// km255 = 0xFF01FF01 (packed -255) are loop-invariant registers.
template <int HI>
__device__ __forceinline__ float amp_of_half(unsigned w, unsigned k2, unsigned km255, float hs) {
    const unsigned spread = __builtin_amdgcn_perm(w, w, HI ? 0x0C030C02u : 0x0C010C00u);
    unsigned h;
    int m;
    asm("v_pk_mad_i16 %0, %1, %2, %3" : "=v"(h) : "v"(spread), "v"(k2), "v"(km255));
    asm("v_dot2_i32_i16 %0, %1, %1, 0" : "=v"(m) : "v"(h));
    return __fsqrt_rn((float)m) * hs;
}

__device__ __forceinline__ unsigned m_of(unsigned i8, unsigned q8, int o2 = 255) {
    const int vi = 2 * (int)i8 - o2, vq = 2 * (int)q8 - o2;
    return (unsigned)(vi * vi + vq * vq);   // = 4 |z|^2
}

// (2I-255)^2 + (2Q-255)^2 of sample HI (0: bytes 0-1, 1: bytes 2-3) of the dword w: v_perm_b32 spreads (I, Q)
// into the two 16-bit halves, v_pk_mad_i16 forms (2I-255, 2Q-255), v_dot2_i32_i16 squares and adds
template <int HI>
__device__ __forceinline__ int msq_of_half(unsigned w, unsigned k2, unsigned km255) {
    const unsigned spread = __builtin_amdgcn_perm(w, w, HI ? 0x0C030C02u : 0x0C010C00u);
    unsigned h;
    int m;
    asm("v_pk_mad_i16 %0, %1, %2, %3" : "=v"(h) : "v"(spread), "v"(k2), "v"(km255));
    asm("v_dot2_i32_i16 %0, %1, %1, 0" : "=v"(m) : "v"(h));
    return m;
}

constexpr size_t kAmpTileSamples = kScanTile / 2;

struct AmpTile {
    double sum;
    long long first;   // absolute sample index or LLONG_MAX
};

__device__ double block_sum_f64(double v, double* sh /* [blockDim/64] */) {
    v = wave_sum_f64(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    double t = 0.0;
    for (unsigned k = 0; k < blockDim.x / 64; ++k) t += sh[k];
    return t;
}

// ---- amplitude totals, in three steps whose results do not depend on how a capture is cut into parts ----
//   first : smallest first-hit index over the tiles (exact)
//   tail  : sum of the amplitudes from `first` to the end of ITS tile (block reduction over a fixed thread stride;
//           the tile's own sum when the hit is the tile's first sample)
//   total : sum of the tile sums BEHIND that tile (fixed thread stride over the global tile index) + tail
// A whole capture runs the three in the amplitude role of scan_tail_kernel; a capture split over GPUs runs first + tail
// where the bytes are (the same role) and total where the tile sums have been gathered (amp_combine_kernel): same code,
// same order, same bits.
__device__ long long amp_block_first(const AmpTile* __restrict__ tiles, size_t ntiles, long long* first_s) {
    if (threadIdx.x == 0) *first_s = 0x7fffffffffffffffll;
    __syncthreads();
    long long f = 0x7fffffffffffffffll;
    // sixteen independent loads in flight per thread: this single workgroup is latency-bound (a 1-GiB capture has
    // 16 384 tiles = 64 per thread: four round trips instead of eight), and beside K2 every round trip is slow
    for (size_t t = threadIdx.x; t < ntiles; t += 16 * (size_t)blockDim.x) {
        long long v[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const size_t tt = t + (size_t)k * blockDim.x;
            v[k] = tt < ntiles ? tiles[tt].first : 0x7fffffffffffffffll;
        }
#pragma unroll
        for (int k = 0; k < 16; ++k) f = v[k] < f ? v[k] : f;
    }
    if (f != 0x7fffffffffffffffll) atomicMin(first_s, f);
    __syncthreads();
    return *first_s;
}

// `first` and the tile array are in the coordinates of `iq` (sample 0 = iq[0], tile 0 starts there)
__device__ double amp_block_tail(const uint8_t* __restrict__ iq, size_t nsamples, const AmpTile* __restrict__ tiles,
                                 long long first, Unpack up, double* sh) {
    const size_t t0 = (size_t)first / kAmpTileSamples;
    if ((size_t)first == t0 * kAmpTileSamples) return tiles[t0].sum;   // hit on the tile's first sample: its sum is the tail
    const size_t e0 = ((t0 + 1) * kAmpTileSamples < nsamples) ? (t0 + 1) * kAmpTileSamples : nsamples;
    double acc = 0.0;
    for (size_t s = (size_t)first + threadIdx.x; s < e0; s += blockDim.x) acc += (double)amp_of(iq[2 * s], iq[2 * s + 1], up);
    return block_sum_f64(acc, sh);
}

__device__ double amp_block_total(const AmpTile* __restrict__ tiles, size_t ntiles, size_t t0, double* sh) {
    double acc = 0.0;
    // The ORDER of the additions is fixed (it is what makes a capture's sum the same bits whether it is summed here, on
    // the combining rank of a split run or after an ingest): per thread, groups of eight tiles a block-stride apart,
    // each group added as ((v0+v1)+(v2+v3))+((v4+v5)+(v6+v7)), the groups in ascending order.  Round 5 only moves the
    // LOADS: two groups' worth are issued before the first addition (16 in flight instead of 8).
    for (size_t t = t0 + 1 + threadIdx.x; t < ntiles; t += 16 * (size_t)blockDim.x) {
        double v[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const size_t tt = t + (size_t)k * blockDim.x;
            v[k] = tt < ntiles ? tiles[tt].sum : 0.0;
        }
#pragma unroll
        for (int g = 0; g < 2; ++g) {
            // a group that starts beyond the end adds +0.0 to a non-negative sum: the value is unchanged, as if the
            // loop had ended there
            if (t + (size_t)(8 * g) * blockDim.x < ntiles)
                acc += ((v[8 * g] + v[8 * g + 1]) + (v[8 * g + 2] + v[8 * g + 3])) + ((v[8 * g + 4] + v[8 * g + 5]) + (v[8 * g + 6] + v[8 * g + 7]));
        }
    }
    return block_sum_f64(acc, sh);
}

// (fused scan only) the last chunk when no tile wrote it or the odd-chunk rule applies: true when the LAST chunk's
// power is decided by the rule, not by the chunk's sums.  The word is written by a one-word memset in front of the tail
// launch (scan_end): until round 6 the tail's amplitude workgroup wrote it while the threshold workgroup of the SAME
// launch read the map -- no order between the two.  Rare by construction (odd length, or length = 1 mod chunk).
static inline bool power_edge_value(size_t nchunks, size_t nbytes, size_t chunk_bytes, int flags, float* val) {
    if (!nchunks) return false;
    const size_t len = nbytes - (nchunks - 1) * chunk_bytes;
    if ((flags & GJ_CP_ODD_CHUNK_ZERO) && (len & 1)) { *val = 0.0f; return true; }
    if ((len >> 1) == 0) { *val = __builtin_nanf(""); return true; }
    return false;
}
// The combining rank: tile sums of the WHOLE capture (gathered, in tile order) + the parts' first hits and tails.
__device__ __forceinline__ void amp_combine_body(const AmpTile* __restrict__ tiles, size_t ntiles,
                                                 const gj_amp_part* __restrict__ parts, int n_parts, size_t nsamples,
                                                 gj_amp_stats* __restrict__ out, double* sh) {
    long long first = 0x7fffffffffffffffll;
    double tail = 0.0;
    for (int p = 0; p < n_parts; ++p)        // every thread: n_parts is small
        if (parts[p].first_index >= 0 && parts[p].first_index < first) { first = parts[p].first_index; tail = parts[p].tail; }
    if (first == 0x7fffffffffffffffll) {
        if (threadIdx.x == 0) {
            out->first_index = -1; out->count = 0; out->sum = 0.0; out->mean = 0.f; out->reserved = 0.f;
        }
        return;
    }
    const double behind = amp_block_total(tiles, ntiles, (size_t)first / kAmpTileSamples, sh);
    if (threadIdx.x == 0) {
        const double total = behind + tail;
        const unsigned long long cnt = nsamples - (size_t)first;
        out->first_index = first;
        out->count = cnt;
        out->sum = total;
        out->mean = (float)(total / (double)cnt);
        out->reserved = 0.f;
    }
}

__global__ __launch_bounds__(256) void amp_combine_kernel(const AmpTile* __restrict__ tiles, size_t ntiles,
                                                           const gj_amp_part* __restrict__ parts, int n_parts,
                                                           size_t nsamples, gj_amp_stats* __restrict__ out) {
    __shared__ double sh[16];
    amp_combine_body(tiles, ntiles, parts, n_parts, nsamples, out, sh);
}

// Rank 0 of a split run, EVERY capture in one launch (gj_split_combine_dev): workgroup a < n_captures runs the
// noise-floor threshold of capture a, workgroup n_captures + a its amplitude totals and its onset -- the bodies of
// power_threshold_kernel, amp_combine_kernel and onset_combine_kernel as they are (same code, same order, same bits),
// one workgroup each, side by side instead of three launches per antenna one after the other.
__global__ __launch_bounds__(kThrThreads) __attribute__((amdgpu_num_vgpr(104))) void combine_stats_kernel(
    const gj_combine_capture* __restrict__ caps, int n_captures, float pct, float ratio) {
    __shared__ ThrShared sh_thr;
    __shared__ double sh_amp[16];
    const int b = blockIdx.x;
    if (b < n_captures) {
        const gj_combine_capture c = caps[b];
        power_threshold_body(c.d_power, (size_t)c.n_chunks, pct, ratio, c.d_stats, nullptr, sh_thr);
    } else {
        const gj_combine_capture c = caps[b - n_captures];
        amp_combine_body(static_cast<const AmpTile*>(c.d_tiles), (size_t)c.n_tiles, c.d_amp_parts, c.n_parts,
                         (size_t)(c.total_bytes / 2), c.d_amp, sh_amp);
        if (threadIdx.x == 0) onset_combine(c.d_onset_parts, c.n_parts, c.d_onset);
    }
}

int launch_combine_stats(gj_ctx* ctx, const gj_combine_capture* d_caps, int n_captures, float pct, float rise_db) {
    const float ratio = (float)pow(10.0, (double)rise_db / 10.0);
    hipLaunchKernelGGL(combine_stats_kernel, dim3(2u * (unsigned)n_captures), dim3(kThrThreads), 0, ctx->stream, d_caps,
                       n_captures, pct, ratio);
    GJ_LAUNCH_CHECK(ctx);
    return GJ_OK;
}

// ---------------------------------------------------------------------------------------
// K4 onset: moving average of |z|^2 against factor x noise, exact integer window sums
// ---------------------------------------------------------------------------------------
constexpr int kOnsetOut = 2048;    // moving-average positions per tile of the exact scan
constexpr int kOnsetMaxWin = 8192;

// What the scan launch hands the tail besides the block sums: the noise sum of a part that does not hold the capture's
// noise span (summed from the copy it brought along), and the threshold the first onset workgroup derived for the record.
struct OnsetScratch {
    unsigned long long noise_S;     // sum of (2I-255)^2 + (2Q-255)^2 = 4 |z|^2 over the first noise_samples samples
    float noise;
    float thr;
};

// Rounding band of the decision (gj_onset.guard_index): the window sums here are exact, the reference's carry ~3e-7
// of float32 rounding (|z|^2 per sample, the pairwise noise mean, the factor).  A moving average below
// threshold * (1 - kOnsetGuard) is below the reference's threshold whatever its rounding did.
constexpr double kOnsetGuard = 1e-6;

__device__ __forceinline__ int onset_pad(int k) { return k + (k >> 5); }   // spreads stride-`per` accesses over banks

// ---------------------------------------------------------------------------------------
// Fused stream scan: ONE pass over the capture feeds K1 (chunk power), K3 (amplitude
// statistics) and K4 (noise-span moments + 512-sample block sums for the screening), instead of
// three passes.
// One workgroup per 64 KiB tile; per 16-byte load (8 samples) a lane computes the byte moments
// (v_dot4_u32_u8) that give both the chunk sums and the block sum of 4|z|^2, and the eight
// amplitudes (v_dot2 on the packed (2I-255, 2Q-255) pair, v_sqrt_f32).
// ---------------------------------------------------------------------------------------
typedef short gj_short2 __attribute__((ext_vector_type(2)));

// Sum of v over the 64 lanes, valid in LANE 63 only: four in-row DPP steps, then row_bcast:15 into rows 1 and 3
// and row_bcast:31 into rows 2 and 3 -- six VALU adds, no readlane, no LDS (group_sum_dpp<64> needs eleven
// instructions to make the total wave-uniform, which the block-sum store does not need).
__device__ __forceinline__ int wave_sum_lane63(int v) {
    v += __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xf, 0xf, false);    // quad_perm [1,0,3,2]
    v += __builtin_amdgcn_update_dpp(0, v, 0x4E, 0xf, 0xf, false);    // quad_perm [2,3,0,1]
    v += __builtin_amdgcn_update_dpp(0, v, 0x141, 0xf, 0xf, false);   // row_half_mirror
    v += __builtin_amdgcn_update_dpp(0, v, 0x140, 0xf, 0xf, false);   // row_mirror: every lane holds its row's sum
    v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, false);   // row_bcast:15, rows 1 and 3
    v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, false);   // row_bcast:31, rows 2 and 3
    return v;
}


template <bool TRACK_FIRST>
__global__ __launch_bounds__(kScanThreads) void stream_scan_kernel(const uint8_t* __restrict__ iq, size_t nsamples,
                                                                   size_t nbytes, size_t chunk_bytes,
                                                                   unsigned tiles_per_chunk, float eps, int flags,
                                                                   float* __restrict__ power,
                                                                   unsigned long long* __restrict__ acc, float thr,
                                                                   AmpTile* __restrict__ tiles,
                                                                   unsigned* __restrict__ cblk, Unpack up,
                                                                   unsigned skip_tiles, unsigned tile_base,
                                                                   unsigned noise_block, const uint8_t* __restrict__ noise_src,
                                                                   int noise_samples, OnsetScratch* __restrict__ sc) {
    // skip_tiles (a part of a split capture): the first tiles of the buffer are the HALO in front of the part's own
    // range -- they feed K4's block sums only; chunk powers and amplitude tiles start at the own range (`nbytes` =
    // its length, `power` / `tiles` / `acc` its arrays).
    // noise_block (a part that does not hold the capture's first noise_samples samples): one extra workgroup sums K4's
    // noise span from the copy the part brought along (`noise_src`) -- exact integers -- for the tail kernel.  A buffer
    // that starts with the span needs nothing here: the tail adds up this kernel's own 512-sample block sums.
    if (blockIdx.x == noise_block) {
        unsigned long long s2 = 0, s1 = 0;
        block_byte_moments(noise_src, 0, (size_t)2 * noise_samples, s2, s1);
        if (threadIdx.x == 0)
            sc->noise_S = (unsigned long long)(4ll * (long long)s2 - 4ll * up.off2 * (long long)s1 + (long long)up.off2 * up.off2 * (2ll * noise_samples));
        return;
    }
    __shared__ unsigned long long red_m[2][kScanThreads / 64];
    __shared__ double red_s[kScanThreads / 64];
    __shared__ long long red_f[kScanThreads / 64];
    const int tid = threadIdx.x;
    const size_t t = (size_t)blockIdx.x + tile_base;   // tile_base: this launch covers the tiles from there on (gj_ingest_*)
    const size_t b0 = t * kScanTile;                               // first byte of the tile
    const size_t use_end = 2 * nsamples;                           // a trailing odd byte is never used
    const size_t b1 = (b0 + kScanTile < use_end) ? b0 + kScanTile : use_end;
    const size_t nvec = (b1 - b0) >> 4;                            // full 16-byte vectors
    const uint4* v = reinterpret_cast<const uint4*>(iq + b0);
    // Everything comes from the per-sample m = (2I-255)^2 + (2Q-255)^2 (exact integers, three instructions per
    // sample): amplitudes are sqrt(m) / 255, chunk power is sum(m) / (4 n), the K4 sums are sums of m.
    unsigned S = 0;                                                // per lane: <= 16 vectors x 8 x 130050 < 2^32
    double sum = 0.0;                                              // sum of sqrt(m); scaled by 1/255 once per tile
    long long first = 0x7fffffffffffffffll;
    unsigned k2 = 0x00020002u, km255 = pk_minus_off2(up.off2);
    asm volatile("" : "+v"(k2), "+v"(km255));                      // keep both in VGPRs (one constant-bus slot per op)
    // `live` is false only for the lanes past the end of a partial last tile (see below): they take
    // part in the wave-wide reductions with zero contributions
    auto body = [&](const uint4& q, unsigned i, const bool live) {   // i: vector index inside the tile (< 4096)
        const unsigned ws[4] = {q.x, q.y, q.z, q.w};
        int m[8];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            m[2 * k] = msq_of_half<0>(ws[k], k2, km255);
            m[2 * k + 1] = msq_of_half<1>(ws[k], k2, km255);
        }
        float part = 0.f;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const float r = __fsqrt_rn((float)m[k]);
            part += r;
            if constexpr (TRACK_FIRST) {
                const long long idx = (long long)((b0 >> 1) + (size_t)i * 8 + k);
                if (live && r * up.half_scale > thr && idx < first) first = idx;   // same expression as K3 alone
            }
        }
        const unsigned m8 = live ? (unsigned)(((m[0] + m[1]) + (m[2] + m[3])) + ((m[4] + m[5]) + (m[6] + m[7]))) : 0u;
        S += m8;
        // block sum of 4|z|^2 over the wave = one 512-sample block
        const int c512 = wave_sum_lane63((int)m8);
        if ((tid & 63) == 63) cblk[(b0 >> 10) + (i >> 6)] = (unsigned)c512;
        sum += (double)(live ? part : 0.f);
    };
    if (nvec == kScanTile / 16) {
        // full tile: 16 vectors per lane, four 16-byte loads in flight before the arithmetic starts
        for (unsigned i = tid; i < (unsigned)(kScanTile / 16); i += 4 * kScanThreads) {
            const uint4 q0 = v[i], q1 = v[i + kScanThreads], q2 = v[i + 2 * kScanThreads], q3 = v[i + 3 * kScanThreads];
            body(q0, i, true);
            body(q1, i + kScanThreads, true);
            body(q2, i + 2 * kScanThreads, true);
            body(q3, i + 3 * kScanThreads, true);
        }
    } else {
        // partial last tile: the trip count is WAVE-uniform (the 512-sample block sums are reduced
        // and stored per wave inside body(), so a wave must never split here); lanes past the end
        // run on a zero vector and are masked out of every sum
        const unsigned lane = tid & 63;
        for (unsigned ib = (unsigned)tid - lane; ib < (unsigned)nvec; ib += kScanThreads) {
            const unsigned i = ib + lane;
            const bool live = i < (unsigned)nvec;
            const uint4 q = live ? v[i] : uint4{0u, 0u, 0u, 0u};
            body(q, i, live);
        }
    }
    // ragged end of the stream (< 8 samples): one lane, scalar
    if (tid == 0 && (b0 + (nvec << 4)) < b1) {
        unsigned c = 0;
        for (size_t n = (b0 >> 1) + nvec * 8; 2 * n < b1; ++n) {
            const unsigned ui = iq[2 * n], uq = iq[2 * n + 1];
            const unsigned mm = m_of(ui, uq, up.off2);
            S += mm;
            c += mm;
            const float r = __fsqrt_rn((float)mm);
            sum += (double)r;
            if (TRACK_FIRST && r * up.half_scale > thr && (long long)n < first) first = (long long)n;
        }
        (void)c;
    }
    // a stream that ends inside a 512-sample block: the waves above wrote the sum of its full
    // vectors only (inactive lanes add 0); one wave recomputes the whole block after the
    // workgroup's stores have landed
    if (b1 == use_end && (b1 & 1023) != 0) {
        __syncthreads();
        if (tid < 64) {
            const size_t jb = (b1 - 1) >> 10;
            int c = 0;
            for (size_t n = jb * 512 + tid; n < nsamples; n += 64) c += (int)m_of(iq[2 * n], iq[2 * n + 1], up.off2);
            c = group_sum_dpp<64>(c);
            if (tid == 0) cblk[jb] = (unsigned)c;
        }
    }
    if constexpr (!TRACK_FIRST) {
        if (tid == 0 && b1 > b0) first = (long long)(b0 >> 1);   // threshold below the smallest amplitude
    }
    const unsigned long long mS = wave_sum_u64(S);
    sum = wave_sum_f64(sum);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const long long o = __shfl_xor(first, off, 64);
        first = o < first ? o : first;
    }
    if ((tid & 63) == 0) { red_m[0][tid >> 6] = mS; red_s[tid >> 6] = sum; red_f[tid >> 6] = first; }
    __syncthreads();
    if (tid == 0 && t >= skip_tiles) {
        const unsigned long long tS = red_m[0][0] + red_m[0][1] + red_m[0][2] + red_m[0][3];
        AmpTile at;
        at.sum = ((red_s[0] + red_s[1]) + (red_s[2] + red_s[3])) * (double)up.half_scale;
        long long f = red_f[0];
        for (int k = 1; k < 4; ++k) f = red_f[k] < f ? red_f[k] : f;
        // first hit in the coordinates of the own range
        at.first = (f == 0x7fffffffffffffffll) ? f : f - (long long)((size_t)skip_tiles * (kScanTile / 2));
        const size_t to = t - skip_tiles;
        tiles[to] = at;
        const size_t c = to / tiles_per_chunk;
        const size_t off = c * chunk_bytes;
        const size_t len = (nbytes - off < chunk_bytes) ? nbytes - off : chunk_bytes;
        const bool zero_rule = (flags & GJ_CP_ODD_CHUNK_ZERO) && ((len & 1) || len == 0);
        if (tiles_per_chunk == 1) {
            power[c] = zero_rule ? 0.0f : ((len >> 1) ? power_from_msum(tS, len >> 1, eps) : __builtin_nanf(""));
        } else {
            atomicAdd(&acc[2 * c], tS);
        }
    }
}

// ---------------------------------------------------------------------------------------
// The scan's tail in ONE launch (VERDICT r04 "next" 3).  Behind stream_scan_kernel the chain used to be five dependent
// launches of one to a few workgroups each -- amplitude totals, screening, exact onset scan, onset record, noise-floor
// threshold -- and a sixth for the TDOA slot; beside K2 each of them waited 5-50 us for its turn, and at the sizes the
// reference runs (10-s captures) that chain, not K2, was the step.  Here they are ROLES of one grid:
//   workgroup 0                 noise-floor threshold of the power map   (power_threshold_body, as it is)
//   workgroup 1                 amplitude totals (first hit, tail of its tile, tiles behind)
//   workgroups 2 .. 2 + nct-1   K4 over kTailBlocks 512-sample blocks each: screening of its blocks against the guard
//                               band, and -- instead of handing a candidate to a second kernel -- the exact scan of
//                               every stretch of ITS range that fails the proof, in order, up to its first crossing.
//                               Ranges are disjoint and every position is either proven quiet or looked at exactly, so
//                               the first crossing of the capture is the smallest of the workgroups' own: no
//                               workgroup waits for another.  Each writes one record; the LAST to arrive (agent-scope
//                               release / acquire around an arrival counter: gj_common.h) reduces the records in
//                               workgroup order, writes the gj_onset record and cuts the TDOA slot at that onset.
// Results: chunk powers, threshold, amplitude statistics, onset index, guard index, noise, margin_hit and the slot are
// the same bits as the separate launches gave.  margin_before -- a bound, not a measurement (gpsjam.h) -- is now a
// function of the capture alone: the records are combined in workgroup order up to the one that holds the crossing,
// where the separate kernels folded in whatever workgroups AHEAD of the crossing had published by then.
// No memset in front: the noise sum is no longer accumulated by atomics in the scan (it is the sum of the scan's own
// 512-sample block sums over the span + the span's ragged end), every word of the scratch is written before it is read,
// and the arrival counter is left at zero by the last arriver.
// ---------------------------------------------------------------------------------------
constexpr size_t kTailSlotBytes = 256u << 10;           // largest slice the tail's last workgroup cuts itself
constexpr size_t kTailPrioBytes = 256u << 20;           // captures up to this size: the tail runs at raised wave priority
constexpr int kTailBlocks = 2048;                       // 512-sample blocks screened per onset workgroup (1 Mi samples)
constexpr int kTailHalo = (kOnsetMaxWin + 510) / 512 + 1;
constexpr int kTailPre = kTailBlocks + kTailHalo + (kTailBlocks + kTailHalo) / 32 + 8;

struct TailRec {
    unsigned long long first;   // first position of the workgroup's range whose moving average crosses, ~0: none
    unsigned long long guard;   // first position inside (or above) the rounding band, ~0: none
    unsigned below;             // largest window sum in front of `first` (exact where looked at, the screening bound elsewhere)
    unsigned pad;
};

struct TailArgs {
    const uint8_t* iq;               // the buffer K4 runs over (a part: halo + own range)
    unsigned long long nsamples;     // of that buffer
    long long sample0;               // capture index of iq[0] (0 for a whole capture)
    unsigned long long total_samples;   // of the whole capture (slot validity)
    unsigned long long slot_nsamples;   // samples of the buffer a slice may be cut from (>= nsamples: a part's tail)
    int has_thr, is_part, valid, has_slot, hi_prio;
    // threshold role
    float* power;
    unsigned long long nchunks;
    float pct, ratio;
    float* stats;
    uint8_t* mask;
    // amplitude role (the OWN range of a part)
    const uint8_t* own_iq;
    unsigned long long own_samples, own_tiles;
    const AmpTile* tiles;
    gj_amp_stats* amp;
    gj_amp_part* amp_part;
    long long own_sample0;
    // onset role
    int window, noise_samples, noise_in_scratch;
    float factor;
    const unsigned* cblk;
    OnsetScratch* sc;
    TailRec* rec;
    unsigned* arrive;
    unsigned nct;
    gj_onset* onset;
    // slot
    uint8_t* slot;
    unsigned long long slice_samples;
    Unpack up;
};

struct TailShared {
    unsigned long long r64[2][kScanThreads / 64];
    unsigned r32[kScanThreads / 64];
    unsigned long long b64[2];
    unsigned b32;
    int last;
    long long start;
};

__device__ __forceinline__ unsigned long long block_min_u64(unsigned long long v, TailShared& sh, int slot) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const unsigned long long o = __shfl_xor(v, off, 64);
        v = o < v ? o : v;
    }
    if ((threadIdx.x & 63) == 0) sh.r64[slot][threadIdx.x >> 6] = v;
    __syncthreads();
    unsigned long long r = sh.r64[slot][0];
#pragma unroll
    for (int k = 1; k < kScanThreads / 64; ++k) r = sh.r64[slot][k] < r ? sh.r64[slot][k] : r;
    __syncthreads();
    return r;
}
__device__ __forceinline__ unsigned block_max_u32(unsigned v, TailShared& sh) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const unsigned o = __shfl_xor(v, off, 64);
        v = o > v ? o : v;
    }
    if ((threadIdx.x & 63) == 0) sh.r32[threadIdx.x >> 6] = v;
    __syncthreads();
    unsigned r = sh.r32[0];
#pragma unroll
    for (int k = 1; k < kScanThreads / 64; ++k) r = sh.r32[k] > r ? sh.r32[k] : r;
    __syncthreads();
    return r;
}
__device__ __forceinline__ unsigned long long block_sum_u64(unsigned long long v, TailShared& sh) {
    v = wave_sum_u64(v);
    if ((threadIdx.x & 63) == 0) sh.r64[0][threadIdx.x >> 6] = v;
    __syncthreads();
    unsigned long long r = 0;
#pragma unroll
    for (int k = 0; k < kScanThreads / 64; ++k) r += sh.r64[0][k];
    __syncthreads();
    return r;
}

// min of a, min of b and max of c over the workgroup in ONE exchange (two barriers instead of six)
__device__ __forceinline__ void block_min2_max(unsigned long long& a, unsigned long long& b, unsigned& c, TailShared& sh) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const unsigned long long oa = __shfl_xor(a, off, 64), ob = __shfl_xor(b, off, 64);
        const unsigned oc = __shfl_xor(c, off, 64);
        a = oa < a ? oa : a;
        b = ob < b ? ob : b;
        c = oc > c ? oc : c;
    }
    if ((threadIdx.x & 63) == 0) { sh.r64[0][threadIdx.x >> 6] = a; sh.r64[1][threadIdx.x >> 6] = b; sh.r32[threadIdx.x >> 6] = c; }
    __syncthreads();
    a = sh.r64[0][0]; b = sh.r64[1][0]; c = sh.r32[0];
#pragma unroll
    for (int k = 1; k < kScanThreads / 64; ++k) {
        a = sh.r64[0][k] < a ? sh.r64[0][k] : a;
        b = sh.r64[1][k] < b ? sh.r64[1][k] : b;
        c = sh.r32[k] > c ? sh.r32[k] : c;
    }
    __syncthreads();
}

// Inclusive prefix sums, in place, of the `need` words pre[onset_pad(1)] .. pre[onset_pad(need)] (pre[0] = 0), in three
// steps: thread spans, span totals, fold.
__device__ __forceinline__ void block_prefix_u32(unsigned* pre, unsigned* thread_tot, int need) {
    const int tid = threadIdx.x;
    const int per = (need + kScanThreads - 1) / kScanThreads;
    const int lo = tid * per;
    const int hi = (lo + per < need) ? lo + per : need;
    unsigned run = 0;
    for (int k = lo; k < hi; ++k) {
        run += pre[onset_pad(k + 1)];
        pre[onset_pad(k + 1)] = run;
    }
    thread_tot[tid] = run;
    __syncthreads();
    if (tid < 64) {
        unsigned t0 = thread_tot[4 * tid], t1 = thread_tot[4 * tid + 1], t2 = thread_tot[4 * tid + 2], t3 = thread_tot[4 * tid + 3];
        unsigned tot = t0 + t1 + t2 + t3, inc = tot;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const unsigned o = __shfl_up(inc, off, 64);
            if (tid >= off) inc += o;
        }
        const unsigned ex = inc - tot;
        thread_tot[4 * tid] = ex;
        thread_tot[4 * tid + 1] = ex + t0;
        thread_tot[4 * tid + 2] = ex + t0 + t1;
        thread_tot[4 * tid + 3] = ex + t0 + t1 + t2;
    }
    __syncthreads();
    const unsigned add = thread_tot[tid];
    for (int k = lo; k < hi; ++k) pre[onset_pad(k + 1)] += add;
    __syncthreads();
}

// K4 of one onset workgroup `w`; writes rec[w].  Whole workgroup.
__device__ void tail_onset_range(const TailArgs& A, unsigned w, unsigned* pre_c, unsigned* thread_tot, unsigned* pre_x,
                                 TailShared& sh) {
    const int tid = threadIdx.x;
    const int window = A.window;
    const size_t nsamples = (size_t)A.nsamples;
    const size_t nout = nsamples - (size_t)window + 1;
    // the threshold (triangulateTDOA.py:41-42,46), from the exact noise sum
    unsigned long long noise_S;
    if (A.noise_in_scratch) {
        noise_S = A.sc->noise_S;                       // a part that does not hold the span: summed by the scan launch
    } else {
        const size_t nb = (size_t)A.noise_samples / 512;
        unsigned long long part = 0;
        for (size_t j = tid; j < nb; j += kScanThreads) part += A.cblk[j];
        for (size_t n = nb * 512 + tid; n < (size_t)A.noise_samples; n += kScanThreads) part += m_of(A.iq[2 * n], A.iq[2 * n + 1], A.up.off2);
        noise_S = block_sum_u64(part, sh);
    }
    float noise = (float)((double)noise_S / (4.0 * (double)A.noise_samples));
    if (noise == 0.f) noise = 1e-9f;
    const float thr_f = noise * A.factor;
    if (w == 0 && tid == 0) { A.sc->noise = noise; A.sc->thr = thr_f; }
    const double thr = (double)thr_f, thr_lo = thr * (1.0 - kOnsetGuard);
    const double scale = 0.25 / (double)window;
    // screening: block sums of this range + the blocks a window that starts in it can reach
    const size_t j0 = (size_t)w * kTailBlocks;
    const int cb = (window + 510) / 512 + 1;
    const size_t nblk_total = (nsamples + 511) / 512;
    size_t jend = j0 + kTailBlocks;
    if (512 * jend > nout) jend = (nout + 511) / 512;
    const int nloc = jend > j0 ? (int)(jend - j0) : 0;
    const int need = nloc ? nloc + cb - 1 : 0;
    for (int k = tid; k < need; k += kScanThreads) {
        const size_t j = j0 + k;
        pre_c[onset_pad(k + 1)] = j < nblk_total ? A.cblk[j] : 0u;
    }
    if (tid == 0) pre_c[0] = 0;
    __syncthreads();
    if (need) block_prefix_u32(pre_c, thread_tot, need);
    unsigned long long first = ~0ull, guard = ~0ull;
    unsigned below = 0;
    const uint16_t* iq16_all = reinterpret_cast<const uint16_t*>(A.iq);
    int kstart = 0;
    while (kstart < nloc) {                              // workgroup-uniform
        // the first block from kstart on that fails the proof
        unsigned long long mine = ~0ull;
        for (int k = kstart + tid; k < nloc; k += kScanThreads) {
            const unsigned U = pre_c[onset_pad(k + cb)] - pre_c[onset_pad(k)];
            if ((double)U * scale > thr_lo) { mine = (unsigned long long)k; break; }
        }
        const unsigned long long k1u = block_min_u64(mine, sh, 0);
        const int k1 = k1u == ~0ull ? nloc : (int)k1u;
        // the blocks in front of it are quiet: they enter the margin with their bound
        unsigned quiet = 0;
        for (int k = kstart + tid; k < k1; k += kScanThreads) {
            const unsigned U = pre_c[onset_pad(k + cb)] - pre_c[onset_pad(k)];
            quiet = U > quiet ? U : quiet;
        }
        quiet = block_max_u32(quiet, sh);
        below = quiet > below ? quiet : below;
        if (k1 >= nloc) break;
        // exact: the kOnsetOut positions from that block on (clipped to this range): prefix sums of 4|z|^2 in LDS,
        // window sums by difference
        const size_t o0 = (j0 + (size_t)k1) * 512;
        size_t o1 = o0 + kOnsetOut;
        if (o1 > jend * 512) o1 = jend * 512;
        if (o1 > nout) o1 = nout;
        const int npos = (int)(o1 - o0);
        const int needx = npos + window - 1;
        const uint16_t* iq16 = iq16_all + o0;
        for (int k = tid; k < needx; k += kScanThreads) {
            const unsigned v = iq16[k];
            pre_x[onset_pad(k + 1)] = m_of(v & 255u, v >> 8, A.up.off2);
        }
        if (tid == 0) pre_x[0] = 0;
        __syncthreads();
        block_prefix_u32(pre_x, thread_tot, needx);
        unsigned Sv[kOnsetOut / kScanThreads];
        unsigned long long best = ~0ull, best_lo = ~0ull;
#pragma unroll
        for (int i = 0; i < kOnsetOut / kScanThreads; ++i) {
            const int k = tid + i * kScanThreads;
            Sv[i] = 0;
            if (k < npos) {
                const unsigned S = pre_x[onset_pad(k + window)] - pre_x[onset_pad(k)];
                Sv[i] = S;
                const double ma = (double)S * scale;
                if (ma > thr_lo && best_lo == ~0ull) best_lo = o0 + k;
                if (ma > thr && best == ~0ull) best = o0 + k;
            }
        }
        // first crossing, first band position and -- in the same exchange -- the largest window sum of the thread's
        // positions in front of ITS first crossing; the positions between the workgroup's first crossing and the
        // thread's own are taken out again below (second pass over registers, only when there was a crossing)
        unsigned long long hit = best, hit_lo = best_lo;
        unsigned b = 0;
#pragma unroll
        for (int i = 0; i < kOnsetOut / kScanThreads; ++i) {
            const int k = tid + i * kScanThreads;
            if (k < npos && o0 + k < best) b = Sv[i] > b ? Sv[i] : b;
        }
        block_min2_max(hit, hit_lo, b, sh);
        if (hit != ~0ull) {                              // workgroup-uniform: only the positions in front of `hit` count
            b = 0;
#pragma unroll
            for (int i = 0; i < kOnsetOut / kScanThreads; ++i) {
                const int k = tid + i * kScanThreads;
                if (k < npos && o0 + k < hit) b = Sv[i] > b ? Sv[i] : b;
            }
            b = block_max_u32(b, sh);
        }
        below = b > below ? b : below;
        if (hit_lo != ~0ull && guard == ~0ull) guard = hit_lo;
        if (hit != ~0ull) { first = hit; break; }
        kstart = k1 + (npos + 511) / 512;
    }
    if (tid == 0) {
        TailRec r;
        r.first = first; r.guard = guard; r.below = below; r.pad = 0;
        A.rec[w] = r;
    }
}

// The last onset workgroup to arrive: records -> gj_onset, then the slot.
__device__ void tail_onset_finish(const TailArgs& A, TailShared& sh) {
    const int tid = threadIdx.x;
    const int window = A.window;
    if (A.valid) {
        unsigned long long wmin = ~0ull;
        for (unsigned w = tid; w < A.nct; w += kScanThreads)
            if (A.rec[w].first != ~0ull) { wmin = w; break; }
        const unsigned long long wstar = block_min_u64(wmin, sh, 0);
        const bool found = wstar != ~0ull;
        const unsigned wlast = found ? (unsigned)wstar : A.nct - 1;
        unsigned below = 0;
        unsigned long long guard = ~0ull;
        for (unsigned w = tid; w <= wlast; w += kScanThreads) {
            const TailRec r = A.rec[w];
            below = r.below > below ? r.below : below;
            guard = r.guard < guard ? r.guard : guard;
        }
        below = block_max_u32(below, sh);
        guard = block_min_u64(guard, sh, 1);
        const unsigned long long i0 = found ? A.rec[wstar].first : 0ull;
        unsigned long long S = 0;
        if (found)
            for (int k = tid; k < window; k += kScanThreads) S += m_of(A.iq[2 * (i0 + k)], A.iq[2 * (i0 + k) + 1], A.up.off2);
        S = block_sum_u64(S, sh);
        if (tid == 0) {
            const double thr = (double)A.sc->thr, scale = 0.25 / (double)window;
            gj_onset o;
            o.start_index = found ? A.sample0 + (long long)i0 + window / 2 : -1;
            o.noise_power = A.sc->noise;
            o.threshold = A.sc->thr;
            o.margin_hit = found ? (float)(((double)S * scale - thr) / thr) : 0.f;
            o.margin_before = (float)((thr - (double)below * scale) / thr);
            unsigned long long ig = guard;
            if (found && i0 < ig) ig = i0;
            o.guard_index = ig != ~0ull ? A.sample0 + (long long)ig + window / 2 : -1;
            *A.onset = o;
            sh.start = o.start_index;
        }
    } else if (tid == 0) {
        gj_onset o;
        o.start_index = -1; o.noise_power = 0.f; o.threshold = 0.f; o.margin_hit = 0.f; o.margin_before = 0.f; o.guard_index = -1;
        *A.onset = o;
        sh.start = -1;
    }
    __syncthreads();
    if (A.has_slot)
        tdoa_slot_body(A.iq, (size_t)A.slot_nsamples, sh.start, (size_t)A.slice_samples, A.slot, A.sample0, (size_t)A.total_samples,
                       (size_t)tid, (size_t)kScanThreads, tid == 0);
}

__global__ __launch_bounds__(kScanThreads) void scan_tail_kernel(TailArgs A) {
    __shared__ ThrShared sh_thr;
    __shared__ double sh_amp[16];
    __shared__ long long first_s;
    __shared__ TailShared sh;
    __shared__ unsigned pre_c[kTailPre];
    __shared__ unsigned thread_tot[kScanThreads];
    extern __shared__ unsigned pre_x[];                // kOnsetOut + window - 1 words + padding
    const int tid = threadIdx.x;
    unsigned b = blockIdx.x;
    // A few dozen workgroups, each a chain of short dependent phases, on the path to K5 -- beside K2, which keeps every
    // SIMD's issue slots busy and does not care about latency.  Raised wave priority lets these waves issue when they
    // are ready (their total instruction count is a rounding error for K2).  Only when the host says the chain is what
    // the step waits for (captures of the reference's size); behind a GiB-class K2 launch it is hidden either way and
    // K2's issue slots are the step (round 4 measured +1.2 % there with the priority raised across the board).
    if (A.hi_prio) __builtin_amdgcn_s_setprio(3);
    if (A.has_thr) {
        if (b == 0) {
            power_threshold_body(A.power, (size_t)A.nchunks, A.pct, A.ratio, A.stats, A.mask, sh_thr);
            return;
        }
        --b;
    }
    if (b == 0) {
        // amplitude totals; a part of a split capture
        // reports its first hit (made global with own_sample0) and the tail of its tile, the tile sums travel as they are
        const long long first = amp_block_first(A.tiles, (size_t)A.own_tiles, &first_s);
        if (first == 0x7fffffffffffffffll) {
            if (tid == 0) {
                if (A.is_part) { A.amp_part->first_index = -1; A.amp_part->count = 0; A.amp_part->sum = 0.0; A.amp_part->tail = 0.0; }
                else { A.amp->first_index = -1; A.amp->count = 0; A.amp->sum = 0.0; A.amp->mean = 0.f; A.amp->reserved = 0.f; }
            }
            return;
        }
        const double tail = amp_block_tail(A.own_iq, (size_t)A.own_samples, A.tiles, first, A.up, sh_amp);
        const double behind = amp_block_total(A.tiles, (size_t)A.own_tiles, (size_t)first / kAmpTileSamples, sh_amp);
        if (tid == 0) {
            const unsigned long long cnt = A.own_samples - (size_t)first;
            if (A.is_part) {
                A.amp_part->first_index = A.own_sample0 + first;
                A.amp_part->count = cnt;
                A.amp_part->sum = behind + tail;
                A.amp_part->tail = tail;
            } else {
                const double total = behind + tail;
                A.amp->first_index = first;
                A.amp->count = cnt;
                A.amp->sum = total;
                A.amp->mean = (float)(total / (double)cnt);
                A.amp->reserved = 0.f;
            }
        }
        return;
    }
    --b;
    // onset workgroups
    if (A.valid) tail_onset_range(A, b, pre_c, thread_tot, pre_x, sh);
    if (tid == 0) {
        const unsigned ticket = arrive_release(A.arrive);
        const int last = ticket == A.nct - 1;
        if (last) last_arriver_acquire(A.arrive);
        sh.last = last;
    }
    __syncthreads();
    if (!sh.last) return;
    tail_onset_finish(A, sh);
}

// One implementation for a whole capture (part == nullptr) and for one part of a capture split over GPUs.
// A part's buffer is [halo][own range]: the halo (whole 64-KiB tiles of the capture in front of the own range, at
// least window - 1 samples; none for the capture's first part) lets K4 evaluate every window that ENDS inside the
// own range, so that the parts' position ranges tile the capture; chunk powers and amplitude tiles cover the own
// range only.  A position inside the halo that crosses is reported by the part in front as well -- the combining
// rank takes the smallest index, so duplicates are harmless.
struct ScanPart {
    size_t halo_bytes;         // multiple of kScanTile
    long long buf_sample0;     // index of the buffer's first sample in the whole capture
    const uint8_t* d_noise;    // the capture's first 2 * noise_samples bytes
    size_t total_samples;      // of the whole capture
    AmpTile* d_tiles;          // [own tiles] out
    gj_amp_part* d_amp;        // out
};

struct ScanState {
    const uint8_t* d_iq;
    size_t nbytes, chunk_bytes, own_bytes, nsamples, own_samples, ntiles, skip, own_tiles, nchunks, tpc, noise_bytes;
    float eps, rssi_threshold, factor;
    int flags, noise_samples, window, valid;
    bool noise_here, track, is_part;
    float* d_power;
    gj_amp_stats* d_amp;
    gj_onset* d_onset;
    ScanPart part;
    size_t off_tiles, off_acc, off_blk, off_rec;
    unsigned nct;   // onset workgroups of the tail
};
static_assert(sizeof(ScanState) <= sizeof(ScanJob::state), "ScanJob::state too small");

bool scan_fusable(const uint8_t* d_iq, size_t nbytes, size_t chunk_bytes) {
    return chunk_bytes >= kScanTile && chunk_bytes % kScanTile == 0 && (reinterpret_cast<uintptr_t>(d_iq) & 15) == 0 && nbytes >= 2;
}

static int scan_begin_impl(gj_ctx* ctx, const uint8_t* d_iq, size_t nbytes, size_t chunk_bytes, float eps, int flags,
                           float* d_power, float rssi_threshold, gj_amp_stats* d_amp, int noise_samples, int window,
                           float factor, gj_onset* d_onset, const ScanPart* part, ScanJob& job) {
    if (noise_samples <= 0 || window <= 0) return fail(ctx, GJ_ERR_INVALID, "noise_samples and window must be > 0");
    if (window > kOnsetMaxWin) return fail(ctx, GJ_ERR_UNSUPPORTED, "window > %d", kOnsetMaxWin);
    ScanState st;
    memset(&st, 0, sizeof(st));
    st.d_iq = d_iq; st.nbytes = nbytes; st.chunk_bytes = chunk_bytes; st.eps = eps; st.flags = flags; st.d_power = d_power;
    st.rssi_threshold = rssi_threshold; st.d_amp = d_amp; st.noise_samples = noise_samples; st.window = window;
    st.factor = factor; st.d_onset = d_onset;
    st.is_part = part != nullptr;
    if (part) st.part = *part;
    const size_t halo = part ? part->halo_bytes : 0;
    st.own_bytes = nbytes - halo;
    st.nsamples = nbytes / 2;                         // of the buffer (halo + own)
    st.own_samples = st.own_bytes / 2;
    st.ntiles = (2 * st.nsamples + kScanTile - 1) / kScanTile;
    st.skip = halo / kScanTile;
    st.own_tiles = st.ntiles - st.skip;
    st.nchunks = gj_chunk_count(st.own_bytes, chunk_bytes);
    st.tpc = chunk_bytes / kScanTile;
    const size_t nblk = (st.nsamples + 511) / 512;
    if (st.ntiles > 0x7fffffffull) return fail(ctx, GJ_ERR_UNSUPPORTED, "capture too long");
    // workspace: [OnsetScratch][AmpTile x own tiles (whole capture only)][acc u64 x 2 x nchunks][c512 u32 x nblk]
    st.off_tiles = 256;
    st.off_acc = st.off_tiles + (part ? 0 : align_up((st.own_tiles + 1) * sizeof(AmpTile), 256));
    st.off_blk = st.off_acc + align_up(st.nchunks * 16, 256);
    // triangulateTDOA.py:39 speaks of the whole capture; a part must itself hold at least one window
    const size_t total_samples = part ? part->total_samples : st.nsamples;
    st.valid = d_onset && total_samples >= (size_t)noise_samples + (size_t)window && st.nsamples >= (size_t)window;
    // K4's noise sum: a buffer that starts with the span gets it from the scan's own block sums (tail kernel); a part
    // that does not hold the span brings a copy (d_noise), summed by one extra workgroup of the scan launch
    st.noise_bytes = (size_t)2 * noise_samples;
    st.noise_here = !part || (part->buf_sample0 == 0 && st.noise_bytes <= nbytes);
    if (st.valid && !st.noise_here && !part->d_noise)
        return fail(ctx, GJ_ERR_INVALID, "this part does not hold the capture's noise span (%zu bytes): d_noise is required", st.noise_bytes);
    // When the offset is a half-integer (off2 odd) no component of 2u - off2 is zero, so every amplitude is at
    // least sqrt(2) * half_scale (0.0055 for the default unpack): a threshold below that makes every sample a hit
    // and the first index needs no tracking.  With an integer offset (gj_set_unpack(128, ...)) amplitudes can be
    // zero and the shortcut never applies.  Same float expression as the kernels' `a > thr`.
    const Unpack upk = unpack_of(ctx);
    const bool all_hit = (ctx->off2 & 1) && (sqrtf(2.0f) * upk.half_scale > rssi_threshold);
    st.track = !all_hit;
    job.ntiles = st.ntiles;
    job.nchunks = st.nchunks;
    st.off_rec = align_up(st.off_blk + (nblk + 16) * sizeof(unsigned), 256);
    {
        const size_t nout = st.valid ? st.nsamples - (size_t)window + 1 : 0;
        // nobody asked for the onset (K3 alone): no onset workgroups at all
        const size_t nct = st.valid ? ((nout + 511) / 512 + kTailBlocks - 1) / kTailBlocks : (d_onset ? 1 : 0);
        if (nct > 0x7ffffff0ull) return fail(ctx, GJ_ERR_UNSUPPORTED, "capture too long");
        st.nct = (unsigned)nct;
    }
    memcpy(job.state, &st, sizeof(st));
    job.ws_bytes = align_up(st.off_rec + (size_t)st.nct * sizeof(TailRec), 256);
    job.ws = nullptr;
    return GJ_OK;
}

int scan_begin(gj_ctx* ctx, const uint8_t* d_iq, size_t nbytes, size_t chunk_bytes, float eps, int flags, float* d_power,
               float rssi_threshold, gj_amp_stats* d_amp, int noise_samples, int window, float factor, gj_onset* d_onset,
               ScanJob& job) {
    if (!scan_fusable(d_iq, nbytes, chunk_bytes)) return fail(ctx, GJ_ERR_UNSUPPORTED, "the fused scan needs chunk_bytes to be a multiple of 65536 and a 16-byte aligned capture");
    return scan_begin_impl(ctx, d_iq, nbytes, chunk_bytes, eps, flags, d_power, rssi_threshold, d_amp, noise_samples, window,
                           factor, d_onset, nullptr, job);
}

int scan_start(gj_ctx* ctx, ScanJob& job) {
    ScanState st;
    memcpy(&st, job.state, sizeof(st));
    // chunks of several tiles accumulate by atomics; nothing else needs clearing (the tail kernel reads only what this
    // job's launches have written, and its arrival counter is kept at zero by its own last workgroup)
    if (st.tpc > 1) GJ_HIP(ctx, hipMemsetAsync(job.ws + st.off_acc, 0, st.nchunks * 16, ctx->stream));
    return GJ_OK;
}

int scan_range(gj_ctx* ctx, const ScanJob& job, size_t tile0, size_t tile1) {
    ScanState st;
    memcpy(&st, job.state, sizeof(st));
    if (tile1 > st.ntiles) tile1 = st.ntiles;
    if (tile0 >= tile1) return GJ_OK;
    OnsetScratch* sc = reinterpret_cast<OnsetScratch*>(job.ws);
    AmpTile* tiles = st.is_part ? st.part.d_tiles : reinterpret_cast<AmpTile*>(job.ws + st.off_tiles);
    unsigned long long* acc = reinterpret_cast<unsigned long long*>(job.ws + st.off_acc);
    unsigned* cblk = reinterpret_cast<unsigned*>(job.ws + st.off_blk);
    const Unpack upk = unpack_of(ctx);
    const unsigned n = (unsigned)(tile1 - tile0);
    // a part without the capture's noise span: one more workgroup, with the first range of tiles
    const bool noise_wg = st.valid && !st.noise_here && tile0 == 0;
    const unsigned noise_block = noise_wg ? n : 0xffffffffu;
    const dim3 grid(n + (noise_wg ? 1u : 0u));
    if (st.track)
        hipLaunchKernelGGL(stream_scan_kernel<true>, grid, dim3(kScanThreads), 0, ctx->stream, st.d_iq, st.nsamples,
                           st.own_bytes, st.chunk_bytes, (unsigned)st.tpc, st.eps, st.flags, st.d_power, acc, st.rssi_threshold,
                           tiles, cblk, upk, (unsigned)st.skip, (unsigned)tile0, noise_block, st.part.d_noise, st.noise_samples, sc);
    else
        hipLaunchKernelGGL(stream_scan_kernel<false>, grid, dim3(kScanThreads), 0, ctx->stream, st.d_iq, st.nsamples,
                           st.own_bytes, st.chunk_bytes, (unsigned)st.tpc, st.eps, st.flags, st.d_power, acc, st.rssi_threshold,
                           tiles, cblk, upk, (unsigned)st.skip, (unsigned)tile0, noise_block, st.part.d_noise, st.noise_samples, sc);
    GJ_LAUNCH_CHECK(ctx);
    return GJ_OK;
}

int scan_end(gj_ctx* ctx, const ScanJob& job, const ScanExtra* extra) {
    ScanState st;
    memcpy(&st, job.state, sizeof(st));
    OnsetScratch* sc = reinterpret_cast<OnsetScratch*>(job.ws);
    AmpTile* tiles = st.is_part ? st.part.d_tiles : reinterpret_cast<AmpTile*>(job.ws + st.off_tiles);
    unsigned long long* acc = reinterpret_cast<unsigned long long*>(job.ws + st.off_acc);
    const size_t halo = st.is_part ? st.part.halo_bytes : 0;
    if (st.tpc > 1) {
        hipLaunchKernelGGL(chunk_power_finalize_kernel, dim3((unsigned)((st.nchunks + 255) / 256)), dim3(256), 0,
                           ctx->stream, acc, st.nchunks, st.own_bytes, st.chunk_bytes, st.eps, st.flags, st.d_power, ctx->off2, true);
        GJ_LAUNCH_CHECK(ctx);
    }
    {
        float edge;
        if (st.d_power && power_edge_value(st.nchunks, st.own_bytes, st.chunk_bytes, st.flags, &edge)) {
            unsigned bits;
            memcpy(&bits, &edge, 4);
            GJ_HIP(ctx, hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(st.d_power + (st.nchunks - 1)), (int)bits, 1, ctx->stream));
        }
    }
    TailArgs A;
    memset(&A, 0, sizeof(A));
    A.iq = st.d_iq;
    A.nsamples = st.nsamples;
    A.sample0 = st.is_part ? st.part.buf_sample0 : 0ll;
    A.total_samples = st.is_part ? st.part.total_samples : st.nsamples;
    A.is_part = st.is_part;
    A.valid = st.valid;
    A.power = st.d_power;
    A.nchunks = st.nchunks;
    if (extra && extra->d_stats && st.nchunks) {
        A.has_thr = 1;
        A.pct = extra->pct;
        A.ratio = (float)pow(10.0, (double)extra->rise_db / 10.0);
        A.stats = extra->d_stats;
        A.mask = extra->d_mask;
    }
    A.own_iq = st.d_iq + halo;
    A.own_samples = st.own_samples;
    A.own_tiles = st.own_tiles;
    A.tiles = tiles;
    A.amp = st.d_amp;
    A.amp_part = st.part.d_amp;
    A.own_sample0 = A.sample0 + (long long)(halo / 2);
    A.window = st.window;
    A.noise_samples = st.noise_samples;
    A.noise_in_scratch = st.valid && !st.noise_here;
    A.factor = st.factor;
    A.cblk = reinterpret_cast<const unsigned*>(job.ws + st.off_blk);
    A.sc = sc;
    A.rec = reinterpret_cast<TailRec*>(job.ws + st.off_rec);
    A.arrive = ctx->d_sync + kSyncTail;
    A.nct = st.nct;
    A.onset = st.d_onset;
    if (extra && extra->d_slot) {
        if (extra->slice_samples < 1) return fail(ctx, GJ_ERR_INVALID, "n_samples must be >= 1");
        if ((reinterpret_cast<uintptr_t>(extra->d_slot) & 15) != 0) return fail(ctx, GJ_ERR_INVALID, "slot must be 16-byte aligned");
        A.slot = extra->d_slot;
        A.slice_samples = extra->slice_samples;
        A.slot_nsamples = extra->slot_buf_bytes ? extra->slot_buf_bytes / 2 : st.nsamples;
        // the tail's last workgroup cuts the slot alone: fine for the reference's slices (50 000 samples = 100 KB, a few
        // round trips), not for the benchmark's 2^19-sample ones (1 MiB by one workgroup is latency-bound: ~70 us) --
        // those get the grid-wide copy kernel behind the tail, one launch more on a chain that a GiB-class K2 hides
        A.has_slot = 2 * extra->slice_samples <= kTailSlotBytes;
    }
    A.hi_prio = st.nbytes <= kTailPrioBytes;
    A.up = unpack_of(ctx);
    const int needx = kOnsetOut + st.window - 1;
    const size_t dyn = (size_t)(needx + needx / 32 + 8) * sizeof(unsigned);
    const unsigned grid = (A.has_thr ? 1u : 0u) + 1u + st.nct;
    hipLaunchKernelGGL(scan_tail_kernel, dim3(grid), dim3(kScanThreads), dyn, ctx->stream, A);
    GJ_LAUNCH_CHECK(ctx);
    if (A.slot && !A.has_slot)
        return launch_tdoa_slot(ctx, st.d_iq, 2 * (size_t)A.slot_nsamples, &st.d_onset->start_index, (size_t)A.slice_samples, A.slot,
                                A.sample0, (size_t)A.total_samples);
    return GJ_OK;
}

static int stream_scan_impl(gj_ctx* ctx, const uint8_t* d_iq, size_t nbytes, size_t chunk_bytes, float eps, int flags,
                            float* d_power, float rssi_threshold, gj_amp_stats* d_amp, int noise_samples, int window,
                            float factor, gj_onset* d_onset, const ScanPart* part, const ScanExtra* extra = nullptr) {
    ScanJob job;
    int rc = scan_begin_impl(ctx, d_iq, nbytes, chunk_bytes, eps, flags, d_power, rssi_threshold, d_amp, noise_samples, window,
                             factor, d_onset, part, job);
    if (rc) return rc;
    rc = ensure_workspace(ctx, job.ws_bytes);
    if (rc) return rc;
    job.ws = ctx->ws;
    rc = scan_start(ctx, job);
    if (!rc) rc = scan_range(ctx, job, 0, job.ntiles);
    if (!rc) rc = scan_end(ctx, job, extra);
    return rc;
}

// K3 and / or K4 WITHOUT a power map (gj_amp_stats_dev, gj_onset_dev, their *_u8 forms, and captures whose chunk size the
// fused pass does not take): the same pass and the same tail -- neither quantity depends on the chunk size, so the pass
// runs with one-tile chunks and its power map lands in the workspace; an output nobody asked for (d_amp or d_onset
// == nullptr) gets no role in the tail (onset) or a scratch record (amplitude: the pass produces the tile sums anyway).
// Same tiles, same integer window sums, same fixed summation orders => the bits of gj_capture_scan_dev.  Until round 6
// these entry points ran chains of their own (amp_tiles -> amp_finalize; onset_noise -> onset_coarse<8> -> onset_scan ->
// onset_finalize: 0.62 ms per GiB for ONE onset against 0.23 ms for the whole fused scan + tail).
// A capture that is not 16-byte aligned (a caller's pointer into the middle of a buffer: resident captures, staging
// areas and parts are all aligned) is first copied to an aligned place in the workspace -- one device-to-device copy,
// still faster than the byte-wise kernels it replaces, and the results no longer depend on the alignment.
int launch_amp_onset(gj_ctx* ctx, const uint8_t* d_iq, size_t nbytes, float rssi_threshold, gj_amp_stats* d_amp,
                     int noise_samples, int window, float factor, gj_onset* d_onset) {
    if (!d_amp && !d_onset) return GJ_OK;
    if (!d_onset) { noise_samples = 1; window = 1; factor = 1.f; }
    if (!d_amp) rssi_threshold = -1.0f;                  // every sample a hit: the pass need not track the first
    const bool aligned = (reinterpret_cast<uintptr_t>(d_iq) & 15) == 0;
    ScanJob job;
    int rc = scan_begin_impl(ctx, d_iq, nbytes, kScanTile, 0.f, 0, nullptr, rssi_threshold, d_amp, noise_samples, window, factor,
                             d_onset, nullptr, job);
    if (rc) return rc;
    const size_t off_power = job.ws_bytes;
    const size_t off_amp = off_power + align_up((job.nchunks + 1) * sizeof(float), 256);
    const size_t off_copy = off_amp + 256;
    rc = ensure_workspace(ctx, off_copy + (aligned ? 0 : align_up(nbytes, 256)));
    if (rc) return rc;
    if (!aligned && nbytes) {
        GJ_HIP(ctx, hipMemcpyAsync(ctx->ws + off_copy, d_iq, nbytes, hipMemcpyDeviceToDevice, ctx->stream));
        d_iq = ctx->ws + off_copy;
    }
    rc = scan_begin_impl(ctx, d_iq, nbytes, kScanTile, 0.f, 0, reinterpret_cast<float*>(ctx->ws + off_power), rssi_threshold,
                         d_amp ? d_amp : reinterpret_cast<gj_amp_stats*>(ctx->ws + off_amp), noise_samples, window, factor, d_onset,
                         nullptr, job);
    if (rc) return rc;
    job.ws = ctx->ws;
    rc = scan_start(ctx, job);
    if (!rc) rc = scan_range(ctx, job, 0, job.ntiles);
    if (!rc) rc = scan_end(ctx, job, nullptr);
    return rc;
}

int launch_amp_stats(gj_ctx* ctx, const uint8_t* d_iq, size_t nbytes, float threshold, gj_amp_stats* d_out) {
    return launch_amp_onset(ctx, d_iq, nbytes, threshold, d_out, 0, 0, 0.f, nullptr);
}

int launch_onset(gj_ctx* ctx, const uint8_t* d_iq, size_t nbytes, int noise_samples, int window, float factor,
                 gj_onset* d_out) {
    return launch_amp_onset(ctx, d_iq, nbytes, 0.f, nullptr, noise_samples, window, factor, d_out);
}

int launch_stream_scan(gj_ctx* ctx, const uint8_t* d_iq, size_t nbytes, size_t chunk_bytes, float eps, int flags,
                       float* d_power, float rssi_threshold, gj_amp_stats* d_amp, int noise_samples, int window,
                       float factor, gj_onset* d_onset, const ScanExtra* extra) {
    if (!scan_fusable(d_iq, nbytes, chunk_bytes)) {   // odd chunk sizes / unaligned captures: K1 alone, then the pass for K3 + K4
        int rc = launch_chunk_power(ctx, d_iq, nbytes, chunk_bytes, eps, flags, d_power);
        if (!rc) rc = launch_amp_onset(ctx, d_iq, nbytes, rssi_threshold, d_amp, noise_samples, window, factor, d_onset);
        if (!rc && extra && extra->d_stats && gj_chunk_count(nbytes, chunk_bytes))
            rc = launch_power_threshold(ctx, d_power, gj_chunk_count(nbytes, chunk_bytes), extra->pct, extra->rise_db, extra->d_stats, extra->d_mask);
        if (!rc && extra && extra->d_slot)
            rc = launch_tdoa_slot(ctx, d_iq, nbytes, &d_onset->start_index, extra->slice_samples, extra->d_slot);
        return rc;
    }
    return stream_scan_impl(ctx, d_iq, nbytes, chunk_bytes, eps, flags, d_power, rssi_threshold, d_amp, noise_samples,
                            window, factor, d_onset, nullptr, extra);
}

// One part of a capture split over GPUs (include/gpsjam.h, gj_part_scan_dev).
int launch_part_scan(gj_ctx* ctx, const gj_part_view& v, size_t chunk_bytes, float eps, int flags, float* d_power,
                     float rssi_threshold, void* d_tiles, gj_amp_part* d_amp, int noise_samples, int window, float factor,
                     gj_onset* d_onset, const ScanExtra* extra) {
    if (v.own_first_byte < v.buf_first_byte || v.own_first_byte + v.own_bytes > v.buf_first_byte + v.buf_bytes ||
        v.own_first_byte + v.own_bytes > v.total_bytes)
        return fail(ctx, GJ_ERR_INVALID, "the own range does not lie inside the buffer / the capture");
    const size_t halo = v.own_first_byte - v.buf_first_byte;
    if (chunk_bytes < kScanTile || chunk_bytes % kScanTile || (reinterpret_cast<uintptr_t>(v.d_buf) & 15) || halo % kScanTile ||
        v.own_first_byte % chunk_bytes || (v.buf_first_byte & 1) || v.own_bytes < 2)
        return fail(ctx, GJ_ERR_UNSUPPORTED, "a part needs chunk_bytes and a halo that are multiples of 65536, an own range that "
                                             "starts on a chunk boundary and a 16-byte aligned buffer");
    if (v.own_first_byte + v.own_bytes != v.total_bytes && v.own_bytes % chunk_bytes)
        return fail(ctx, GJ_ERR_UNSUPPORTED, "only the capture's last part may end inside a chunk");
    if (v.own_first_byte != 0 && halo / 2 + 1 < (size_t)window)
        return fail(ctx, GJ_ERR_INVALID, "the halo must hold at least window - 1 samples");
    ScanPart p;
    p.halo_bytes = halo;
    p.buf_sample0 = (long long)(v.buf_first_byte / 2);
    p.d_noise = v.d_noise;
    p.total_samples = v.total_bytes / 2;
    p.d_tiles = static_cast<AmpTile*>(d_tiles);
    p.d_amp = d_amp;
    // the scan covers [halo][own]; what lies behind the own range in the buffer (the slice tail) is not scanned
    return stream_scan_impl(ctx, v.d_buf, halo + v.own_bytes, chunk_bytes, eps, flags, d_power, rssi_threshold, nullptr,
                            noise_samples, window, factor, d_onset, &p, extra);
}

size_t amp_tile_count(size_t nbytes) { return (nbytes / 2 * 2 + kScanTile - 1) / kScanTile; }

int launch_amp_combine(gj_ctx* ctx, const void* d_tiles, size_t ntiles, const gj_amp_part* d_parts, int n_parts,
                       size_t total_bytes, gj_amp_stats* d_out) {
    if (n_parts < 1) return fail(ctx, GJ_ERR_INVALID, "n_parts must be >= 1");
    if (ntiles != amp_tile_count(total_bytes)) return fail(ctx, GJ_ERR_INVALID, "a capture of %zu bytes has %zu tiles, not %zu", total_bytes, amp_tile_count(total_bytes), ntiles);
    hipLaunchKernelGGL(amp_combine_kernel, dim3(1), dim3(256), 0, ctx->stream, static_cast<const AmpTile*>(d_tiles), ntiles,
                       d_parts, n_parts, total_bytes / 2, d_out);
    GJ_LAUNCH_CHECK(ctx);
    return GJ_OK;
}

// ---------------------------------------------------------------------------------------
// byte histogram of raw_chunk[::stride] for every chunk the waterfall keeps
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void histogram_kernel(const uint8_t* __restrict__ iq, size_t nbytes, size_t chunk_bytes,
                                                        size_t min_bytes, int stride, size_t per_chunk, size_t total,
                                                        unsigned long long* __restrict__ hist) {
    __shared__ unsigned h[256];
    h[threadIdx.x] = 0;
    __syncthreads();
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const size_t c = i / per_chunk, k = i % per_chunk;
        const size_t off = c * chunk_bytes;
        const size_t len = (nbytes - off < chunk_bytes) ? nbytes - off : chunk_bytes;
        if (len < min_bytes) continue;
        const size_t p = k * (size_t)stride;
        if (p < len) atomicAdd(&h[iq[off + p]], 1u);
    }
    __syncthreads();
    if (h[threadIdx.x]) atomicAdd(&hist[threadIdx.x], (unsigned long long)h[threadIdx.x]);
}

int launch_histogram(gj_ctx* ctx, const uint8_t* d_iq, size_t nbytes, size_t chunk_samples, int nperseg, int stride,
                     unsigned long long* d_hist) {
    if (stride <= 0 || chunk_samples == 0) return fail(ctx, GJ_ERR_INVALID, "bad stride/chunk");
    GJ_HIP(ctx, hipMemsetAsync(d_hist, 0, 256 * sizeof(unsigned long long), ctx->stream));
    const size_t chunk_bytes = 2 * chunk_samples;
    const size_t nchunks = (nbytes + chunk_bytes - 1) / chunk_bytes;
    if (nchunks == 0) return GJ_OK;
    const size_t per_chunk = (chunk_bytes + stride - 1) / stride;
    const size_t total = nchunks * per_chunk;
    size_t blocks = (total + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(histogram_kernel, dim3((unsigned)blocks), dim3(256), 0, ctx->stream, d_iq, nbytes, chunk_bytes,
                       (size_t)2 * nperseg, stride, per_chunk, total, d_hist);
    GJ_LAUNCH_CHECK(ctx);
    return GJ_OK;
}

}   // namespace gj
