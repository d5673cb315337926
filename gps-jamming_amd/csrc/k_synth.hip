// Synthetic RTL-SDR capture generator, bit-identical to gpsjam/synth.py (integer-only):
// splitmix64 counter hash -> centred Irwin-Hall(8) variate -> fixed-point gain ->
// truncate toward zero -> clip to int8 -> +128.  Mirrors the value distribution produced by
// the reference's simulator chain (simulate/frontend/weaken_gps.py:27-28,
// add_jammer_and_mix.py:170-177).  Used by bench.py and the full-size GPU tests so that
// gigabyte inputs never have to be shipped.
#include "gj_common.h"

namespace gj {

__device__ __forceinline__ unsigned long long sm64(unsigned long long x) {
    x += 0x9E3779B97F4A7C15ull;
    unsigned long long z = x;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

__device__ __forceinline__ long long lanes_sum(unsigned long long h) {
    return (long long)((h & 0xffffull) + ((h >> 16) & 0xffffull) + ((h >> 32) & 0xffffull) + (h >> 48));
}

__device__ __forceinline__ long long gauss(unsigned long long key, long long idx) {
    const unsigned long long c = (unsigned long long)idx * 2ull + key;
    return lanes_sum(sm64(c)) + lanes_sum(sm64(c + 1ull)) - 262140ll;
}

__device__ __forceinline__ unsigned synth_byte(const gj_synth_params& p, long long n, int comp) {
    long long v = (gauss(p.key_noise, 2 * n + comp) * (long long)p.noise_k) >> 16;
    const long long src = n - p.delay;
    if (src >= p.jam_start && src < p.jam_end) v += (gauss(p.key_common, 2 * src + comp) * (long long)p.jam_k) >> 16;
    v += comp ? p.dc_q_q8 : p.dc_i_q8;
    long long t = (v >= 0) ? (v >> 8) : -((-v) >> 8);
    t = t < -128 ? -128 : (t > 127 ? 127 : t);
    return (unsigned)(t + 128);
}

__global__ __launch_bounds__(256) void synth_kernel(gj_synth_params p, long long first_sample, size_t n_samples,
                                                    uint8_t* __restrict__ out) {
    const size_t ngroups = (n_samples + 7) / 8;
    for (size_t gidx = blockIdx.x * (size_t)blockDim.x + threadIdx.x; gidx < ngroups;
         gidx += (size_t)gridDim.x * blockDim.x) {
        const size_t s0 = gidx * 8;
        unsigned w[4] = {0, 0, 0, 0};
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            if (s0 + k < n_samples) {
                const long long n = first_sample + (long long)(s0 + k);
                const unsigned pair = synth_byte(p, n, 0) | (synth_byte(p, n, 1) << 8);
                w[k >> 1] |= pair << (16 * (k & 1));
            }
        }
        if (s0 + 8 <= n_samples && ((reinterpret_cast<uintptr_t>(out) & 15) == 0)) {
            reinterpret_cast<uint4*>(out)[gidx] = uint4{w[0], w[1], w[2], w[3]};
        } else {
            for (int k = 0; k < 8 && s0 + k < n_samples; ++k) {
                const unsigned pair = (w[k >> 1] >> (16 * (k & 1))) & 0xffffu;
                out[2 * (s0 + k)] = (uint8_t)(pair & 255u);
                out[2 * (s0 + k) + 1] = (uint8_t)(pair >> 8);
            }
        }
    }
}

// ---------------------------------------------------------------------------------------
// result vector of one stream (see gj_pack_result_dev in gpsjam.h)
// ---------------------------------------------------------------------------------------
// block = 64 bins x 16 row lanes; blocks [0, nperseg/64) reduce the waterfall to its mean
// spectrum, the following blocks copy the header, the power map and the pair block
// (blockIdx.x / gridDim.x: the position inside ONE result vector's grid row; blockIdx.y is the caller's)
__device__ __forceinline__ void pack_result_body(size_t n_chunks, const float* __restrict__ power,
                                                 const float* __restrict__ stats, const gj_amp_stats* __restrict__ amp,
                                                 const gj_onset* __restrict__ onset, const float* __restrict__ psd,
                                                 size_t rows, int nperseg, int rank, int n_pairs, int pair_cap,
                                                 const int* __restrict__ pairs, const int* __restrict__ lags,
                                                 const float* __restrict__ peaks, const float* __restrict__ margins,
                                                 double* __restrict__ out, float (*part)[64]) {
    const unsigned spec_blocks = (unsigned)((nperseg + 63) / 64);
    if (blockIdx.x < spec_blocks) {
        const int kx = threadIdx.x & 63, ry = threadIdx.x >> 6;
        const int k = blockIdx.x * 64 + kx;
        float s = 0.f;
        if (k < nperseg)
            for (size_t r = ry; r < rows; r += 16) s += psd[r * (size_t)nperseg + k];
        part[ry][kx] = s;
        __syncthreads();
        if (ry == 0 && k < nperseg) {
            float t = 0.f;
#pragma unroll
            for (int j = 0; j < 16; ++j) t += part[j][kx];
            out[GJ_RESULT_HEADER + n_chunks + k] = rows ? (double)(t / (float)rows) : 0.0;
        }
        return;
    }
    const size_t head = GJ_RESULT_HEADER + n_chunks;
    const size_t pair0 = head + (size_t)nperseg;
    const size_t total = head + (size_t)GJ_RESULT_PAIR_FIELDS * pair_cap;
    for (size_t i = (blockIdx.x - spec_blocks) * (size_t)blockDim.x + threadIdx.x; i < total;
         i += (size_t)(gridDim.x - spec_blocks) * blockDim.x) {
        double v = 0.0;
        size_t dst = i;
        if (i < GJ_RESULT_HEADER) {
            switch (i) {
                case 0: v = (double)n_chunks; break;
                case 1: v = stats[0]; break;
                case 2: v = stats[1]; break;
                case 3: v = stats[2]; break;
                case 4: v = (double)amp->first_index; break;
                case 5: v = (double)amp->count; break;
                case 6: v = amp->mean; break;
                case 7: v = (double)onset->start_index; break;
                case 8: v = rank == 0 ? 0.0 : (double)GJ_LAG_INVALID; break;   // lag against antenna 0: filled in by the solver's owner
                case 9: v = 0.0; break;
                case 10: v = onset->noise_power; break;
                case 11: v = (double)rows; break;
                case 12: v = (double)nperseg; break;
                case 13: v = (double)rank; break;
                case 14: v = (double)n_pairs; break;
                case 15: v = (double)pair_cap; break;
                case 16: v = onset->margin_hit; break;      // the decision margins travel with the result (gj_onset)
                case 17: v = onset->margin_before; break;
                case 18: v = (double)onset->guard_index; break;
                case 19: v = onset->threshold; break;
                case 20: v = (double)rank; break;           // antenna: the stream's own capture
                case 22: v = 1.0; break;                    // parts: the whole capture
                case 26: v = amp->sum; break;
                case 32: case 33: case 34: case 35:         // the gj_onset record as it is (32 bytes)
                    v = reinterpret_cast<const double*>(onset)[i - 32]; break;
                case 36: case 37: case 38: case 39:         // the gj_amp_stats record as it is (32 bytes)
                    v = reinterpret_cast<const double*>(amp)[i - 36]; break;
                default: v = 0.0;
            }
        } else if (i < head) {
            v = power[i - GJ_RESULT_HEADER];
        } else {   // pair block: {i, j, lag, peak, margin} per pair this stream solved, zero-padded to the capacity
            const size_t q = i - head;
            const int pr = (int)(q / GJ_RESULT_PAIR_FIELDS), fld = (int)(q % GJ_RESULT_PAIR_FIELDS);
            dst = pair0 + q;
            if (pr < n_pairs) {
                switch (fld) {
                    case 0: v = (double)pairs[2 * pr]; break;
                    case 1: v = (double)pairs[2 * pr + 1]; break;
                    case 2: v = (double)lags[pr]; break;
                    case 3: v = peaks[pr]; break;
                    default: v = margins[pr];
                }
            }
        }
        out[dst] = v;
    }
}

__global__ __launch_bounds__(1024) void pack_result_kernel(size_t n_chunks, const float* __restrict__ power,
                                                           const float* __restrict__ stats,
                                                           const gj_amp_stats* __restrict__ amp,
                                                           const gj_onset* __restrict__ onset, const float* __restrict__ psd,
                                                           size_t rows, int nperseg, int rank, int n_pairs, int pair_cap,
                                                           const int* __restrict__ pairs, const int* __restrict__ lags,
                                                           const float* __restrict__ peaks, const float* __restrict__ margins,
                                                           double* __restrict__ out) {
    __shared__ float part[16][64];
    pack_result_body(n_chunks, power, stats, amp, onset, psd, rows, nperseg, rank, n_pairs, pair_cap, pairs, lags, peaks,
                     margins, out, part);
}

int launch_pack_result(gj_ctx* ctx, size_t n_chunks, const float* d_power, const float* d_stats, const gj_amp_stats* d_amp,
                       const gj_onset* d_onset, const float* d_psd, size_t rows, int nperseg, int rank, int n_pairs,
                       int pair_cap, const int32_t* d_pairs, const int32_t* d_lags, const float* d_peaks,
                       const float* d_margins, double* d_out) {
    const unsigned spec_blocks = (unsigned)((nperseg + 63) / 64);
    size_t copy_blocks = (GJ_RESULT_HEADER + n_chunks + (size_t)GJ_RESULT_PAIR_FIELDS * pair_cap + 1023) / 1024;
    if (copy_blocks > 256) copy_blocks = 256;
    hipLaunchKernelGGL(pack_result_kernel, dim3(spec_blocks + (unsigned)copy_blocks), dim3(1024), 0, ctx->stream, n_chunks,
                       d_power, d_stats, d_amp, d_onset, d_psd, rows, nperseg, rank, n_pairs, pair_cap, d_pairs, d_lags,
                       d_peaks, d_margins, d_out);
    GJ_LAUNCH_CHECK(ctx);
    return GJ_OK;
}

// ---------------------------------------------------------------------------------------
// result vector of ONE PART of a capture split over GPUs (gj_pack_part_dev): the same header, then what the
// combining rank needs to rebuild the capture's arrays -- own chunk powers, own amplitude tiles, solved pairs, own
// PSD rows (float32, two per double slot)
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void pack_part_kernel(gj_part_pack a, double* __restrict__ out) {
    const size_t o_power = GJ_RESULT_HEADER;
    const size_t o_tiles = o_power + a.chunk_cap;
    const size_t o_pairs = o_tiles + 2 * a.tile_cap;
    const size_t o_rows = o_pairs + (size_t)GJ_RESULT_PAIR_FIELDS * a.pair_cap;
    const size_t n_front = o_rows;
    const size_t row_floats = a.rows_cap * (size_t)a.nperseg;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    const size_t gid = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    for (size_t i = gid; i < n_front; i += stride) {
        double v = 0.0;
        if (i < GJ_RESULT_HEADER) {
            switch (i) {
                case 0: v = (double)a.n_chunks; break;
                case 4: v = (double)a.d_amp->first_index; break;
                case 5: v = (double)a.d_amp->count; break;
                case 7: v = (double)a.d_onset->start_index; break;
                case 8: v = (double)GJ_LAG_INVALID; break;
                case 10: v = a.d_onset->noise_power; break;
                case 11: v = (double)a.rows; break;
                case 12: v = (double)a.nperseg; break;
                case 13: v = (double)a.rank; break;
                case 14: v = (double)a.n_pairs; break;
                case 15: v = (double)a.pair_cap; break;
                case 16: v = a.d_onset->margin_hit; break;
                case 17: v = a.d_onset->margin_before; break;
                case 18: v = (double)a.d_onset->guard_index; break;
                case 19: v = a.d_onset->threshold; break;
                case 20: v = (double)a.antenna; break;
                case 21: v = (double)a.part; break;
                case 22: v = (double)a.parts; break;
                case 23: v = (double)a.first_chunk; break;
                case 24: v = (double)a.first_row; break;
                case 25: v = (double)a.first_sample; break;
                case 26: v = a.d_amp->sum; break;
                case 27: v = a.d_amp->tail; break;
                case 28: v = (double)a.n_tiles; break;
                case 29: v = (double)a.first_tile; break;
                case 32: case 33: case 34: case 35: v = reinterpret_cast<const double*>(a.d_onset)[i - 32]; break;
                case 36: case 37: case 38: case 39: v = reinterpret_cast<const double*>(a.d_amp)[i - 36]; break;
                default: v = 0.0;
            }
        } else if (i < o_tiles) {
            const size_t c = i - o_power;
            v = c < a.n_chunks ? (double)a.d_power[c] : 0.0;
        } else if (i < o_pairs) {
            const size_t q = i - o_tiles;                       // (sum, first) of tile q / 2: the 16-byte record as it is
            v = (q / 2 < a.n_tiles) ? reinterpret_cast<const double*>(a.d_tiles)[q] : 0.0;
        } else {
            const size_t q = i - o_pairs;
            const int pr = (int)(q / GJ_RESULT_PAIR_FIELDS), fld = (int)(q % GJ_RESULT_PAIR_FIELDS);
            if (pr < a.n_pairs) {
                switch (fld) {
                    case 0: v = (double)a.d_pairs[2 * pr]; break;
                    case 1: v = (double)a.d_pairs[2 * pr + 1]; break;
                    case 2: v = (double)a.d_lags[pr]; break;
                    case 3: v = a.d_peaks[pr]; break;
                    default: v = a.d_margins[pr];
                }
            }
        }
        out[i] = v;
    }
    float* rows_out = reinterpret_cast<float*>(out + o_rows);
    const size_t have = a.rows * (size_t)a.nperseg;
    for (size_t i = gid; i < row_floats + (row_floats & 1); i += stride) rows_out[i] = i < have ? a.d_psd[i] : 0.f;
}

int launch_pack_part(gj_ctx* ctx, const gj_part_pack& a, double* d_out) {
    const size_t n = GJ_RESULT_HEADER + a.chunk_cap + 2 * a.tile_cap + (size_t)GJ_RESULT_PAIR_FIELDS * a.pair_cap +
                     a.rows_cap * (size_t)a.nperseg;
    size_t blocks = (n + 255) / 256;
    if (blocks > 1024) blocks = 1024;
    hipLaunchKernelGGL(pack_part_kernel, dim3((unsigned)blocks), dim3(256), 0, ctx->stream, a, d_out);
    GJ_LAUNCH_CHECK(ctx);
    return GJ_OK;
}

__global__ void onset_combine_kernel(const gj_onset* __restrict__ parts, int n, gj_onset* __restrict__ out) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    onset_combine(parts, n, out);
}

int launch_onset_combine(gj_ctx* ctx, const gj_onset* d_parts, int n_parts, gj_onset* d_out) {
    if (n_parts < 1) return fail(ctx, GJ_ERR_INVALID, "n_parts must be >= 1");
    hipLaunchKernelGGL(onset_combine_kernel, dim3(1), dim3(64), 0, ctx->stream, d_parts, n_parts, d_out);
    GJ_LAUNCH_CHECK(ctx);
    return GJ_OK;
}

// ---------------------------------------------------------------------------------------
// Rank 0 of a split run (gj_split_combine_dev): every capture rebuilt from the gathered part vectors and finished in
// THREE launches whatever the number of antennas -- assemble (all copies of all captures), statistics (threshold |
// amplitude totals + onset, one workgroup each per capture: k_scan.hip), pack (one grid row per capture).  The copy
// list and the capture descriptors are static for a deployment: validated on the host and uploaded once
// (gj_combine_plan_create).
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void combine_assemble_kernel(const unsigned char* __restrict__ rows,
                                                               const gj_combine_copy* __restrict__ copies) {
    const gj_combine_copy c = copies[blockIdx.y];
    const unsigned char* src = rows + c.src_byte;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < c.count; i += stride) {
        const unsigned char* p = src + i * (size_t)c.src_stride;
        switch (c.kind) {
            case GJ_COPY_F64_F32: reinterpret_cast<float*>(c.dst)[i] = (float)*reinterpret_cast<const double*>(p); break;
            case GJ_COPY_F64: reinterpret_cast<double*>(c.dst)[i] = *reinterpret_cast<const double*>(p); break;
            case GJ_COPY_F32: reinterpret_cast<float*>(c.dst)[i] = *reinterpret_cast<const float*>(p); break;
            default: reinterpret_cast<int*>(c.dst)[i] = (int)*reinterpret_cast<const double*>(p);   // GJ_COPY_F64_I32
        }
    }
}

__global__ __launch_bounds__(1024) void pack_result_batch_kernel(const gj_combine_capture* __restrict__ caps, int nperseg,
                                                                 const int* __restrict__ pairs, const int* __restrict__ lags,
                                                                 const float* __restrict__ peaks,
                                                                 const float* __restrict__ margins) {
    __shared__ float part[16][64];
    const gj_combine_capture c = caps[blockIdx.y];
    pack_result_body((size_t)c.n_chunks, c.d_power, c.d_stats, c.d_amp, c.d_onset, c.d_psd, (size_t)c.rows, nperseg, c.antenna,
                     c.n_pairs, c.pair_cap, pairs, lags, peaks, margins, c.d_out, part);
}

// the same with the descriptors in the kernel arguments (gj_pack_results_dev: up to GJ_MAX_ANTENNAS captures of one
// deployment in ONE launch, no device-side descriptor array to keep)
struct PackBatch {
    gj_combine_capture c[GJ_MAX_ANTENNAS];
};
__global__ __launch_bounds__(1024) void pack_result_multi_kernel(PackBatch B, int nperseg, const int* __restrict__ pairs,
                                                                 const int* __restrict__ lags, const float* __restrict__ peaks,
                                                                 const float* __restrict__ margins) {
    __shared__ float part[16][64];
    const gj_combine_capture& c = B.c[blockIdx.y];
    pack_result_body((size_t)c.n_chunks, c.d_power, c.d_stats, c.d_amp, c.d_onset, c.d_psd, (size_t)c.rows, nperseg, c.antenna,
                     c.n_pairs, c.pair_cap, pairs, lags, peaks, margins, c.d_out, part);
}

int launch_pack_results(gj_ctx* ctx, const gj_combine_capture* caps, int n_caps, int nperseg, const int32_t* d_pairs,
                        const int32_t* d_lags, const float* d_peaks, const float* d_margins) {
    if (n_caps < 1 || n_caps > GJ_MAX_ANTENNAS) return fail(ctx, GJ_ERR_INVALID, "n_captures must be 1..%d", GJ_MAX_ANTENNAS);
    PackBatch B;
    memset(&B, 0, sizeof(B));
    size_t max_chunks = 0;
    int max_pair_cap = 0;
    for (int a = 0; a < n_caps; ++a) {
        const gj_combine_capture& c = caps[a];
        if (!c.d_power || !c.d_stats || !c.d_amp || !c.d_onset || !c.d_psd || !c.d_out)
            return fail(ctx, GJ_ERR_INVALID, "capture %d: null buffer", a);
        if (c.n_pairs < 0 || c.pair_cap < 0 || c.n_pairs > c.pair_cap) return fail(ctx, GJ_ERR_INVALID, "capture %d: %d pairs, capacity %d", a, c.n_pairs, c.pair_cap);
        if (c.n_pairs && (!d_pairs || !d_lags || !d_peaks || !d_margins)) return fail(ctx, GJ_ERR_INVALID, "null pair buffer");
        B.c[a] = c;
        if (c.n_chunks > max_chunks) max_chunks = (size_t)c.n_chunks;
        if (c.pair_cap > max_pair_cap) max_pair_cap = c.pair_cap;
    }
    const unsigned spec_blocks = (unsigned)((nperseg + 63) / 64);
    size_t copy_blocks = (GJ_RESULT_HEADER + max_chunks + (size_t)GJ_RESULT_PAIR_FIELDS * max_pair_cap + 1023) / 1024;
    if (copy_blocks > 256) copy_blocks = 256;
    hipLaunchKernelGGL(pack_result_multi_kernel, dim3(spec_blocks + (unsigned)copy_blocks, (unsigned)n_caps), dim3(1024), 0, ctx->stream,
                       B, nperseg, d_pairs, d_lags, d_peaks, d_margins);
    GJ_LAUNCH_CHECK(ctx);
    return GJ_OK;
}

}   // namespace gj

struct gj_combine_plan {
    gj_combine_copy* d_copies = nullptr;
    gj_combine_capture* d_caps = nullptr;
    int n_copies = 0, n_caps = 0, nperseg = 0;
    size_t max_count = 0, max_chunks = 0, rows_bytes = 0;
    int max_pair_cap = 0;
    float pct = 5.f, rise_db = 6.f;
    const int32_t *d_pairs = nullptr, *d_lags = nullptr;
    const float *d_peaks = nullptr, *d_margins = nullptr;
};

namespace gj {

static size_t copy_dst_bytes(uint32_t kind) { return kind == GJ_COPY_F64 ? 8 : 4; }
static size_t copy_src_bytes(uint32_t kind) { return kind == GJ_COPY_F32 ? 4 : 8; }

// Everything the three kernels will index is checked HERE, once, on the host: a copy that reads outside the gathered
// vectors or writes outside the arena, or a descriptor whose arrays do not fit, never reaches the GPU.  Every bound is
// taken by DIVISION (count <= room / element): the fields are caller-supplied 64-bit numbers, and a product such as
// count * stride wraps for count = 2^61 and would pass a comparison of sums.
int combine_plan_check(gj_ctx* ctx, const gj_combine_copy* copies, int n_copies, const gj_combine_capture* caps, int n_caps,
                       size_t rows_bytes, const void* d_arena, size_t arena_bytes, int nperseg, bool have_pairs, size_t* max_count_out,
                       size_t* max_chunks_out, int* max_pair_cap_out) {
    const uintptr_t a0 = reinterpret_cast<uintptr_t>(d_arena);
    if (arena_bytes > UINTPTR_MAX - a0) return fail(ctx, GJ_ERR_INVALID, "the arena wraps the address space");
    // `count` elements of `elem` bytes at p lie inside the arena
    auto inside = [&](const void* p, unsigned long long count, size_t elem, size_t align) {
        const uintptr_t u = reinterpret_cast<uintptr_t>(p);
        if (!p || u % align != 0 || u < a0 || u - a0 > arena_bytes) return false;
        return count <= (arena_bytes - (u - a0)) / elem;
    };
    size_t max_count = 0;
    for (int k = 0; k < n_copies; ++k) {
        const gj_combine_copy& c = copies[k];
        if (c.kind > GJ_COPY_F64_I32) return fail(ctx, GJ_ERR_INVALID, "copy %d: kind %u", k, c.kind);
        if (c.count == 0) continue;
        const size_t se = copy_src_bytes(c.kind), de = copy_dst_bytes(c.kind);
        if (c.src_stride < se || c.src_byte % se || c.src_stride % se)
            return fail(ctx, GJ_ERR_INVALID, "copy %d: source offset / stride not aligned to its element", k);
        // last element ends at src_byte + (count - 1) * stride + se <= rows_bytes
        if (rows_bytes < se || c.src_byte > rows_bytes - se || c.count - 1 > (rows_bytes - se - c.src_byte) / c.src_stride)
            return fail(ctx, GJ_ERR_INVALID, "copy %d reads %llu elements from byte %llu, stride %u: beyond the %zu bytes gathered", k,
                        (unsigned long long)c.count, (unsigned long long)c.src_byte, c.src_stride, rows_bytes);
        if (!inside(reinterpret_cast<const void*>(c.dst), c.count, de, de))
            return fail(ctx, GJ_ERR_INVALID, "copy %d writes outside the arena", k);
        if (c.count > max_count) max_count = (size_t)c.count;
    }
    size_t max_chunks = 0;
    int max_pair_cap = 0;
    for (int a = 0; a < n_caps; ++a) {
        const gj_combine_capture& c = caps[a];
        if (c.n_chunks == 0 || c.n_parts < 1 || c.n_pairs < 0 || c.pair_cap < 0 || c.n_pairs > c.pair_cap)
            return fail(ctx, GJ_ERR_INVALID, "capture %d: empty power map, no parts or more pairs than capacity", a);
        if (c.n_tiles != amp_tile_count(c.total_bytes))
            return fail(ctx, GJ_ERR_INVALID, "capture %d of %llu bytes has %zu tiles, not %llu", a, (unsigned long long)c.total_bytes,
                        amp_tile_count(c.total_bytes), (unsigned long long)c.n_tiles);
        // result vector: 40 + n_chunks + nperseg + 5 pair_cap doubles; the sum cannot wrap once each term is below 2^60
        if (c.n_chunks > (1ull << 60) || c.rows > (1ull << 60))
            return fail(ctx, GJ_ERR_INVALID, "capture %d: %llu chunks, %llu rows", a, (unsigned long long)c.n_chunks, (unsigned long long)c.rows);
        const unsigned long long out_len = GJ_RESULT_HEADER + c.n_chunks + (unsigned long long)nperseg +
                                           (unsigned long long)GJ_RESULT_PAIR_FIELDS * (unsigned long long)c.pair_cap;
        if (!inside(c.d_power, c.n_chunks, 4, 4) || !inside(c.d_stats, 3, 4, 4) || !inside(c.d_tiles, c.n_tiles, 16, 8) ||
            !inside(c.d_amp_parts, (unsigned long long)c.n_parts, sizeof(gj_amp_part), 8) ||
            !inside(c.d_onset_parts, (unsigned long long)c.n_parts, sizeof(gj_onset), 8) || !inside(c.d_amp, 1, sizeof(gj_amp_stats), 8) ||
            !inside(c.d_onset, 1, sizeof(gj_onset), 8) || !inside(c.d_psd, c.rows ? c.rows : 1, (size_t)nperseg * 4, 4) ||
            !inside(c.d_out, out_len, 8, 8))
            return fail(ctx, GJ_ERR_INVALID, "capture %d: an array lies outside the arena", a);
        if (c.n_pairs && !have_pairs) return fail(ctx, GJ_ERR_INVALID, "null pair buffer");
        if (c.n_chunks > max_chunks) max_chunks = (size_t)c.n_chunks;
        if (c.pair_cap > max_pair_cap) max_pair_cap = c.pair_cap;
    }
    if (max_count_out) *max_count_out = max_count;
    if (max_chunks_out) *max_chunks_out = max_chunks;
    if (max_pair_cap_out) *max_pair_cap_out = max_pair_cap;
    return GJ_OK;
}

int combine_plan_create(gj_ctx* ctx, const gj_combine_copy* copies, int n_copies, const gj_combine_capture* caps, int n_caps,
                        size_t rows_bytes, const void* d_arena, size_t arena_bytes, int nperseg, float pct, float rise_db,
                        const int32_t* d_pairs, const int32_t* d_lags, const float* d_peaks, const float* d_margins,
                        gj_combine_plan** out) {
    size_t max_count = 0, max_chunks = 0;
    int max_pair_cap = 0;
    const int rc = combine_plan_check(ctx, copies, n_copies, caps, n_caps, rows_bytes, d_arena, arena_bytes, nperseg,
                                      d_pairs && d_lags && d_peaks && d_margins, &max_count, &max_chunks, &max_pair_cap);
    if (rc) return rc;
    gj_combine_plan* p = new (std::nothrow) gj_combine_plan();
    if (!p) return GJ_ERR_NOMEM;
    p->n_copies = n_copies; p->n_caps = n_caps; p->nperseg = nperseg; p->max_count = max_count; p->max_chunks = max_chunks;
    p->max_pair_cap = max_pair_cap; p->rows_bytes = rows_bytes; p->pct = pct; p->rise_db = rise_db;
    p->d_pairs = d_pairs; p->d_lags = d_lags; p->d_peaks = d_peaks; p->d_margins = d_margins;
    if (hipMalloc(&p->d_copies, (size_t)n_copies * sizeof(gj_combine_copy)) != hipSuccess ||
        hipMalloc(&p->d_caps, (size_t)n_caps * sizeof(gj_combine_capture)) != hipSuccess ||
        hipMemcpy(p->d_copies, copies, (size_t)n_copies * sizeof(gj_combine_copy), hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(p->d_caps, caps, (size_t)n_caps * sizeof(gj_combine_capture), hipMemcpyHostToDevice) != hipSuccess) {
        if (p->d_copies) (void)hipFree(p->d_copies);
        if (p->d_caps) (void)hipFree(p->d_caps);
        delete p;
        return fail(ctx, GJ_ERR_NOMEM, "combine plan");
    }
    *out = p;
    return GJ_OK;
}

void combine_plan_destroy(gj_combine_plan* p) {
    if (!p) return;
    (void)hipFree(p->d_copies);
    (void)hipFree(p->d_caps);
    delete p;
}

int launch_split_combine(gj_ctx* ctx, const gj_combine_plan* p, const double* d_rows) {
    // 1. every copy of every capture: grid row = one copy, 256-thread workgroups striding over its elements
    size_t bx = (p->max_count + 255) / 256;
    if (bx > 64) bx = 64;
    if (bx < 1) bx = 1;
    hipLaunchKernelGGL(combine_assemble_kernel, dim3((unsigned)bx, (unsigned)p->n_copies), dim3(256), 0, ctx->stream,
                       reinterpret_cast<const unsigned char*>(d_rows), p->d_copies);
    GJ_LAUNCH_CHECK(ctx);
    // 2. threshold | amplitude totals + onset, two workgroups per capture
    int rc = launch_combine_stats(ctx, p->d_caps, p->n_caps, p->pct, p->rise_db);
    if (rc) return rc;
    // 3. one result vector per capture
    const unsigned spec_blocks = (unsigned)((p->nperseg + 63) / 64);
    size_t copy_blocks = (GJ_RESULT_HEADER + p->max_chunks + (size_t)GJ_RESULT_PAIR_FIELDS * p->max_pair_cap + 1023) / 1024;
    if (copy_blocks > 256) copy_blocks = 256;
    hipLaunchKernelGGL(pack_result_batch_kernel, dim3(spec_blocks + (unsigned)copy_blocks, (unsigned)p->n_caps), dim3(1024), 0,
                       ctx->stream, p->d_caps, p->nperseg, p->d_pairs, p->d_lags, p->d_peaks, p->d_margins);
    GJ_LAUNCH_CHECK(ctx);
    return GJ_OK;
}

int launch_synth(gj_ctx* ctx, const gj_synth_params& p, int64_t first_sample, size_t n_samples, uint8_t* d_out) {
    if (n_samples == 0) return GJ_OK;
    size_t blocks = ((n_samples + 7) / 8 + 255) / 256;
    if (blocks > (size_t)ctx->num_cus * 32) blocks = (size_t)ctx->num_cus * 32;
    hipLaunchKernelGGL(synth_kernel, dim3((unsigned)blocks), dim3(256), 0, ctx->stream, p, (long long)first_sample,
                       n_samples, d_out);
    GJ_LAUNCH_CHECK(ctx);
    return GJ_OK;
}

}   // namespace gj
