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
__global__ __launch_bounds__(1024) void pack_result_kernel(size_t n_chunks, const float* __restrict__ power,
                                                           const float* __restrict__ stats,
                                                           const gj_amp_stats* __restrict__ amp,
                                                           const gj_onset* __restrict__ onset, const float* __restrict__ psd,
                                                           size_t rows, int nperseg, int rank, int n_pairs, int pair_cap,
                                                           const int* __restrict__ pairs, const int* __restrict__ lags,
                                                           const float* __restrict__ peaks, const float* __restrict__ margins,
                                                           double* __restrict__ out) {
    __shared__ float part[16][64];
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

// The capture's onset from its parts' (each already in capture coordinates): the part with the smallest start >= 0
// decides index and margin_hit; guard = smallest guard >= 0; margin_before = the smallest reported by the parts up to
// and including that one (all of them when nothing crossed); noise and threshold are the same on every part.
__global__ void onset_combine_kernel(const gj_onset* __restrict__ parts, int n, gj_onset* __restrict__ out) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
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

int launch_onset_combine(gj_ctx* ctx, const gj_onset* d_parts, int n_parts, gj_onset* d_out) {
    if (n_parts < 1) return fail(ctx, GJ_ERR_INVALID, "n_parts must be >= 1");
    hipLaunchKernelGGL(onset_combine_kernel, dim3(1), dim3(64), 0, ctx->stream, d_parts, n_parts, d_out);
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
