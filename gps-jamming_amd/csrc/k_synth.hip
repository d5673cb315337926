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
