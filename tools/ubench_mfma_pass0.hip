// Micro-benchmark that settles VERDICT r01 "weak 3": would the matrix pipe, idle in K2, pay for itself if the
// FIRST radix-16 pass of the 4096-point Welch transform ran on it?
//
// Pass 0 is the only pass whose inputs are exact small integers (2u - 255, nine bits): they are exact in fp16, so
// a DFT-as-GEMM needs a split (hi + lo fp16) only of the constant W16 matrix: per 32 columns
// D[32 x 32] = (A_hi + A_lo)[32 x 32] . B[32 x 32], K = 32 = 16 samples x (re, im), i.e. four
// v_mfma_f32_32x32x16_f16 -- IF the Hann window is not applied before the transform (w (2u - 255) is not an fp16
// integer) but as a three-term fix-up after it, which costs VALU work again (the fix-up couples neighbouring
// outputs of a column that the MFMA result layout spreads over two lanes).
//
// Both variants below do the whole arithmetic of one K2 wave-step in registers (no loads, no LDS: the regime that
// is most favourable to the matrix pipe), at K2's occupancy (256-thread workgroups, three per CU):
//   VALU  : 32 byte->float converts, window (16 packed add + 16 packed mul), pass 0 as dft16_fma (fft_core.h),
//           then passes 1 and 2 (dft16_fma_tw with register twiddles) and the |X|^2 accumulation -- K2's own code;
//   MFMA<W>: 16 v_perm_b32 + 16 v_pk_fma_f16 (bytes -> fp16 integers), 8 MFMAs (two 32-column groups x two
//           k-steps x hi/lo), W packed-f32 FMAs standing in for the window fix-up (W = 0: no window at all = the
//           upper bound of the gain), then the same passes 1, 2 and accumulation.
// The MFMA variant's numbers are not a DFT (the operand values are arbitrary): only its timing matters.
//   hipcc -O3 --offload-arch=gfx950 -fno-slp-vectorize -I gps-jamming_amd/csrc tools/ubench_mfma_pass0.hip -o /tmp/ub && /tmp/ub
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

#include "fft_core.h"

using namespace gj;

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half2v __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int MODE, int W>   // MODE 0: VALU front end, 1: MFMA front end with W stand-in fix-up FMAs
__global__ __launch_bounds__(256, 3) void step_kernel(float* __restrict__ out, const unsigned* __restrict__ seed, int iters) {
    const int tid = threadIdx.x;
    const InnerTw ktw = inner_twiddles();
    c2 tw1[15], tw2[15];
#pragma unroll
    for (int k = 0; k < 15; ++k) {
        tw1[k] = make_c2(__cosf(0.01f * (tid + k)), __sinf(0.01f * (tid + k)));
        tw2[k] = make_c2(__cosf(0.02f * (tid + 3 * k)), __sinf(0.02f * (tid + 3 * k)));
    }
    float w16[16];
#pragma unroll
    for (int s = 0; s < 16; ++s) w16[s] = 0.5f - 0.5f * __cosf(0.0015f * (tid + 256 * s));
    unsigned raw[16];
#pragma unroll
    for (int s = 0; s < 16; ++s) raw[s] = seed[(tid + 37 * s) & 1023];
    float acc[16];
#pragma unroll
    for (int s = 0; s < 16; ++s) acc[s] = 0.f;
    // MFMA operands: A (the split W16 matrix) is loop invariant, 4 x 4 VGPRs
    half8 a_hi0, a_lo0, a_hi1, a_lo1;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const float c = __cosf(0.3926991f * ((tid & 31) * (8 * (tid >> 5) + j)));
        a_hi0[j] = (_Float16)c;
        a_lo0[j] = (_Float16)(c - (float)(_Float16)c);
        a_hi1[j] = (_Float16)(0.5f * c);
        a_lo1[j] = (_Float16)(0.5f * c - (float)(_Float16)(0.5f * c));
    }
    const c2 khalf = make_c2(-127.5f, -127.5f);
    const half2v two = {(_Float16)2.0f, (_Float16)2.0f}, off = {(_Float16)-2303.0f, (_Float16)-2303.0f};   // 2 (1024 + u) - 2303 = 2u - 255
    for (int it = 0; it < iters; ++it) {
        c2 v[16];
#pragma unroll
        for (int s = 0; s < 16; ++s) raw[s] = (raw[s] + 0x00030005u) & 0x00ff00ffu;   // new bytes every step (same in both variants)
        if constexpr (MODE == 0) {
#pragma unroll
            for (int s = 0; s < 16; ++s) {
                const c2 f = make_c2((float)(raw[s] & 255u), (float)(raw[s] >> 16));
                const c2 t = cadd(f, khalf);
                v[s] = make_c2(t.x * w16[s], t.y * w16[s]);
            }
            fft_pass<4096, 0, false, true>(v, tw1, ktw);
        } else {
#pragma unroll
            for (int g = 0; g < 2; ++g) {
                half2v h[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    // (0x6400 | I, 0x6400 | Q) as two fp16 = (1024 + I, 1024 + Q); one v_perm_b32, one v_pk_fma_f16
                    const unsigned sp = __builtin_amdgcn_perm(raw[8 * g + j], 0x64646464u, 0x04060400u);
                    h[j] = __builtin_bit_cast(half2v, sp) * two + off;
                }
                half8 b0, b1;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    b0[2 * j] = h[j][0]; b0[2 * j + 1] = h[j][1];
                    b1[2 * j] = h[4 + j][0]; b1[2 * j + 1] = h[4 + j][1];
                }
                f32x16 d = {0.f};
                d = __builtin_amdgcn_mfma_f32_32x32x16_f16(a_hi0, b0, d, 0, 0, 0);
                d = __builtin_amdgcn_mfma_f32_32x32x16_f16(a_lo0, b0, d, 0, 0, 0);
                d = __builtin_amdgcn_mfma_f32_32x32x16_f16(a_hi1, b1, d, 0, 0, 0);
                d = __builtin_amdgcn_mfma_f32_32x32x16_f16(a_lo1, b1, d, 0, 0, 0);
#pragma unroll
                for (int j = 0; j < 8; ++j) v[8 * g + j] = make_c2(d[2 * j], d[2 * j + 1]);
            }
            // stand-in for the window fix-up: W dependent-on-neighbour packed FMAs
#pragma unroll
            for (int k = 0; k < W; ++k) v[k & 15] = fma2(v[(k + 1) & 15], tw1[k % 15], v[k & 15]);
        }
        fft_pass<4096, 1, false, true>(v, tw1, ktw);
        fft_pass<4096, 2, false, true>(v, tw2, ktw);
#pragma unroll
        for (int s = 0; s < 16; ++s) acc[s] = fmaf(v[s].x, v[s].x, fmaf(v[s].y, v[s].y, acc[s]));
    }
    float r = 0.f;
#pragma unroll
    for (int s = 0; s < 16; ++s) r += acc[s];
    out[blockIdx.x * 256 + tid] = r;
}

template <int MODE, int W>
static double run(const char* name, float* d_out, unsigned* d_seed, int iters, double base_ms) {
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    const int blocks = 256 * 3;
    for (int k = 0; k < 3; ++k) hipLaunchKernelGGL((step_kernel<MODE, W>), dim3(blocks), dim3(256), 0, 0, d_out, d_seed, iters);
    hipDeviceSynchronize();
    float best = 1e30f;
    for (int k = 0; k < 7; ++k) {
        hipEventRecord(a);
        hipLaunchKernelGGL((step_kernel<MODE, W>), dim3(blocks), dim3(256), 0, 0, d_out, d_seed, iters);
        hipEventRecord(b);
        hipEventSynchronize(b);
        float ms;
        hipEventElapsedTime(&ms, a, b);
        best = ms < best ? ms : best;
    }
    // one wave-step = one iteration of one wave; three waves share a SIMD
    const double ns_per_step_simd = best * 1e6 / iters / 3.0;
    printf("%-34s %8.3f ms   %7.1f ns per wave-step and SIMD   %s%+.1f %%\n", name, best, ns_per_step_simd,
           base_ms > 0 ? "vs VALU " : "", base_ms > 0 ? (best / base_ms - 1.0) * 100.0 : 0.0);
    return best;
}

int main() {
    float* d_out;
    unsigned* d_seed;
    hipMalloc(&d_out, 256 * 3 * 256 * sizeof(float));
    hipMalloc(&d_seed, 1024 * sizeof(unsigned));
    std::vector<unsigned> seed(1024);
    for (int i = 0; i < 1024; ++i) seed[i] = (unsigned)(i * 2654435761u) & 0x00ff00ffu;
    hipMemcpy(d_seed, seed.data(), 4096, hipMemcpyHostToDevice);
    const int iters = 4000;
    printf("K2 wave-step arithmetic in registers, 768 workgroups x 256 threads (3 per CU), %d steps\n", iters);
    const double base = run<0, 0>("VALU pass 0 + window (K2 today)", d_out, d_seed, iters, 0.0);
    run<1, 0>("MFMA pass 0, no window (bound)", d_out, d_seed, iters, base);
    run<1, 32>("MFMA pass 0 + 32 fix-up FMAs", d_out, d_seed, iters, base);
    run<1, 64>("MFMA pass 0 + 64 fix-up FMAs", d_out, d_seed, iters, base);
    run<1, 96>("MFMA pass 0 + 96 fix-up FMAs", d_out, d_seed, iters, base);
    return 0;
}
