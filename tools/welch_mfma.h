// Pass 0 of the 4096-point Welch transform on the matrix pipe (gfx950), building block of welch_mfma_kernel.
//
// The sixteen inputs of a pass-0 butterfly are raw samples: u - 128 is an exact small integer in fp16, so the DFT over
// n2 (x[n1 + 256 n2], n2 = 0..15) is a GEMM whose only inexact operand is the constant W16 matrix, split hi + lo in
// fp16 (22 significant bits):  D[32 x 32] = (A_hi + A_lo)[32 x 32] . B[32 x 32],  K = 16 samples x (re, im),
// rows = 16 outputs x (re, im), columns = 32 butterflies -- four v_mfma_f32_32x32x16_f16 per 32 butterflies, eight per
// wave-step.  Lane l feeds column l % 32 with the samples n2 = 8 (l / 32) + 0..7 and receives the outputs
// k2 = 8 (l / 32) + 0..7 of that column; sixteen v_permlane32_swap then pair the halves up so that thread tid holds
// all sixteen outputs of butterfly n1 = tid, which is what the exchange schedule of fft_core.h (X4096) expects.
#pragma once
#include <hip/hip_runtime.h>

#include "fft_core.h"

#ifndef GJ_MFMA_ABLATE
#define GJ_MFMA_ABLATE 0
#endif

namespace gj {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half2v __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

struct MfmaDft16 {
    half8 hi[2], lo[2];   // [m]: K block m = samples j = 4 m .. 4 m + 3 of the lane's eight, (re, im) interleaved
};

// cos(2 pi j / 16), j = 0..15, rounded from double
__device__ __forceinline__ float cos16(int j) {
    constexpr float t[16] = {1.0f, 0.92387953251128674f, 0.70710678118654752f, 0.38268343236508977f, 0.0f,
                             -0.38268343236508977f, -0.70710678118654752f, -0.92387953251128674f, -1.0f,
                             -0.92387953251128674f, -0.70710678118654752f, -0.38268343236508977f, 0.0f,
                             0.38268343236508977f, 0.70710678118654752f, 0.92387953251128674f};
    return t[j & 15];
}

// The lane's share of the split DFT-16 matrix.  MFMA A layout (32 x 16 per instruction): lane l holds row l % 32,
// k = 8 (l / 32) + 0..7.  Row r = 8 a + 4 H + 2 b + p is output k2 = 8 H + 2 a + b, part p (0 re, 1 im): with this
// order the D registers of lane half H come out as the complex pairs (d[2 q], d[2 q + 1]) = output k2 = 8 H + q.
__device__ __forceinline__ MfmaDft16 mfma_dft16_matrix(int lane) {
    MfmaDft16 A;
    const int r = lane & 31, hl = lane >> 5;
    const int k2 = 8 * ((r >> 2) & 1) + 2 * (r >> 3) + ((r >> 1) & 1), p = r & 1;
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int n2 = 8 * hl + 4 * m + (i >> 1), pin = i & 1;
            const float c = cos16(k2 * n2), s = cos16(k2 * n2 - 4);   // sin x = cos(x - pi/2)
            // (c - i s)(xr + i xi): re = c xr + s xi, im = c xi - s xr
            const float w = (p == 0) ? (pin == 0 ? c : s) : (pin == 0 ? -s : c);
            const _Float16 h = (_Float16)w;
            A.hi[m][i] = h;
            A.lo[m][i] = (_Float16)(w - (float)h);
        }
    return A;
}

// raw = I | Q << 8 (a 16-bit load) -> (I - 128, Q - 128) as two fp16, exact: 0x6400 | u is the fp16 1024 + u
__device__ __forceinline__ half2v unpack_f16(unsigned raw) {
    const unsigned sp = __builtin_amdgcn_perm(raw, 0x64646464u, 0x00050004u);   // bytes (I, 0x64, Q, 0x64)
    const half2v k = {(_Float16)-1152.0f, (_Float16)-1152.0f};
    return __builtin_bit_cast(half2v, sp) + k;
}

// x[g][j]: the lane's samples of butterfly group g (n1 = 64 wave + 32 g + lane % 32), n2 = 8 (lane / 32) + j.
// On return v[k2] = sum over n2 of x[n1 = 64 wave + lane][n2] W16^(n2 k2), k2 = 0..15.
__device__ __forceinline__ void pass0_mfma(const half2v (&x)[2][8], const MfmaDft16& A, c2 (&v)[16]) {
    f32x16 d[2];
#pragma unroll
    for (int g = 0; g < 2; ++g) {
        half8 b0, b1;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            b0[2 * j] = x[g][j][0];
            b0[2 * j + 1] = x[g][j][1];
            b1[2 * j] = x[g][4 + j][0];
            b1[2 * j + 1] = x[g][4 + j][1];
        }
        f32x16 acc = {0.f};
#if GJ_MFMA_ABLATE == 1   // timing only: no matrix instructions, the operands just kept alive
#pragma unroll
        for (int r = 0; r < 8; ++r) { acc[r] = (float)b0[r] * (float)A.hi[0][r]; acc[8 + r] = (float)b1[r] * (float)A.lo[1][r]; }
#elif GJ_MFMA_ABLATE == 2   // timing only: hi terms only (half the matrix instructions)
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(A.hi[0], b0, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(A.hi[1], b1, acc, 0, 0, 0);
#else
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(A.hi[0], b0, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(A.hi[1], b1, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(A.lo[0], b0, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(A.lo[1], b1, acc, 0, 0, 0);
#endif
        d[g] = acc;
    }
    // lanes 32..63 of d[0] <-> lanes 0..31 of d[1]: every lane then holds k2 = 0..7 in d[0] and k2 = 8..15 in d[1]
    // of ITS butterfly n1 = 64 wave + lane
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(d[0][r]), __float_as_uint(d[1][r]), false, false);
        d[0][r] = __uint_as_float(sw[0]);
        d[1][r] = __uint_as_float(sw[1]);
    }
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        v[q] = make_c2(d[0][2 * q], d[0][2 * q + 1]);
        v[8 + q] = make_c2(d[1][2 * q], d[1][2 * q + 1]);
    }
}

}   // namespace gj
