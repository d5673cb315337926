// Numerical check of csrc/welch_mfma.h on the GPU: pass 0 of a 4096-point segment on the matrix pipe against a
// double-precision DFT-16 of the same bytes on the host.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -I gps-jamming_amd/csrc -I tools tools/mfma_pass0_check.hip -o tools/mfma_pass0_check
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <vector>

#include "welch_mfma.h"   // tools/welch_mfma.h

using namespace gj;

__global__ __launch_bounds__(256) void k(const unsigned char* iq, float2* out) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const MfmaDft16 A = mfma_dft16_matrix(lane);
    half2v x[2][8];
#pragma unroll
    for (int g = 0; g < 2; ++g)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int n1 = 64 * wave + 32 * g + (lane & 31), n2 = 8 * (lane >> 5) + j;
            const unsigned raw = *reinterpret_cast<const unsigned short*>(iq + 2 * (n1 + 256 * n2));
            x[g][j] = unpack_f16(raw);
        }
    c2 v[16];
    pass0_mfma(x, A, v);
#pragma unroll
    for (int k2 = 0; k2 < 16; ++k2) out[tid * 16 + k2] = make_float2(v[k2].x, v[k2].y);
}

int main() {
    std::vector<unsigned char> h(8192);
    unsigned s = 12345;
    for (auto& b : h) { s = s * 1664525u + 1013904223u; b = (unsigned char)(s >> 24); }
    h[0] = 0; h[1] = 255; h[2] = 255; h[3] = 0;
    unsigned char* d_iq;
    float2* d_out;
    hipMalloc(&d_iq, 8192);
    hipMalloc(&d_out, 4096 * sizeof(float2));
    hipMemcpy(d_iq, h.data(), 8192, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(256), 0, 0, d_iq, d_out);
    std::vector<float2> o(4096);
    if (hipMemcpy(o.data(), d_out, 4096 * sizeof(float2), hipMemcpyDeviceToHost) != hipSuccess) { printf("FAIL: hip error\n"); return 1; }
    double worst = 0.0, scale = 0.0;
    int bad = 0;
    for (int n1 = 0; n1 < 256; ++n1)
        for (int k2 = 0; k2 < 16; ++k2) {
            double re = 0, im = 0;
            for (int n2 = 0; n2 < 16; ++n2) {
                const double xr = (double)h[2 * (n1 + 256 * n2)] - 128.0, xi = (double)h[2 * (n1 + 256 * n2) + 1] - 128.0;
                const double a = -2.0 * M_PI * (double)((n2 * k2) & 15) / 16.0;
                re += xr * cos(a) - xi * sin(a);
                im += xr * sin(a) + xi * cos(a);
            }
            const double e = fmax(fabs(o[n1 * 16 + k2].x - re), fabs(o[n1 * 16 + k2].y - im));
            if (e > worst) worst = e;
            scale = fmax(scale, fmax(fabs(re), fabs(im)));
            if (e > 1e-3 && bad < 5) { printf("n1 %d k2 %d: got (%g, %g) want (%g, %g)\n", n1, k2, o[n1 * 16 + k2].x, o[n1 * 16 + k2].y, re, im); ++bad; }
        }
    printf("%s: worst abs error %.3e at magnitudes up to %.1f (relative %.2e)\n", worst < 1e-3 ? "OK" : "FAIL", worst, scale, worst / scale);
    return worst < 1e-3 ? 0 : 1;
}
