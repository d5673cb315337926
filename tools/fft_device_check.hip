// Device check of csrc/fft_core.h: one workgroup, N-point FFT(s) of random data, against a
// double-precision DFT on the host; both twiddle schemes.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <vector>
#include "../gps-jamming_amd/csrc/fft_core.h"
using namespace gj;

template <int N, int PASS, bool TWO>
__device__ void passes(c2 (&v)[16], cf* lds, int base, int jl, const cf* tab) {
    constexpr int NP = fft_npass(N);
    c2 tw[15];
    for (int i = 0; i < 15; ++i) tw[i] = make_c2(1.f, 0.f);
    constexpr bool two = TWO && PASS > 0 && fft_radix(N, PASS) == 16;
    if constexpr (two) load_twiddles6<N, PASS>(tw, tab, jl);
    else if constexpr (PASS > 0) load_twiddles<N, PASS>(tw, tab, jl);
    fft_pass<N, PASS, TWO>(v, tw, inner_twiddles());
    if constexpr (PASS + 1 < NP) {
        lds_scatter<N, PASS>(v, lds, base, jl);
        __syncthreads();
        lds_gather<N>(v, lds, base, jl);
        __syncthreads();
        passes<N, PASS + 1, TWO>(v, lds, base, jl, tab);
    }
}

template <int N, bool TWO>
__global__ __launch_bounds__(256) void fft_kernel(const cf* in, cf* out, const cf* tab) {
    constexpr int TF = N / 16;
    __shared__ cf lds[lds_span(kBlockPoints) + 64];
    const int b = threadIdx.x / TF, jl = threadIdx.x % TF;
    c2 v[16];
    for (int s = 0; s < 16; ++s) v[s] = to_c2(in[b * N + jl + TF * s]);
    passes<N, 0, TWO>(v, lds, b * lds_span(N), jl, tab);
    for (int s = 0; s < 16; ++s) out[b * N + jl + TF * s] = to_cf(v[s]);
}

// the Welch kernel's arrangement: all twiddles loaded up front into tw[3][15], reused in a loop
template <int N, int PASS, bool TWO>
__device__ __forceinline__ void passes2(c2 (&v)[16], cf* lds, int base, int jl, const c2 (&tw)[3][15], const InnerTw& k) {
    constexpr int NP = fft_npass(N);
    fft_pass<N, PASS, TWO>(v, tw[PASS], k);
    if constexpr (PASS + 1 < NP) {
        lds_scatter<N, PASS>(v, lds, base, jl);
        __syncthreads();
        lds_gather<N>(v, lds, base, jl);
        __syncthreads();
        passes2<N, PASS + 1, TWO>(v, lds, base, jl, tw, k);
    }
}
template <int N, bool TWO>
__global__ __launch_bounds__(256, 2) void fft_kernel2(const cf* in, cf* out, const cf* tab, int reps) {
    constexpr int TF = N / 16, NP = fft_npass(N);
    __shared__ cf lds[lds_span(kBlockPoints) + 64];
    const int b = threadIdx.x / TF, jl = threadIdx.x % TF;
    const InnerTw ktw = inner_twiddles();
    c2 tw[3][15];
    for (int p = 0; p < 3; ++p)
        for (int k = 0; k < 15; ++k) tw[p][k] = make_c2(1.f, 0.f);
    if constexpr (NP > 1) {
        if constexpr (TWO && fft_radix(N, 1) == 16) load_twiddles6<N, 1>(tw[1], tab, jl);
        else load_twiddles<N, 1>(tw[1], tab, jl);
    }
    if constexpr (NP > 2) {
        if constexpr (TWO && fft_radix(N, 2) == 16) load_twiddles6<N, 2>(tw[2], tab, jl);
        else load_twiddles<N, 2>(tw[2], tab, jl);
    }
    for (int r = 0; r < reps; ++r) {
        c2 v[16];
        for (int s = 0; s < 16; ++s) v[s] = to_c2(in[b * N + jl + TF * s]);
        passes2<N, 0, TWO>(v, lds, b * lds_span(N), jl, tw, ktw);
        for (int s = 0; s < 16; ++s) out[b * N + jl + TF * s] = to_cf(v[s]);
    }
}

template <int N, bool TWO>
double run(const cf* d_tab, int variant = 0) {
    std::vector<cf> in(4096), out(4096);
    for (auto& x : in) x = cf{(float)(rand() % 511 - 255), (float)(rand() % 511 - 255)};
    cf *d_in, *d_out;
    (void)hipMalloc(&d_in, 4096 * 8); (void)hipMalloc(&d_out, 4096 * 8);
    (void)hipMemcpy(d_in, in.data(), 4096 * 8, hipMemcpyHostToDevice);
    if (variant == 0) hipLaunchKernelGGL((fft_kernel<N, TWO>), dim3(1), dim3(256), 0, 0, d_in, d_out, d_tab);
    else hipLaunchKernelGGL((fft_kernel2<N, TWO>), dim3(1), dim3(256), 0, 0, d_in, d_out, d_tab, 2);
    (void)hipMemcpy(out.data(), d_out, 4096 * 8, hipMemcpyDeviceToHost);
    double worst = 0;
    for (int b = 0; b < 4096 / N; ++b) {
        double norm = 0;
        std::vector<double> re(N), im(N);
        for (int k = 0; k < N; ++k) {
            double sr = 0, si = 0;
            for (int n = 0; n < N; ++n) {
                const double a = -2.0 * M_PI * (double)((long long)k * n % N) / N;
                sr += in[b * N + n].x * cos(a) - in[b * N + n].y * sin(a);
                si += in[b * N + n].x * sin(a) + in[b * N + n].y * cos(a);
            }
            re[k] = sr; im[k] = si; norm += sr * sr + si * si;
        }
        norm = sqrt(norm / N);
        for (int k = 0; k < N; ++k) {
            const double e = hypot(out[b * N + k].x - re[k], out[b * N + k].y - im[k]) / norm;
            if (e > worst) worst = e;
        }
    }
    return worst;
}

int main() {
    std::vector<cf> tab(4096);
    for (int m = 0; m < 4096; ++m) tab[m] = cf{(float)cos(-2.0 * M_PI * m / 4096), (float)sin(-2.0 * M_PI * m / 4096)};
    cf* d_tab;
    (void)hipMalloc(&d_tab, 4096 * 8);
    (void)hipMemcpy(d_tab, tab.data(), 4096 * 8, hipMemcpyHostToDevice);
    printf("N=4096 direct %.3e two-step %.3e\n", run<4096, false>(d_tab), run<4096, true>(d_tab));
    printf("N=1024 direct %.3e two-step %.3e\n", run<1024, false>(d_tab), run<1024, true>(d_tab));
    printf("N= 256 direct %.3e two-step %.3e\n", run<256, false>(d_tab), run<256, true>(d_tab));
    printf("kernel-style N=4096 direct %.3e two-step %.3e\n", run<4096, false>(d_tab, 1), run<4096, true>(d_tab, 1));
    printf("kernel-style N=1024 direct %.3e two-step %.3e\n", run<1024, false>(d_tab, 1), run<1024, true>(d_tab, 1));
    printf("N=  64 direct %.3e\n", run<64, false>(d_tab));
    return 0;
}
