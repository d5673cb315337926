// One wave that samples the shader clock: s_memtime counts shader cycles, s_memrealtime a constant 100 MHz
// (MI355X_MICROARCH.md).  A program of its own (a second HIP runtime inside the Python process finds no device), run
// beside the load:   hipcc -O3 --offload-arch=gfx950 tools/clock_probe.hip -o tools/clock_probe
//   python tools/run_kernel.py welch --reps 600 & sleep 0.6; tools/clock_probe welch
#include <hip/hip_runtime.h>

__global__ void probe_kernel(unsigned long long* out, int n, unsigned gap_ticks) {
    if (threadIdx.x != 0) return;
    for (int i = 0; i < n; ++i) {
        const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
        unsigned long long r1 = r0;
        while (r1 - r0 < gap_ticks) {
            __builtin_amdgcn_s_sleep(32);
            r1 = __builtin_amdgcn_s_memrealtime();
        }
        const unsigned long long c1 = __builtin_amdgcn_s_memtime();
        out[2 * i] = c1 - c0;
        out[2 * i + 1] = r1 - r0;
    }
}

extern "C" int clock_probe_run(int n, double* mhz, int gap_us) {
    hipStream_t s;
    if (hipStreamCreateWithFlags(&s, hipStreamNonBlocking) != hipSuccess) return -1;
    unsigned long long* d = nullptr;
    if (hipMalloc(&d, sizeof(unsigned long long) * 2 * n) != hipSuccess) return -2;
    hipLaunchKernelGGL(probe_kernel, dim3(1), dim3(64), 0, s, d, n, (unsigned)(gap_us * 100));
    if (hipGetLastError() != hipSuccess) return -3;
    if (hipStreamSynchronize(s) != hipSuccess) return -4;
    unsigned long long* h = new unsigned long long[2 * n];
    (void)hipMemcpy(h, d, sizeof(unsigned long long) * 2 * n, hipMemcpyDeviceToHost);
    for (int i = 0; i < n; ++i) mhz[i] = h[2 * i + 1] ? 100.0 * (double)h[2 * i] / (double)h[2 * i + 1] : 0.0;
    delete[] h;
    (void)hipFree(d);
    (void)hipStreamDestroy(s);
    return n;
}

#include <algorithm>
#include <cstdio>
#include <vector>
int main(int argc, char** argv) {
    const int n = 1500;
    std::vector<double> mhz(n);
    const int got = clock_probe_run(n, mhz.data(), 20);
    if (got <= 0) { printf("probe failed with code %d\n", got); return 1; }
    std::sort(mhz.begin(), mhz.end());
    printf("%s: %d samples of ~20 us, shader clock median %.0f MHz, 5th-95th percentile %.0f-%.0f MHz\n", argc > 1 ? argv[1] : "?", got,
           mhz[n / 2], mhz[n / 20], mhz[n - n / 20]);
    return 0;
}
