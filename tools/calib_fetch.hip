// Calibration of rocprofv3's FETCH_SIZE for the access pattern of welch_kernel: a 1-GiB buffer
// read exactly once (a) with 2-byte loads, 64 lanes x 2 B = one 128-B line per wave-instruction
// (K2's pattern) and (b) with 16-byte loads per lane (the pattern MI355X_MICROARCH.md calibrated:
// FETCH_SIZE = bytes / 2).  Run under `rocprofv3 --pmc FETCH_SIZE`; tools/pmc_summarize.py turns
// the two readings into the factor applied to K2's FETCH_SIZE.
//   hipcc -O3 --offload-arch=gfx950 tools/calib_fetch.hip -o tools/calib_fetch
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>

__global__ __launch_bounds__(256) void calib_read_u16(const uint16_t* __restrict__ p, size_t n, unsigned* out) {
    unsigned acc = 0;
    // a workgroup walks a contiguous 64 KiB run, 256 lanes x 2 B per step, 16 loads in flight
    const size_t per_wg = 32768;
    for (size_t base = (size_t)blockIdx.x * per_wg; base < n; base += (size_t)gridDim.x * per_wg) {
        for (size_t i = threadIdx.x; i < per_wg; i += 256 * 16) {
#pragma unroll
            for (int s = 0; s < 16; ++s) acc += p[base + i + 256 * s];
        }
    }
    if (acc == 0x12345678u) out[0] = acc;
}

__global__ __launch_bounds__(256) void calib_read_x4(const uint4* __restrict__ p, size_t n, unsigned* out) {
    unsigned acc = 0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const uint4 q = p[i];
        acc += q.x ^ q.y ^ q.z ^ q.w;
    }
    if (acc == 0x12345678u) out[0] = acc;
}

int main() {
    const size_t bytes = (size_t)1 << 30;
    uint8_t* d = nullptr;
    unsigned* out = nullptr;
    if (hipMalloc(&d, bytes) != hipSuccess || hipMalloc(&out, 4) != hipSuccess) return 1;
    (void)hipMemset(d, 1, bytes);
    for (int rep = 0; rep < 3; ++rep) {
        hipLaunchKernelGGL(calib_read_u16, dim3(16384), dim3(256), 0, 0, (const uint16_t*)d, bytes / 2, out);
        hipLaunchKernelGGL(calib_read_x4, dim3(8192), dim3(256), 0, 0, (const uint4*)d, bytes / 16, out);
    }
    if (hipDeviceSynchronize() != hipSuccess) return 2;
    printf("calib_fetch: 3 x (u16, x4) reads of %zu bytes done\n", bytes);
    return 0;
}
