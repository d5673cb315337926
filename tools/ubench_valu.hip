// Micro-benchmark: f32 VALU issue rates on gfx950 that decide how the FFT butterflies should be
// written -- v_fma_f32 / v_add_f32 vs v_pk_fma_f32 / v_pk_add_f32, at 1, 2 and 4 waves per SIMD
// -- plus LDS ds_write_b64/ds_read_b64 exchange throughput.
//   hipcc -O3 --offload-arch=gfx950 tools/ubench_valu.hip -o /tmp/ubench && /tmp/ubench
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

typedef float f2 __attribute__((ext_vector_type(2)));

#define REP16(x) x x x x x x x x x x x x x x x x

template <int MODE>
__global__ __launch_bounds__(256) void valu_kernel(float* out, int iters) {
    float a0 = threadIdx.x, a1 = 1.f, a2 = 2.f, a3 = 3.f, a4 = 4.f, a5 = 5.f, a6 = 6.f, a7 = 7.f;
    f2 p0 = {a0, 1.f}, p1 = {1.f, 2.f}, p2 = {2.f, 3.f}, p3 = {3.f, 4.f}, p4 = {4.f, 1.f}, p5 = {5.f, 1.f},
       p6 = {6.f, 1.f}, p7 = {7.f, 1.f};
    const float b = 1.0001f, c = 0.5f;
    const f2 pb = {1.0001f, 0.9999f}, pc = {0.5f, 0.25f};
    for (int i = 0; i < iters; ++i) {
        if (MODE == 0) {
            REP16(asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n"
                               "v_fma_f32 %3, %3, %8, %9\n v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n"
                               "v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n"
                               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
                               : "v"(b), "v"(c));)
        } else if (MODE == 1) {
            REP16(asm volatile("v_pk_fma_f32 %0, %0, %8, %9\n v_pk_fma_f32 %1, %1, %8, %9\n v_pk_fma_f32 %2, %2, %8, %9\n"
                               "v_pk_fma_f32 %3, %3, %8, %9\n v_pk_fma_f32 %4, %4, %8, %9\n v_pk_fma_f32 %5, %5, %8, %9\n"
                               "v_pk_fma_f32 %6, %6, %8, %9\n v_pk_fma_f32 %7, %7, %8, %9\n"
                               : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7)
                               : "v"(pb), "v"(pc));)
        } else if (MODE == 2) {
            REP16(asm volatile("v_add_f32 %0, %0, %8\n v_add_f32 %1, %1, %8\n v_add_f32 %2, %2, %8\n"
                               "v_add_f32 %3, %3, %8\n v_add_f32 %4, %4, %8\n v_add_f32 %5, %5, %8\n"
                               "v_add_f32 %6, %6, %8\n v_add_f32 %7, %7, %8\n"
                               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
                               : "v"(c));)
        } else {
            REP16(asm volatile("v_pk_add_f32 %0, %0, %8\n v_pk_add_f32 %1, %1, %8\n v_pk_add_f32 %2, %2, %8\n"
                               "v_pk_add_f32 %3, %3, %8\n v_pk_add_f32 %4, %4, %8\n v_pk_add_f32 %5, %5, %8\n"
                               "v_pk_add_f32 %6, %6, %8\n v_pk_add_f32 %7, %7, %8\n"
                               : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7)
                               : "v"(pc));)
        }
    }
    float r = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + p0.x + p0.y + p1.x + p1.y + p2.x + p2.y + p3.x + p3.y + p4.x +
              p4.y + p5.x + p5.y + p6.x + p6.y + p7.x + p7.y;
    if (r == 12345.678f) out[0] = r;
}

// LDS exchange: every thread writes 16 x 8 B (stride pattern of the FFT scatter) and reads 16 x 8 B
__global__ __launch_bounds__(256) void lds_kernel(float* out, int iters, int mode) {
    __shared__ float2 lds[4096 + 256 + 64];
    const int j = threadIdx.x;
    float2 v[16];
    for (int s = 0; s < 16; ++s) v[s] = make_float2(j + s, j - s);
    for (int i = 0; i < iters; ++i) {
        if (mode == 0) {   // padded: 17 j + t  /  j + j/16 + 272 s
            for (int t = 0; t < 16; ++t) lds[17 * j + t] = v[t];
            __syncthreads();
            for (int s = 0; s < 16; ++s) v[s] = lds[j + (j >> 4) + 272 * s];
            __syncthreads();
        } else if (mode == 1) {   // unpadded (conflicting) 16 j + t  /  j + 256 s
            for (int t = 0; t < 16; ++t) lds[16 * j + t] = v[t];
            __syncthreads();
            for (int s = 0; s < 16; ++s) v[s] = lds[j + 256 * s];
            __syncthreads();
        } else {   // xor swizzle
            for (int t = 0; t < 16; ++t) lds[(16 * j + t) ^ (j & 15)] = v[t];
            __syncthreads();
            for (int s = 0; s < 16; ++s) { int q = j + 256 * s; v[s] = lds[q ^ ((q >> 4) & 15)]; }
            __syncthreads();
        }
    }
    float r = 0;
    for (int s = 0; s < 16; ++s) r += v[s].x + v[s].y;
    if (r == 12345.678f) out[0] = r;
}

template <typename F>
static float time_ms(F&& launch) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    launch();
    hipDeviceSynchronize();
    hipEventRecord(e0);
    launch();
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    return ms;
}

int main() {
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount;
    printf("device %s, %d CUs, clock %d kHz\n", prop.name, cus, prop.clockRate);
    float* out;
    hipMalloc(&out, 4);
    const int iters = 2000;
    const char* names[4] = {"v_fma_f32", "v_pk_fma_f32", "v_add_f32", "v_pk_add_f32"};
    for (int mode = 0; mode < 4; ++mode)
        for (int wps = 1; wps <= 4; wps *= 2) {
            const int blocks = cus * wps;   // 256-thread blocks: 1 wave per SIMD each
            float ms = 0;
            if (mode == 0) ms = time_ms([&] { hipLaunchKernelGGL(valu_kernel<0>, dim3(blocks), dim3(256), 0, 0, out, iters); });
            if (mode == 1) ms = time_ms([&] { hipLaunchKernelGGL(valu_kernel<1>, dim3(blocks), dim3(256), 0, 0, out, iters); });
            if (mode == 2) ms = time_ms([&] { hipLaunchKernelGGL(valu_kernel<2>, dim3(blocks), dim3(256), 0, 0, out, iters); });
            if (mode == 3) ms = time_ms([&] { hipLaunchKernelGGL(valu_kernel<3>, dim3(blocks), dim3(256), 0, 0, out, iters); });
            const double winstr = (double)iters * 128.0 * 4.0 * wps * cus;   // wave-instructions
            const double lanes = winstr * 64.0 * ((mode & 1) ? 2.0 : 1.0);
            printf("%-14s waves/SIMD=%d  %8.3f ms  %7.2f Gwave-instr/s  %7.2f T lane-ops/s  (%.2f cyc/wave-instr/SIMD @2.4GHz)\n",
                   names[mode], wps, ms, winstr / ms * 1e-6, lanes / ms * 1e-9,
                   2.4e9 * (ms * 1e-3) / ((double)iters * 128.0 * wps));
        }
    const char* lnames[3] = {"lds padded", "lds unpadded", "lds xor"};
    for (int mode = 0; mode < 3; ++mode)
        for (int wps = 1; wps <= 4; wps *= 2) {
            if (wps == 4 && mode != 0) {}
            const int blocks = cus * wps;
            const int it = 2000;
            float ms = time_ms([&] { hipLaunchKernelGGL(lds_kernel, dim3(blocks), dim3(256), 0, 0, out, it, mode); });
            const double bytes = (double)it * 4096.0 * 8.0 * 2.0 * blocks;
            printf("%-14s blocks/CU=%d  %8.3f ms  %7.2f TB/s LDS (w+r)  %.1f cyc per exchange per block @2.4GHz\n",
                   lnames[mode], wps, ms, bytes / ms * 1e-9, 2.4e9 * ms * 1e-3 / it / wps);
        }
    return 0;
}
