// Do plain and packed f32 vector instructions share one issue path on gfx950, or can a mix run faster than the sum?
// Three waves per SIMD; per iteration and wave: 224 v_pk_fma_f32, 224 v_fma_f32, or 112 + 112 alternating.
//   hipcc -O3 --offload-arch=gfx950 tools/ubench_mix.hip -o tools/ubench_mix && tools/ubench_mix
#include <hip/hip_runtime.h>

#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));

#define PK4                                                                                                          \
    asm volatile("v_pk_fma_f32 %0, %0, %4, %5\n v_pk_fma_f32 %1, %1, %4, %5\n v_pk_fma_f32 %2, %2, %4, %5\n v_pk_fma_f32 %3, %3, %4, %5\n" \
                 : "+v"(p[0]), "+v"(p[1]), "+v"(p[2]), "+v"(p[3]) : "v"(pb), "v"(pc));
#define PL4                                                                                                          \
    asm volatile("v_fma_f32 %0, %0, %4, %5\n v_fma_f32 %1, %1, %4, %5\n v_fma_f32 %2, %2, %4, %5\n v_fma_f32 %3, %3, %4, %5\n" \
                 : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]) : "v"(b), "v"(c));
#define MIX4                                                                                                         \
    asm volatile("v_pk_fma_f32 %0, %0, %4, %5\n v_fma_f32 %2, %2, %6, %7\n v_pk_fma_f32 %1, %1, %4, %5\n v_fma_f32 %3, %3, %6, %7\n" \
                 : "+v"(p[0]), "+v"(p[1]), "+v"(a[0]), "+v"(a[1]) : "v"(pb), "v"(pc), "v"(b), "v"(c));
#define R8(X) X X X X X X X X
#define R56(X) R8(X) R8(X) R8(X) R8(X) R8(X) R8(X) R8(X)

template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, int iters) {
    __shared__ float pad[12288];   // 48 KiB: three workgroups per CU
    f2 p[4];
    float a[4];
    for (int i = 0; i < 4; ++i) { p[i] = f2{(float)(threadIdx.x + i), 1.f}; a[i] = (float)i; }
    const f2 pb = {1.0001f, 0.9999f}, pc = {0.5f, 0.25f};
    const float b = 1.0001f, c = 0.5f;
    for (int i = 0; i < iters; ++i) {
        if (MODE == 0) { R56(PK4) }
        if (MODE == 1) { R56(PL4) }
        if (MODE == 2) { R56(MIX4) }
    }
    float r = 0;
    for (int i = 0; i < 4; ++i) r += p[i].x + p[i].y + a[i];
    if (r == 12345.678f) { out[0] = r; pad[threadIdx.x] = r; }
}

template <int MODE>
static double run(const char* name, int cus, float* out) {
    const int iters = 2000;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    hipLaunchKernelGGL((k<MODE>), dim3(cus * 3), dim3(256), 0, 0, out, iters);
    (void)hipDeviceSynchronize();
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL((k<MODE>), dim3(cus * 3), dim3(256), 0, 0, out, iters);
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        float ms = 0;
        (void)hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    const double cyc = best * 1e-3 * 2.4e9 / iters / 3.0;
    printf("%-40s %8.3f ms  %8.1f cycles per wave-iteration (224 instructions) @2.4 GHz = %.2f per instruction\n", name, best, cyc, cyc / 224.0);
    return cyc;
}

int main() {
    hipDeviceProp_t prop;
    (void)hipGetDeviceProperties(&prop, 0);
    float* out;
    (void)hipMalloc(&out, 4);
    const double pk = run<0>("224 v_pk_fma_f32", prop.multiProcessorCount, out);
    const double pl = run<1>("224 v_fma_f32", prop.multiProcessorCount, out);
    const double mx = run<2>("112 v_pk_fma_f32 + 112 v_fma_f32, alternating", prop.multiProcessorCount, out);
    printf("sum of the halves %.1f, measured mix %.1f (%+.1f %%)\n", 0.5 * (pk + pl), mx, (mx / (0.5 * (pk + pl)) - 1.0) * 100.0);
    return 0;
}
