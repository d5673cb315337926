// Does LDS exchange traffic overlap with packed-f32 VALU work on gfx950?
//   mode 0: VALU only   mode 1: LDS exchange only   mode 2: both, same wave, independent streams
//   mode 3: 512-thread block, waves 0-3 VALU only, waves 4-7 LDS only (2 waves per SIMD)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ void valu_chunk(f2 (&p)[8], f2 pb, f2 pc) {
#pragma unroll
    for (int r = 0; r < 14; ++r)
        asm volatile("v_pk_fma_f32 %0, %0, %8, %9\n v_pk_fma_f32 %1, %1, %8, %9\n v_pk_fma_f32 %2, %2, %8, %9\n"
                     "v_pk_fma_f32 %3, %3, %8, %9\n v_pk_fma_f32 %4, %4, %8, %9\n v_pk_fma_f32 %5, %5, %8, %9\n"
                     "v_pk_fma_f32 %6, %6, %8, %9\n v_pk_fma_f32 %7, %7, %8, %9\n"
                     : "+v"(p[0]), "+v"(p[1]), "+v"(p[2]), "+v"(p[3]), "+v"(p[4]), "+v"(p[5]), "+v"(p[6]), "+v"(p[7])
                     : "v"(pb), "v"(pc));   // 112 packed ops ~ one radix-16 pass
}

__global__ __launch_bounds__(512) void k(float* out, int iters, int mode) {
    __shared__ float2 lds[2][4096 + 256 + 64];
    const int j = threadIdx.x & 255, grp = threadIdx.x >> 8;
    f2 p[8];
    for (int i = 0; i < 8; ++i) p[i] = f2{(float)(j + i), 1.f};
    const f2 pb = {1.0001f, 0.9999f}, pc = {0.5f, 0.25f};
    float2 v[16];
    for (int s = 0; s < 16; ++s) v[s] = make_float2(j + s, j - s);
    const bool do_valu = (mode == 0) || (mode == 2) || (mode == 3 && grp == 0);
    const bool do_lds = (mode == 1) || (mode == 2) || (mode == 3 && grp == 1);
    float2* L = lds[grp];
    for (int i = 0; i < iters; ++i) {
        if (do_lds) {
            for (int t = 0; t < 16; ++t) L[17 * j + t] = v[t];
        }
        if (do_valu) valu_chunk(p, pb, pc);
        __syncthreads();
        if (do_lds) {
            for (int s = 0; s < 16; ++s) v[s] = L[j + (j >> 4) + 272 * s];
        }
        if (do_valu) valu_chunk(p, pb, pc);
        __syncthreads();
    }
    float r = 0;
    for (int s = 0; s < 16; ++s) r += v[s].x + v[s].y;
    for (int i = 0; i < 8; ++i) r += p[i].x + p[i].y;
    if (r == 12345.678f) out[0] = r;
}

int main() {
    hipDeviceProp_t prop;
    (void)hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount;
    float* out;
    (void)hipMalloc(&out, 4);
    const int iters = 1000;
    const char* names[4] = {"valu only", "lds only", "both, same wave", "split waves (512 thr)"};
    for (int mode = 0; mode < 4; ++mode)
        for (int bpc = 1; bpc <= 2; ++bpc) {
            const int threads = (mode == 3) ? 512 : 256;
            hipEvent_t e0, e1;
            (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
            hipLaunchKernelGGL(k, dim3(cus * bpc), dim3(threads), 0, 0, out, iters, mode);
            (void)hipDeviceSynchronize();
            (void)hipEventRecord(e0);
            hipLaunchKernelGGL(k, dim3(cus * bpc), dim3(threads), 0, 0, out, iters, mode);
            (void)hipEventRecord(e1);
            (void)hipEventSynchronize(e1);
            float ms = 0;
            (void)hipEventElapsedTime(&ms, e0, e1);
            printf("%-24s blocks/CU=%d  %8.3f ms  -> %.1f ns per iteration per block-slot\n", names[mode], bpc, ms,
                   ms * 1e6 / iters / bpc);
        }
    return 0;
}
