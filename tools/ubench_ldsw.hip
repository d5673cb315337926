// Do LDS stores take vector-issue time on their SIMD?  (gfx950)
// Every wave runs `iters` times: 224 v_pk_fma_f32 (two radix-16 passes' worth) with LDS instructions of one kind
// interleaved at regular distance, no barrier, one s_waitcnt at the end of the iteration.  Three 256-thread
// workgroups per CU (48 KiB of LDS each), as K2.  Printed: cycles per iteration per SIMD (three waves).
//   hipcc -O3 --offload-arch=gfx950 tools/ubench_ldsw.hip -o tools/ubench_ldsw && tools/ubench_ldsw
#include <hip/hip_runtime.h>

#include <cstdio>
#include <utility>
typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));

#define PK7                                                                                                      \
    asm volatile("v_pk_fma_f32 %0, %0, %7, %8\n v_pk_fma_f32 %1, %1, %7, %8\n v_pk_fma_f32 %2, %2, %7, %8\n"      \
                 "v_pk_fma_f32 %3, %3, %7, %8\n v_pk_fma_f32 %4, %4, %7, %8\n v_pk_fma_f32 %5, %5, %7, %8\n"      \
                 "v_pk_fma_f32 %6, %6, %7, %8\n"                                                                  \
                 : "+v"(p[0]), "+v"(p[1]), "+v"(p[2]), "+v"(p[3]), "+v"(p[4]), "+v"(p[5]), "+v"(p[6])             \
                 : "v"(pb), "v"(pc));

// MODE: 0 none, 1 ds_write_b64, 2 2 x ds_write_b32, 3 2 x ds_write_addtid_b32, 4 ds_read_b64, 5 ds_read_b128 (every 4th slot),
//       6 ds_write_b128 (every 2nd slot)
#define SLOT(S)                                                                                                        \
    {                                                                                                                  \
        if (VALU) { PK7 }                                                                                              \
        if (MODE == 1) asm volatile("ds_write_b64 %0, %1 offset:%2" ::"v"(a8), "v"(d), "n"(((S) & 15) * 2048) : "memory"); \
        if (MODE == 2) {                                                                                               \
            asm volatile("ds_write_b32 %0, %1 offset:%2" ::"v"(a8), "v"(d.x), "n"(((S) & 15) * 2048) : "memory");      \
            asm volatile("ds_write_b32 %0, %1 offset:%2" ::"v"(a8), "v"(d.y), "n"(((S) & 15) * 2048 + 4) : "memory");  \
        }                                                                                                              \
        if (MODE == 3) {                                                                                               \
            asm volatile("ds_write_addtid_b32 %0 offset:%1" ::"v"(d.x), "n"(((S) & 15) * 2048) : "memory");            \
            asm volatile("ds_write_addtid_b32 %0 offset:%1" ::"v"(d.y), "n"(((S) & 15) * 2048 + 1024) : "memory");     \
        }                                                                                                              \
        if (MODE == 4) asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(d) : "v"(a8), "n"(((S) & 15) * 2048) : "memory"); \
        if (MODE == 5 && ((S) & 3) == 0)                                                                               \
            asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(q) : "v"(a16), "n"(((S) & 7) * 4096) : "memory");      \
        if (MODE == 6 && ((S) & 1) == 0)                                                                               \
            asm volatile("ds_write_b128 %0, %1 offset:%2" ::"v"(a16), "v"(q), "n"(((S) & 7) * 4096) : "memory");       \
    }
#define SLOT4(S) SLOT(S) SLOT((S) + 1) SLOT((S) + 2) SLOT((S) + 3)
#define SLOT16(S) SLOT4(S) SLOT4((S) + 4) SLOT4((S) + 8) SLOT4((S) + 12)

template <int MODE, bool VALU>
__global__ __launch_bounds__(256) void k(float* out, int iters) {
    __shared__ float2 lds[6144];   // 48 KiB: three workgroups per CU
    const int j = threadIdx.x;
    f2 p[7];
    for (int i = 0; i < 7; ++i) p[i] = f2{(float)(j + i), 1.f};
    const f2 pb = {1.0001f, 0.9999f}, pc = {0.5f, 0.25f};
    f2 d = {(float)j, 2.f};
    f4 q = {0.f, 0.f, 0.f, 0.f};
    const unsigned a8 = (unsigned)(size_t)(lds) + 8u * j;        // conflict-free: consecutive lanes, consecutive 8 B
    const unsigned a16 = (unsigned)(size_t)(lds) + 16u * j;
    const unsigned m0v = (unsigned)(size_t)(lds) + 256u * (j >> 6);
    asm volatile("s_mov_b32 m0, %0" ::"s"(__builtin_amdgcn_readfirstlane(m0v)));
    for (int i = 0; i < iters; ++i) {
        SLOT16(0) SLOT16(16)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    float r = d.x + d.y + q.x + q.y + q.z + q.w;
    for (int i = 0; i < 7; ++i) r += p[i].x + p[i].y;
    if (r == 12345.678f) out[0] = r;
}

template <int MODE, bool VALU>
static void run(const char* name, int cus, float* out) {
    const int iters = 2000, wgs = cus * 3;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    hipLaunchKernelGGL((k<MODE, VALU>), dim3(wgs), dim3(256), 0, 0, out, iters);
    (void)hipDeviceSynchronize();
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL((k<MODE, VALU>), dim3(wgs), dim3(256), 0, 0, out, iters);
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        float ms = 0;
        (void)hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    printf("%-44s %8.3f ms  %8.1f cycles per iteration per SIMD (3 waves) @2.4 GHz\n", name, best, best * 1e-3 * 2.4e9 / iters);
}

int main() {
    hipDeviceProp_t prop;
    (void)hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount;
    float* out;
    (void)hipMalloc(&out, 4);
    printf("per iteration and wave: 224 v_pk_fma_f32 and/or the LDS instructions named; 3 workgroups per CU\n");
    run<0, true>("VALU only", cus, out);
    run<1, false>("32 ds_write_b64 only", cus, out);
    run<1, true>("VALU + 32 ds_write_b64", cus, out);
    run<2, false>("64 ds_write_b32 only", cus, out);
    run<2, true>("VALU + 64 ds_write_b32", cus, out);
    run<3, false>("64 ds_write_addtid_b32 only", cus, out);
    run<3, true>("VALU + 64 ds_write_addtid_b32", cus, out);
    run<4, false>("32 ds_read_b64 only", cus, out);
    run<4, true>("VALU + 32 ds_read_b64", cus, out);
    run<5, false>("8 ds_read_b128 only", cus, out);
    run<5, true>("VALU + 8 ds_read_b128", cus, out);
    run<6, false>("16 ds_write_b128 only", cus, out);
    run<6, true>("VALU + 16 ds_write_b128", cus, out);
    return 0;
}
