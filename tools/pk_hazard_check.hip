// Does gfx950 need a wait state between a v_pk_*_f32 and a packed op that reads its result?
// hipcc pads such pairs with s_nop (its hazard recogniser reads the default op_sel_hi bit of a
// VOP3P source as "writes the high half", and assumes the worst of every inline asm).  This runs
// long chains of back-to-back DEPENDENT packed ops inside ONE asm statement (no padding possible)
// and compares every lane bit for bit with the same arithmetic done on the host.
//   hipcc -O2 --offload-arch=gfx950 tools/pk_hazard_check.hip -o tools/pk_hazard_check
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstring>
#include <vector>

typedef float f2 __attribute__((ext_vector_type(2)));

__global__ void chain(const f2* in, const f2* coef, f2* out, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    f2 x = in[i], a = coef[0], b = coef[1], t;
#pragma unroll 1
    for (int rep = 0; rep < 8; ++rep) {
        asm volatile(
            // complex multiply x*a (dependent pair), then dependent add, fma, rotated add, mul
            "v_pk_mul_f32 %1, %0, %2 op_sel:[0,0] op_sel_hi:[0,1]\n"
            "v_pk_fma_f32 %0, %0, %2, %1 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[0,1,0]\n"
            "v_pk_add_f32 %0, %0, %3\n"
            "v_pk_fma_f32 %0, %0, %2, %3\n"
            "v_pk_add_f32 %0, %0, %0 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]\n"
            "v_pk_mul_f32 %0, %0, %3\n"
            "v_pk_fma_f32 %0, %0, 2.0, %3 neg_lo:[0,0,1] neg_hi:[0,0,1]\n"
            "v_pk_add_f32 %0, %0, %3 neg_lo:[0,1] neg_hi:[0,1]\n"
            : "+v"(x), "=&v"(t)
            : "v"(a), "v"(b));
    }
    out[i] = x;
}

static void host_chain(float& x0, float& x1, const float a[2], const float b[2]) {
    for (int rep = 0; rep < 8; ++rep) {
        float t0 = x0 * a[0], t1 = x0 * a[1];
        float y0 = fmaf(x1, -a[1], t0), y1 = fmaf(x1, a[0], t1);
        y0 += b[0]; y1 += b[1];
        y0 = fmaf(y0, a[0], b[0]); y1 = fmaf(y1, a[1], b[1]);
        float z0 = y0 + y1, z1 = y1 - y0;            // a + (-j) a : (a.x + a.y, a.y - a.x)
        z0 *= b[0]; z1 *= b[1];
        z0 = fmaf(z0, 2.0f, -b[0]); z1 = fmaf(z1, 2.0f, -b[1]);
        z0 -= b[0]; z1 -= b[1];
        x0 = z0; x1 = z1;
    }
}

int main() {
    const int n = 1 << 20;
    std::vector<float> in(2 * n), out(2 * n);
    for (int i = 0; i < 2 * n; ++i) in[i] = (float)((i * 2654435761u >> 12) % 2001) / 1000.0f - 1.0f;
    const float coef[4] = {0.92387953f, -0.38268343f, 0.3125f, -0.171875f};
    f2 *d_in, *d_coef, *d_out;
    if (hipMalloc(&d_in, 8 * n) || hipMalloc(&d_out, 8 * n) || hipMalloc(&d_coef, 16)) return 2;
    (void)hipMemcpy(d_in, in.data(), 8 * n, hipMemcpyHostToDevice);
    (void)hipMemcpy(d_coef, coef, 16, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(chain, dim3(n / 256), dim3(256), 0, 0, d_in, d_coef, d_out, n);
    if (hipMemcpy(out.data(), d_out, 8 * n, hipMemcpyDeviceToHost) != hipSuccess) return 3;
    long bad = 0;
    for (int i = 0; i < n; ++i) {
        float x0 = in[2 * i], x1 = in[2 * i + 1];
        host_chain(x0, x1, coef, coef + 2);
        if (memcmp(&x0, &out[2 * i], 4) || memcmp(&x1, &out[2 * i + 1], 4)) {
            if (bad < 5) printf("lane %d: host (%.9g, %.9g) device (%.9g, %.9g)\n", i, x0, x1, out[2 * i], out[2 * i + 1]);
            ++bad;
        }
    }
    printf("pk_hazard_check: %ld of %d lanes differ -> %s\n", bad, n,
           bad ? "back-to-back dependent packed ops are NOT safe" : "no wait state needed between dependent v_pk_*_f32");
    return bad ? 1 : 0;
}
