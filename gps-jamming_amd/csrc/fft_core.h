// Block FFT core for gfx950: one 256-thread workgroup transforms 4096 complex points per
// step = (4096/N) independent N-point FFTs, 16 points per thread held in VGPRs, Stockham
// autosort passes of radix 16 (then one 2/4/8 pass when log2 N is not a multiple of 4),
// LDS exchange between passes with one pad slot per 16 points (bank-conflict-free
// ds_write_b64, 2-way ds_read_b64 -- see tools/lds_bank_sim.py).
//
// The arithmetic and the index maps in this header are plain C++ so that
// tests/host_fft_emul.cpp can run the exact same code on the CPU, one "thread" at a time,
// against numpy's FFT.  Nothing here is a CPU fallback for the product: the kernels that
// include this header only exist as gfx950 code objects.
//
// Replaces the pocketfft calls under scipy.signal.welch (skrypty/widmo_plot.py:48) and
// scipy.signal.correlate (skrypty/triangulateTDOA.py:86).
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define GJ_HD __host__ __device__ __forceinline__
#else
#define GJ_HD inline
#endif

namespace gj {

// `cf` is the storage type (global memory, LDS, kernel signatures).  `c2` is the compute
// type: on the GPU a 2 x f32 vector living in an aligned VGPR pair so that every complex
// add / multiply is ONE packed instruction (v_pk_add_f32 / v_pk_mul_f32 / v_pk_fma_f32 with
// op_sel / neg modifiers doing the swaps and sign flips of j-multiplication for free).
// Measured on MI355X (profiles/r01_ubench_valu_lds.txt): a single wave issues one scalar
// VALU op per ~5 cycles but also one PACKED op per ~5 cycles, so the packed form halves the
// issue-bound time of the butterflies; hipcc's own SLP packing of the scalar form is slower
// than either (one v_mov per packed op, 300+ VGPRs).
// On the host (tests/host_fft_emul.cpp) c2 is the plain struct and the same butterflies run
// as scalar arithmetic.
struct alignas(8) cf {
    float x, y;
};

#if defined(__HIP_DEVICE_COMPILE__)
typedef float c2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ c2 to_c2(cf a) { return c2{a.x, a.y}; }
__device__ __forceinline__ cf to_cf(c2 a) { return cf{a.x, a.y}; }
__device__ __forceinline__ c2 make_c2(float x, float y) { return c2{x, y}; }
__device__ __forceinline__ c2 cadd(c2 a, c2 b) { return a + b; }
__device__ __forceinline__ c2 csub(c2 a, c2 b) { return a - b; }
// a * w = (a.x w.x - a.y w.y, a.x w.y + a.y w.x)
__device__ __forceinline__ c2 cmul(c2 a, c2 w) {
    c2 t, r;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel:[0,0] op_sel_hi:[0,1]" : "=v"(t) : "v"(a), "v"(w));
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[0,1,0]"
        : "=v"(r)
        : "v"(a), "v"(w), "v"(t));
    return r;
}
// same with a wave-uniform constant w held in an SGPR pair (the fixed W16 / W8 twiddles inside
// the butterflies): costs no VGPRs
__device__ __forceinline__ c2 cmul_k(c2 a, c2 w) {
    c2 t, r;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel:[0,0] op_sel_hi:[0,1]" : "=v"(t) : "v"(a), "s"(w));
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[0,1,0]"
        : "=v"(r)
        : "v"(a), "s"(w), "v"(t));
    return r;
}
// c + a * w  (two packed FMAs: the twiddle multiply rides on the butterfly's first addition)
__device__ __forceinline__ c2 cmul_add(c2 a, c2 w, c2 c) {
    c2 t, r;
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[0,1,1]" : "=v"(t) : "v"(a), "v"(w), "v"(c));
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[0,1,0]"
        : "=v"(r)
        : "v"(a), "v"(w), "v"(t));
    return r;
}
__device__ __forceinline__ c2 cmul_add_k(c2 a, c2 w, c2 c) {   // w wave-uniform, in SGPRs
    c2 t, r;
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[0,1,1]" : "=v"(t) : "v"(a), "s"(w), "v"(c));
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[0,1,0]"
        : "=v"(r)
        : "v"(a), "s"(w), "v"(t));
    return r;
}
// The two halves of a complex multiply(-add) as separate operations.  gfx950 needs one wait
// state between a packed-f32 op and a packed op that reads its result (hipcc pads with s_nop);
// the butterflies below issue the first halves of two independent products, then the second
// halves, so no dependent pair is adjacent.
//   cmul_1(a, w)        = (a.x w.x,       a.x w.y)
//   cmul_add_1(a, w, c) = (c.x + a.x w.x, c.y + a.x w.y)
//   cmul_2(a, w, t)     = (t.x - a.y w.y, t.y + a.y w.x)
__device__ __forceinline__ c2 cmul_1(c2 a, c2 w) {
    c2 t;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel:[0,0] op_sel_hi:[0,1]" : "=v"(t) : "v"(a), "v"(w));
    return t;
}
__device__ __forceinline__ c2 cmul_add_1(c2 a, c2 w, c2 c) {
    c2 t;
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[0,1,1]" : "=v"(t) : "v"(a), "v"(w), "v"(c));
    return t;
}
__device__ __forceinline__ c2 cmul_2(c2 a, c2 w, c2 t) {
    c2 r;
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[0,1,0]"
        : "=v"(r)
        : "v"(a), "v"(w), "v"(t));
    return r;
}
__device__ __forceinline__ c2 cmul_1_k(c2 a, c2 w) {   // _k: w wave-uniform, in SGPRs
    c2 t;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel:[0,0] op_sel_hi:[0,1]" : "=v"(t) : "v"(a), "s"(w));
    return t;
}
__device__ __forceinline__ c2 cmul_add_1_k(c2 a, c2 w, c2 c) {
    c2 t;
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[0,1,1]" : "=v"(t) : "v"(a), "s"(w), "v"(c));
    return t;
}
__device__ __forceinline__ c2 cmul_2_k(c2 a, c2 w, c2 t) {
    c2 r;
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[0,1,0]"
        : "=v"(r)
        : "v"(a), "s"(w), "v"(t));
    return r;
}
// 2 u - t   (the "other" butterfly output when t = u + w x is already known)
__device__ __forceinline__ c2 twice_minus(c2 u, c2 t) {
    c2 r;
    asm("v_pk_fma_f32 %0, %1, 2.0, %2 neg_lo:[0,0,1] neg_hi:[0,0,1]" : "=v"(r) : "v"(u), "v"(t));
    return r;
}
// a + (-j) b = (a.x + b.y, a.y - b.x)      a + (+j) b = (a.x - b.y, a.y + b.x)
__device__ __forceinline__ c2 add_mj(c2 a, c2 b) {
    c2 r;
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ c2 add_pj(c2 a, c2 b) {
    c2 r;
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
// a * (-j) = (a.y, -a.x)
__device__ __forceinline__ c2 mul_mj(c2 a) {
    c2 r;
    asm("v_pk_mul_f32 %0, %1, 1.0 op_sel:[1,0] op_sel_hi:[0,0] neg_hi:[1,0]" : "=v"(r) : "v"(a));
    return r;
}
// acc += (a.x^2, a.y^2)
__device__ __forceinline__ void acc_sq(c2& acc, c2 a) { asm("v_pk_fma_f32 %0, %1, %1, %0" : "+v"(acc) : "v"(a)); }
// elementwise a*b + c
__device__ __forceinline__ c2 fma2(c2 a, c2 b, c2 c) { return __builtin_elementwise_fma(a, b, c); }
// a * p.x + q.x  /  a * p.y + q.y  (real scale and offset taken from one half of packed pairs)
__device__ __forceinline__ c2 fma_lo(c2 a, c2 p, c2 q) {
    c2 r;
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(r) : "v"(a), "v"(p), "v"(q));
    return r;
}
__device__ __forceinline__ c2 fma_hi(c2 a, c2 p, c2 q) {
    c2 r;
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,1,1] op_sel_hi:[1,1,1]" : "=v"(r) : "v"(a), "v"(p), "v"(q));
    return r;
}
// a * p.x / a * p.y (real scale taken from one half of a packed pair of scalars)
__device__ __forceinline__ c2 scale_lo(c2 a, c2 p) {
    c2 r;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel:[0,0] op_sel_hi:[1,0]" : "=v"(r) : "v"(a), "v"(p));
    return r;
}
__device__ __forceinline__ c2 scale_hi(c2 a, c2 p) {
    c2 r;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,1]" : "=v"(r) : "v"(a), "v"(p));
    return r;
}
#else
typedef cf c2;
GJ_HD c2 to_c2(cf a) { return a; }
GJ_HD cf to_cf(c2 a) { return a; }
GJ_HD c2 make_c2(float x, float y) { return c2{x, y}; }
GJ_HD c2 cmul(c2 a, c2 b) { return c2{a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x}; }
GJ_HD c2 cmul_k(c2 a, c2 b) { return cmul(a, b); }
GJ_HD c2 cmul_add(c2 a, c2 w, c2 c) { return c2{c.x + a.x * w.x - a.y * w.y, c.y + a.x * w.y + a.y * w.x}; }
GJ_HD c2 cmul_add_k(c2 a, c2 w, c2 c) { return cmul_add(a, w, c); }
GJ_HD c2 cmul_1(c2 a, c2 w) { return c2{a.x * w.x, a.x * w.y}; }
GJ_HD c2 cmul_add_1(c2 a, c2 w, c2 c) { return c2{c.x + a.x * w.x, c.y + a.x * w.y}; }
GJ_HD c2 cmul_2(c2 a, c2 w, c2 t) { return c2{t.x - a.y * w.y, t.y + a.y * w.x}; }
GJ_HD c2 cmul_1_k(c2 a, c2 w) { return cmul_1(a, w); }
GJ_HD c2 cmul_add_1_k(c2 a, c2 w, c2 c) { return cmul_add_1(a, w, c); }
GJ_HD c2 cmul_2_k(c2 a, c2 w, c2 t) { return cmul_2(a, w, t); }
GJ_HD c2 twice_minus(c2 u, c2 t) { return c2{2.0f * u.x - t.x, 2.0f * u.y - t.y}; }
GJ_HD c2 cadd(c2 a, c2 b) { return c2{a.x + b.x, a.y + b.y}; }
GJ_HD c2 csub(c2 a, c2 b) { return c2{a.x - b.x, a.y - b.y}; }
GJ_HD c2 add_mj(c2 a, c2 b) { return c2{a.x + b.y, a.y - b.x}; }
GJ_HD c2 add_pj(c2 a, c2 b) { return c2{a.x - b.y, a.y + b.x}; }
GJ_HD c2 mul_mj(c2 a) { return c2{a.y, -a.x}; }
GJ_HD void acc_sq(c2& acc, c2 a) { acc.x += a.x * a.x; acc.y += a.y * a.y; }
GJ_HD c2 fma2(c2 a, c2 b, c2 c) { return c2{a.x * b.x + c.x, a.y * b.y + c.y}; }
GJ_HD c2 fma_lo(c2 a, c2 p, c2 q) { return c2{a.x * p.x + q.x, a.y * p.x + q.x}; }
GJ_HD c2 fma_hi(c2 a, c2 p, c2 q) { return c2{a.x * p.y + q.y, a.y * p.y + q.y}; }
GJ_HD c2 scale_lo(c2 a, c2 p) { return c2{a.x * p.x, a.y * p.x}; }
GJ_HD c2 scale_hi(c2 a, c2 p) { return c2{a.x * p.y, a.y * p.y}; }
#endif

constexpr int kPointsPerThread = 16;
constexpr int kBlockThreads = 256;
constexpr int kBlockPoints = kPointsPerThread * kBlockThreads;   // 4096
constexpr int kTwiddleTable = 4096;                              // W_4096^m, m = 0..4095

constexpr int fft_npass(int n) {
    int p = 0;
    while (n >= 16) { n /= 16; ++p; }
    return p + (n > 1 ? 1 : 0);
}
constexpr int fft_radix(int n, int pass) {
    for (int i = 0; i < pass; ++i) n /= 16;
    return n >= 16 ? 16 : n;
}
constexpr int fft_ns(int n, int pass) {   // product of the radices of the passes before `pass`
    int ns = 1;
    for (int i = 0; i < pass; ++i) ns *= fft_radix(n, i);
    return ns;
}

// ---- small DFTs, forward sign (W = exp(-2 pi i / R)), in place, natural order out -------
GJ_HD void dft2(c2& a, c2& b) {
    c2 t = a;
    a = cadd(t, b);
    b = csub(t, b);
}

GJ_HD void dft4(c2& x0, c2& x1, c2& x2, c2& x3) {
    c2 t0 = cadd(x0, x2), t1 = csub(x0, x2), t2 = cadd(x1, x3), t3 = csub(x1, x3);
    x0 = cadd(t0, t2);
    x2 = csub(t0, t2);
    x1 = add_mj(t1, t3);
    x3 = add_pj(t1, t3);
}

constexpr float kSqrtHalf = 0.70710678118654752440f;
constexpr float kCosPi8 = 0.92387953251128675613f;
constexpr float kSinPi8 = 0.38268343236508977173f;

// the non-trivial constant twiddles inside the 8- and 16-point butterflies
struct InnerTw {
    c2 w16_1, w16_2, w16_3, w16_6, w16_9;
};
GJ_HD InnerTw inner_twiddles() {
    InnerTw k;
    k.w16_1 = make_c2(kCosPi8, -kSinPi8);
    k.w16_2 = make_c2(kSqrtHalf, -kSqrtHalf);    // = W8^1
    k.w16_3 = make_c2(kSinPi8, -kCosPi8);
    k.w16_6 = make_c2(-kSqrtHalf, -kSqrtHalf);   // = W8^3
    k.w16_9 = make_c2(-kCosPi8, kSinPi8);
    return k;
}

template <int R>
GJ_HD void dft(c2 (&a)[R], const InnerTw& k);

template <>
GJ_HD void dft<2>(c2 (&a)[2], const InnerTw&) { dft2(a[0], a[1]); }

template <>
GJ_HD void dft<4>(c2 (&a)[4], const InnerTw&) { dft4(a[0], a[1], a[2], a[3]); }

template <>
GJ_HD void dft<8>(c2 (&a)[8], const InnerTw& k) {
    // n = n1 + 2 n2, k = 4 k1 + k2 :  radix-4 over n2, twiddle W8^(n1 k2), radix-2 over n1
    dft4(a[0], a[2], a[4], a[6]);   // n1 = 0 : A0[k2] in a[2 k2]
    dft4(a[1], a[3], a[5], a[7]);   // n1 = 1 : A1[k2] in a[2 k2 + 1]
    a[3] = cmul_k(a[3], k.w16_2);     // W8^1
    a[5] = mul_mj(a[5]);            // W8^2
    a[7] = cmul_k(a[7], k.w16_6);     // W8^3
    c2 r[8];
#pragma unroll
    for (int k2 = 0; k2 < 4; ++k2) {
        r[k2] = cadd(a[2 * k2], a[2 * k2 + 1]);
        r[k2 + 4] = csub(a[2 * k2], a[2 * k2 + 1]);
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) a[i] = r[i];
}

template <>
GJ_HD void dft<16>(c2 (&a)[16], const InnerTw& k) {
    // n = n1 + 4 n2, k = 4 k1 + k2 :  radix-4 over n2, twiddle W16^(n1 k2), radix-4 over n1
#pragma unroll
    for (int n1 = 0; n1 < 4; ++n1) dft4(a[n1], a[n1 + 4], a[n1 + 8], a[n1 + 12]);   // A[n1][k2] in a[n1 + 4 k2]
    a[5] = cmul_k(a[5], k.w16_1);     // n1 = 1, k2 = 1
    a[9] = cmul_k(a[9], k.w16_2);     // n1 = 1, k2 = 2
    a[13] = cmul_k(a[13], k.w16_3);   // n1 = 1, k2 = 3
    a[6] = cmul_k(a[6], k.w16_2);     // n1 = 2, k2 = 1
    a[10] = mul_mj(a[10]);          // n1 = 2, k2 = 2 : W16^4 = -j
    a[14] = cmul_k(a[14], k.w16_6);   // n1 = 2, k2 = 3
    a[7] = cmul_k(a[7], k.w16_3);     // n1 = 3, k2 = 1
    a[11] = cmul_k(a[11], k.w16_6);   // n1 = 3, k2 = 2
    a[15] = cmul_k(a[15], k.w16_9);   // n1 = 3, k2 = 3
    c2 r[16];
#pragma unroll
    for (int k2 = 0; k2 < 4; ++k2) {
        c2 y0 = a[4 * k2], y1 = a[4 * k2 + 1], y2 = a[4 * k2 + 2], y3 = a[4 * k2 + 3];
        dft4(y0, y1, y2, y3);       // X[4 k1 + k2] = y_k1
        r[k2] = y0;
        r[4 + k2] = y1;
        r[8 + k2] = y2;
        r[12 + k2] = y3;
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) a[i] = r[i];
}

// ---- FMA-form radix-4 butterflies: X = DFT4(x0, w1 x1, w2 x2, w3 x3) in 12 packed ops instead
// of 3 complex multiplies + 8 additions = 14 (t0 = x0 + w2 x2 as two FMAs, t1 = 2 x0 - t0, ...).
GJ_HD void bfly4_tw(c2& x0, c2& x1, c2& x2, c2& x3, c2 w1, c2 w2, c2 w3) {
    const c2 t0a = cmul_add_1(x2, w2, x0);
    const c2 u1a = cmul_1(x1, w1);
    const c2 t0 = cmul_2(x2, w2, t0a);
    const c2 u1 = cmul_2(x1, w1, u1a);
    const c2 t1 = twice_minus(x0, t0);
    const c2 t2a = cmul_add_1(x3, w3, u1);
    const c2 t2 = cmul_2(x3, w3, t2a);
    const c2 t3 = twice_minus(u1, t2);
    x0 = cadd(t0, t2);
    x2 = csub(t0, t2);
    x1 = add_mj(t1, t3);
    x3 = add_pj(t1, t3);
}
GJ_HD void bfly4_tw_k(c2& x0, c2& x1, c2& x2, c2& x3, c2 w1, c2 w2, c2 w3) {   // constant (SGPR) twiddles
    const c2 t0a = cmul_add_1_k(x2, w2, x0);
    const c2 u1a = cmul_1_k(x1, w1);
    const c2 t0 = cmul_2_k(x2, w2, t0a);
    const c2 u1 = cmul_2_k(x1, w1, u1a);
    const c2 t1 = twice_minus(x0, t0);
    const c2 t2a = cmul_add_1_k(x3, w3, u1);
    const c2 t2 = cmul_2_k(x3, w3, t2a);
    const c2 t3 = twice_minus(u1, t2);
    x0 = cadd(t0, t2);
    x2 = csub(t0, t2);
    x1 = add_mj(t1, t3);
    x3 = add_pj(t1, t3);
}
// all four inputs twiddled: 14 packed ops instead of 16
GJ_HD void bfly4_tw4(c2& x0, c2& x1, c2& x2, c2& x3, c2 w0, c2 w1, c2 w2, c2 w3) {
    const c2 u0a = cmul_1(x0, w0);
    const c2 u1a = cmul_1(x1, w1);
    const c2 u0 = cmul_2(x0, w0, u0a);
    const c2 u1 = cmul_2(x1, w1, u1a);
    const c2 t0a = cmul_add_1(x2, w2, u0);
    const c2 t2a = cmul_add_1(x3, w3, u1);
    const c2 t0 = cmul_2(x2, w2, t0a);
    const c2 t2 = cmul_2(x3, w3, t2a);
    const c2 t1 = twice_minus(u0, t0);
    const c2 t3 = twice_minus(u1, t2);
    x0 = cadd(t0, t2);
    x2 = csub(t0, t2);
    x1 = add_mj(t1, t3);
    x3 = add_pj(t1, t3);
}

// second radix-4 layer of the 16-point butterfly (constant twiddles W16^(n1 k2)), FMA form;
// input A[n1][k2] in a[n1 + 4 k2], output natural order
GJ_HD void dft16_layer2(c2 (&a)[16], const InnerTw& k) {
    dft4(a[0], a[1], a[2], a[3]);                                    // k2 = 0: no twiddles
    bfly4_tw_k(a[4], a[5], a[6], a[7], k.w16_1, k.w16_2, k.w16_3);   // k2 = 1
    {                                                                // k2 = 2: W16^2, -j, W16^6
        c2 &y0 = a[8], &y1 = a[9], &y2 = a[10], &y3 = a[11];
        const c2 u1a = cmul_1_k(y1, k.w16_2);
        const c2 t0 = add_mj(y0, y2);                                // y0 - j y2
        const c2 u1 = cmul_2_k(y1, k.w16_2, u1a);
        const c2 t1 = add_pj(y0, y2);                                // y0 + j y2
        const c2 t2a = cmul_add_1_k(y3, k.w16_6, u1);
        const c2 t2 = cmul_2_k(y3, k.w16_6, t2a);
        const c2 t3 = twice_minus(u1, t2);
        y0 = cadd(t0, t2);
        y2 = csub(t0, t2);
        y1 = add_mj(t1, t3);
        y3 = add_pj(t1, t3);
    }
    bfly4_tw_k(a[12], a[13], a[14], a[15], k.w16_3, k.w16_6, k.w16_9);   // k2 = 3
    c2 r[16];
#pragma unroll
    for (int k2 = 0; k2 < 4; ++k2) {
        r[k2] = a[4 * k2];            // X[4 k1 + k2] = k1-th output of group k2
        r[4 + k2] = a[4 * k2 + 1];
        r[8 + k2] = a[4 * k2 + 2];
        r[12 + k2] = a[4 * k2 + 3];
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) a[i] = r[i];
}

// 16-point butterfly of the first pass (no input twiddles), FMA form
GJ_HD void dft16_fma(c2 (&a)[16], const InnerTw& k) {
#pragma unroll
    for (int n1 = 0; n1 < 4; ++n1) dft4(a[n1], a[n1 + 4], a[n1 + 8], a[n1 + 12]);
    dft16_layer2(a, k);
}

// 16-point butterfly with input twiddles tw[t-1] = W^(t k), t = 1..15, FMA form:
// 54 + 47 packed ops instead of 30 + 64 + 17
GJ_HD void dft16_fma_tw(c2 (&a)[16], const c2* tw, const InnerTw& k) {
    bfly4_tw(a[0], a[4], a[8], a[12], tw[3], tw[7], tw[11]);
#pragma unroll
    for (int n1 = 1; n1 < 4; ++n1)
        bfly4_tw4(a[n1], a[n1 + 4], a[n1 + 8], a[n1 + 12], tw[n1 - 1], tw[n1 + 3], tw[n1 + 7], tw[n1 + 11]);
    dft16_layer2(a, k);
}

// Radix-16 butterfly with its input twiddles W^(t k), t = n1 + 4 n2, applied in two steps:
// W^(4 n2 k) before the first radix-4 stage and W^(n1 k) after it (constant over the sum on
// n2).  Six twiddle values (12 VGPRs) instead of fifteen (30) per pass for nine more complex
// multiplies: the registers buy a third resident workgroup per CU.
// tw6 = { W^(4k), W^(8k), W^(12k), W^(k), W^(2k), W^(3k) }.
GJ_HD void dft16_twiddled(c2 (&a)[16], const c2* tw6, const InnerTw& k) {
#pragma unroll
    for (int n2 = 1; n2 < 4; ++n2)
#pragma unroll
        for (int n1 = 0; n1 < 4; ++n1) a[n1 + 4 * n2] = cmul(a[n1 + 4 * n2], tw6[n2 - 1]);
#pragma unroll
    for (int n1 = 0; n1 < 4; ++n1) dft4(a[n1], a[n1 + 4], a[n1 + 8], a[n1 + 12]);   // A[n1][k2] in a[n1 + 4 k2]
#pragma unroll
    for (int n1 = 1; n1 < 4; ++n1)
#pragma unroll
        for (int k2 = 0; k2 < 4; ++k2) a[n1 + 4 * k2] = cmul(a[n1 + 4 * k2], tw6[2 + n1]);
    a[5] = cmul_k(a[5], k.w16_1);
    a[9] = cmul_k(a[9], k.w16_2);
    a[13] = cmul_k(a[13], k.w16_3);
    a[6] = cmul_k(a[6], k.w16_2);
    a[10] = mul_mj(a[10]);
    a[14] = cmul_k(a[14], k.w16_6);
    a[7] = cmul_k(a[7], k.w16_3);
    a[11] = cmul_k(a[11], k.w16_6);
    a[15] = cmul_k(a[15], k.w16_9);
    c2 r[16];
#pragma unroll
    for (int k2 = 0; k2 < 4; ++k2) {
        c2 y0 = a[4 * k2], y1 = a[4 * k2 + 1], y2 = a[4 * k2 + 2], y3 = a[4 * k2 + 3];
        dft4(y0, y1, y2, y3);
        r[k2] = y0;
        r[4 + k2] = y1;
        r[8 + k2] = y2;
        r[12 + k2] = y3;
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) a[i] = r[i];
}

// LDS element index (8-byte elements) of logical point i of a transform whose region
// starts at `base`.
// Additive padding (one spare slot per 16) rather than an XOR swizzle: every address a thread
// touches in one exchange is then "thread base + compile-time constant", so the 16 scatter
// and 16 gather addresses fold into the DS instructions' immediate offsets instead of
// occupying 64 loop-invariant VGPRs.  Writes are conflict free, reads 2-way
// (tools/lds_bank_sim.py); N = 4096 in the Welch kernel uses the X4096 schedule below, which
// is conflict free on both sides.
GJ_HD int lds_slot(int base, int i) { return base + i + (i >> 4); }
constexpr int lds_span(int n) { return n + n / 16; }   // slots one N-point transform occupies

// One Stockham pass over the 16 register-resident points of thread `jl` (0 <= jl < N/16).
// v[s] holds in[jl + (N/16) s] on entry.  On exit the logical output of butterfly u,
// leg t sits in v[u + t*(16/R)] and belongs at index out_index<N,PASS>(jl, u, t) of the
// next pass' input (or IS X[jl + (N/16)(u + t*(16/R))] after the last pass).
// tw[u*(R-1) + t-1] = W_(Ns R)^(t * ((jl + (N/16) u) mod Ns)), unused when PASS == 0.
// TWO_STEP (radix-16 passes after the first only): tw holds the six values of dft16_twiddled.
// FMA_FORM (radix-16 passes): the FMA-form butterflies above (same twiddle registers as the
// default form).
template <int N, int PASS, bool TWO_STEP = false, bool FMA_FORM = false>
GJ_HD void fft_pass(c2 (&v)[16], const c2* tw, const InnerTw& k) {
    constexpr int R = fft_radix(N, PASS);
    constexpr int G = 16 / R;   // butterflies per thread
    if constexpr (TWO_STEP && PASS > 0 && R == 16) {
        dft16_twiddled(v, tw, k);
        return;
    }
    if constexpr (FMA_FORM && R == 16) {
        if constexpr (PASS > 0) dft16_fma_tw(v, tw, k);
        else dft16_fma(v, k);
        return;
    }
#pragma unroll
    for (int u = 0; u < G; ++u) {
        c2 a[R];
#pragma unroll
        for (int t = 0; t < R; ++t) a[t] = v[u + t * G];
        if (PASS > 0) {
#pragma unroll
            for (int t = 1; t < R; ++t) a[t] = cmul(a[t], tw[u * (R - 1) + t - 1]);
        }
        dft<R>(a, k);
#pragma unroll
        for (int t = 0; t < R; ++t) v[u + t * G] = a[t];
    }
}

template <int N, int PASS>
GJ_HD int out_index(int jl, int u, int t) {
    constexpr int R = fft_radix(N, PASS);
    constexpr int NS = fft_ns(N, PASS);
    const int q = jl + (N / 16) * u;
    const int k = q & (NS - 1);
    return (q - k) * R + k + t * NS;
}

// index into the W_4096 table of the twiddle of butterfly u, leg t of thread jl
template <int N, int PASS>
GJ_HD int twiddle_index(int jl, int u, int t) {
    constexpr int R = fft_radix(N, PASS);
    constexpr int NS = fft_ns(N, PASS);
    const int k = (jl + (N / 16) * u) & (NS - 1);
    return (t * k * (kTwiddleTable / (NS * R))) & (kTwiddleTable - 1);
}

// the six twiddles of dft16_twiddled for thread jl (radix-16 pass, one butterfly per thread)
template <int N, int PASS>
GJ_HD void load_twiddles6(c2* tw6, const cf* table, int jl) {
    constexpr int NS = fft_ns(N, PASS);
    const int k = jl & (NS - 1);
    constexpr int step = kTwiddleTable / (NS * 16);
    const int e[6] = {4, 8, 12, 1, 2, 3};
#pragma unroll
    for (int i = 0; i < 6; ++i) tw6[i] = to_c2(table[(e[i] * k * step) & (kTwiddleTable - 1)]);
}

template <int N, int PASS>
GJ_HD void load_twiddles(c2* tw, const cf* table, int jl) {
    constexpr int R = fft_radix(N, PASS);
    constexpr int G = 16 / R;
#pragma unroll
    for (int u = 0; u < G; ++u)
#pragma unroll
        for (int t = 1; t < R; ++t) tw[u * (R - 1) + t - 1] = to_c2(table[twiddle_index<N, PASS>(jl, u, t)]);
}

// The slot of index i0 + c is slot(i0) + c + c/16 whenever c is a multiple of 16 (adding a
// multiple of 16 leaves the low four bits alone), so every address below is written as
// "slot of the thread's own base index + compile-time constant" explicitly -- hipcc does not
// derive that from (i0 + c) >> 4 by itself and otherwise spends a VGPR per address.
template <int N, int PASS>
GJ_HD void lds_scatter(const c2 (&v)[16], cf* lds, int base, int jl) {
    constexpr int R = fft_radix(N, PASS);
    constexpr int G = 16 / R;
    constexpr int NS = fft_ns(N, PASS);
#pragma unroll
    for (int u = 0; u < G; ++u) {
        if constexpr (NS % 16 == 0) {
            const int s0 = lds_slot(base, out_index<N, PASS>(jl, u, 0));
#pragma unroll
            for (int t = 0; t < R; ++t) lds[s0 + t * NS + t * NS / 16] = to_cf(v[u + t * G]);
        } else if constexpr (NS == 1 && R == 16) {
            const int s0 = lds_slot(base, out_index<N, PASS>(jl, u, 0));   // index 16 q: slot 17 q
#pragma unroll
            for (int t = 0; t < R; ++t) lds[s0 + t] = to_cf(v[u + t * G]);
        } else {
#pragma unroll
            for (int t = 0; t < R; ++t) lds[lds_slot(base, out_index<N, PASS>(jl, u, t))] = to_cf(v[u + t * G]);
        }
    }
}

template <int N>
GJ_HD void lds_gather(c2 (&v)[16], const cf* lds, int base, int jl) {
    constexpr int TF = N / 16;
    if constexpr (TF % 16 == 0) {
        const int s0 = lds_slot(base, jl);
#pragma unroll
        for (int s = 0; s < 16; ++s) v[s] = to_c2(lds[s0 + TF * s + TF * s / 16]);
    } else {
#pragma unroll
        for (int s = 0; s < 16; ++s) v[s] = to_c2(lds[lds_slot(base, jl + TF * s)]);
    }
}

// ---------------------------------------------------------------------------------------
// Conflict-free exchange schedule for N = 4096 (256 threads, three radix-16 passes)
// ---------------------------------------------------------------------------------------
// With the identity thread->butterfly map no additive layout is conflict free for both the
// 16-lane write groups and the 32-lane read groups (tools/lds_bank_sim.py).  It becomes
// possible when the thread changes role at the first exchange: thread `tid` computes
// butterfly jl0 = tid in pass 0 (so the global loads stay coalesced) and butterfly
// jl1 = 16 (tid & 15) + (tid >> 4) in passes 1 and 2, with one layout per exchange:
//   exchange 0: slot(256 S + 16 M + T) = 286 S + 17 M + c0(T),  c0(T) = (T & ~1) + 16 (T & 1)
//   exchange 1: slot(256 S + 16 M + T) = 287 S + 18 M + T
// Every address is still "thread base + compile-time constant".  Bank arithmetic (8-byte
// slots, 16 slots per write group, 32 per read group):
//   exchange 0 write: 16 lanes = 16 M at one (S, T): 17 M covers every residue mod 16;
//   exchange 0 read : 32 lanes = 16 M x T in {2h, 2h+1}: 17 M + {c0(2h), c0(2h) + 16} mod 32;
//   exchange 1 write: 16 lanes = 16 S at one (M, T): 287 S is odd -> every residue mod 16;
//   exchange 1 read : 32 lanes = 16 M x T in {2h, 2h+1}: 18 M covers the even residues mod 32.
// tools/lds_bank_sim.py --xpose counts 1.0 cycles per group for all four; SQ_LDS_BANK_CONFLICT
// reads 0 on MI355X (67 M per launch with the generic layout).
// After the last pass thread tid holds X[jl1 + 256 s] in v[s].
struct X4096 {
    static constexpr int kSpan = 4592;   // slots per buffer (max of the two layouts)
    static constexpr int c0(int t) { return (t & ~1) + 16 * (t & 1); }
    GJ_HD static int jl1(int tid) { return 16 * (tid & 15) + (tid >> 4); }
    GJ_HD static int slot0(int i) { return 286 * (i >> 8) + 17 * ((i >> 4) & 15) + c0(i & 15); }
    GJ_HD static int slot1(int i) { return 287 * (i >> 8) + 18 * ((i >> 4) & 15) + (i & 15); }
};

// EX = 0: scatter the outputs of pass 0 (thread role jl0 = tid); EX = 1: of pass 1 (role jl1)
template <int EX>
GJ_HD void x4096_scatter(const c2 (&v)[16], cf* lds, int tid) {
    if constexpr (EX == 0) {
        const int base = 286 * (tid >> 4) + 17 * (tid & 15);        // slot0(16 tid + t) - c0(t)
#pragma unroll
        for (int t = 0; t < 16; ++t) lds[base + X4096::c0(t)] = to_cf(v[t]);
    } else {
        const int base = 287 * (tid & 15) + (tid >> 4);             // slot1(256 (jl1 >> 4) + (jl1 & 15) + 16 t) - 18 t
#pragma unroll
        for (int t = 0; t < 16; ++t) lds[base + 18 * t] = to_cf(v[t]);
    }
}

// gather in[jl1 + 256 s] for the next pass (role jl1 on both exchanges)
template <int EX>
GJ_HD void x4096_gather(c2 (&v)[16], const cf* lds, int tid) {
    if constexpr (EX == 0) {
        const int base = 17 * (tid & 15) + X4096::c0(tid >> 4);
#pragma unroll
        for (int s = 0; s < 16; ++s) v[s] = to_c2(lds[base + 286 * s]);
    } else {
        const int base = 18 * (tid & 15) + (tid >> 4);
#pragma unroll
        for (int s = 0; s < 16; ++s) v[s] = to_c2(lds[base + 287 * s]);
    }
}

}   // namespace gj
