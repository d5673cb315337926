// GNSS acquisition search as a batched FFT cross-correlation (gfx950) -- SURVEY section 8(f)-4.
// Replaces, for all PRNs and Doppler bins at once, the reference receiver's per-channel loop
//   sdraqcuisition      GpsJammerApp/backend/sdracq.c:3-50    (up to `intg` non-coherent steps, early stop)
//   pcorrelator         GpsJammerApp/backend/sdrcmn.c:742-773 (per Doppler bin: mixcarr -> cpxcpx -> cpxconv)
//   mixcarr (SSE2 form) sdrcmn.c:618-705                      (16-entry int8 cos/sin table, phase index per sample)
//   cpxconv             sdrcmn.c:124-147                      (FFT . conj(code FFT) . IFFT -> |.|^2 / m^2, summed)
//   checkacquisition    sdracq.c:52-84                        (peak, +-2 chip exclusion, peak ratio > 3)
// on the block FFT of fft_core.h (the machinery of K2 / K5):
//   acq_prep_kernel : C_p   = FFT(resampled code of PRN p, zero-padded to nfft)            once per search
//                     X_s,f = FFT(mix(data window s, Doppler bin f) * CSCALE/m)            n_freq x intg transforms, same launch
//   acq_inv_kernel  : P_p,f += |IFFT(-X_s,f conj(C_p))|^2 / m^2 over the first nsamp lags   n_freq x n_prn per step
//   acq_check_kernel: per PRN the reference's peak test; a PRN that passes stops integrating (device flag)
// The data FFT does not depend on the PRN, so it is computed once per (step, bin) and reused by all PRNs
// (the reference recomputes it in every channel thread).  The mixer's phase index per sample comes as a table
// built on the host exactly like the reference builds it (doubles accumulated in its order, truncation toward
// zero): the integer mixer is bit-exact, everything behind the first FFT is float32 like FFTW's.
#include "gj_common.h"

namespace gj {

struct AcqParams {
    const uint8_t* iq;
    unsigned long long first_sample;
    int nsamp, nfft, n_freq, n_prn, intg, offset;
    float scale;   // CSCALE / m
    float inv_m2;  // 1 / m^2
};

template <int N, int PASS>
__device__ __forceinline__ void acq_passes(c2 (&v)[16], cf* lds, int base, int jl, const cf* twtab) {
    constexpr int NP = fft_npass(N);
    c2 tw[15];
    if constexpr (PASS > 0) load_twiddles<N, PASS>(tw, twtab, jl);
    fft_pass<N, PASS, false, true>(v, tw, inner_twiddles());
    if constexpr (PASS + 1 < NP) {
        lds_scatter<N, PASS>(v, lds, base, jl);
        __syncthreads();
        lds_gather<N>(v, lds, base, jl);
        __syncthreads();
        acq_passes<N, PASS + 1>(v, lds, base, jl, twtab);
    }
}

// the same with the pass twiddles already in registers (loops that transform many times)
template <int N, int PASS>
__device__ __forceinline__ void acq_passes_tw(c2 (&v)[16], cf* lds, int base, int jl, const c2 (&tw)[3][15]) {
    constexpr int NP = fft_npass(N);
    fft_pass<N, PASS, false, true>(v, tw[PASS], inner_twiddles());
    if constexpr (PASS + 1 < NP) {
        lds_scatter<N, PASS>(v, lds, base, jl);
        __syncthreads();
        lds_gather<N>(v, lds, base, jl);
        __syncthreads();
        acq_passes_tw<N, PASS + 1>(v, lds, base, jl, tw);
    }
}

// 16-entry mixer tables of the reference's SSE2 path: (char)floor(cos(2 pi i / 16) / CSCALE + 0.5), CSCALE = 1/32
__constant__ signed char kAcqCos[16] = {32, 30, 23, 12, 0, -12, -23, -30, -32, -30, -23, -12, 0, 12, 23, 30};
__constant__ signed char kAcqSin[16] = {0, 12, 23, 30, 32, 30, 23, 12, 0, -12, -23, -30, -32, -30, -23, -12};

// one transform per workgroup (N = 4096) or 4096 / N of them (smaller N): `which` = transform index
// (blk of nblk: the workgroup's index among those that transform codes)
template <int N>
__device__ __forceinline__ void acq_code_body(const short* __restrict__ codes, int nsamp, int n_prn, const cf* __restrict__ twtab,
                                              cf* __restrict__ cspec, unsigned long long* __restrict__ gmax, int n_gmax, cf* lds,
                                              int blk, int nblk) {
    constexpr int TF = N / 16, B = kBlockPoints / N;
    // first launch of a search: clear the running row maxima of acq_inv_all_kernel (the next launch in stream order)
    for (int i = blk * kBlockThreads + threadIdx.x; i < n_gmax; i += nblk * kBlockThreads) gmax[i] = 0ull;
    const int tid = threadIdx.x, b = tid / TF, jl = tid % TF;
    const int p = blk * B + b;
    const bool live = p < n_prn;
    c2 v[16];
#pragma unroll
    for (int s = 0; s < 16; ++s) {
        const int n = jl + TF * s;
        v[s] = make_c2((live && n < nsamp) ? (float)codes[(size_t)p * nsamp + n] : 0.f, 0.f);   // cpxcpx(rcode, NULL, 1.0)
    }
    acq_passes<N, 0>(v, lds, b * lds_span(N), jl, twtab);
    if (live) {
        // Stored as -C_p / m.  1/m: cpxconv's |.|^2 / m^2 (sdrcmn.c:141-143) then needs no multiply per lag; m = N is a
        // power of two, so the scaling is exact and commutes with every float operation behind it.  The sign, together
        // with the CONJUGATED data spectrum acq_fwd_body stores, turns cpxconv's product (real = -p0 q0 - p1 q1,
        // imag = p0 q1 - p1 q0, conjugated for the forward-FFT inverse: sdrcmn.c:131-135) into one plain complex multiply
        // conj(X) * (-C): four multiply-adds per point and no sign flips (five operations before).
        constexpr float inv_m = -1.0f / (float)N;
#pragma unroll
        for (int s = 0; s < 16; ++s) cspec[(size_t)p * N + jl + TF * s] = to_cf(make_c2(v[s].x * inv_m, v[s].y * inv_m));
    }
}

template <int N>
__device__ __forceinline__ void acq_fwd_body(const AcqParams& P, const uint8_t* __restrict__ phase, const cf* __restrict__ twtab,
                                             cf* __restrict__ xspec, cf* lds, int blk) {
    constexpr int TF = N / 16, B = kBlockPoints / N;
    const int tid = threadIdx.x, b = tid / TF, jl = tid % TF;
    const int t = blk * B + b;                 // transform = step * n_freq + bin
    const bool live = t < P.intg * P.n_freq;
    const int s_idx = live ? t / P.n_freq : 0, f = live ? t % P.n_freq : 0;
    const uint16_t* src = reinterpret_cast<const uint16_t*>(P.iq) + P.first_sample + (size_t)s_idx * P.nsamp;
    const uint8_t* ph = phase + (size_t)f * N;
    c2 v[16];
#pragma unroll
    for (int s = 0; s < 16; ++s) {
        const int n = jl + TF * s;
        const unsigned w = src[n];
        const int di = (int)(w & 255u) - P.offset, dq = (int)(w >> 8) - P.offset;   // (char)(u - 128), sdrrcv.c:104-106
        const int k = ph[n] & 15;
        const int co = kAcqCos[k], si = kAcqSin[k];
        const int ii = co * di - si * dq, qq = si * di + co * dq;                    // int16 in the reference: |.| <= 8192
        v[s] = make_c2((float)ii * P.scale, (float)qq * P.scale);                    // cpxcpx(dataI, dataQ, CSCALE / m)
    }
    acq_passes<N, 0>(v, lds, b * lds_span(N), jl, twtab);
    if (live) {
#pragma unroll
        for (int s = 0; s < 16; ++s) xspec[(size_t)t * N + jl + TF * s] = to_cf(make_c2(v[s].x, -v[s].y));   // conj(X): see acq_code_body
    }
}
// ONE launch for everything a search transforms before the correlations: the first code_blocks workgroups the PRNs'
// codes, the others the mixed data windows (independent work; two launches of a few microseconds each cost a
// launch-to-launch gap in a chain that is 0.18 ms long)
template <int N>
__global__ __launch_bounds__(kBlockThreads) void acq_prep_kernel(AcqParams P, const short* __restrict__ codes,
                                                                 const uint8_t* __restrict__ phase, const cf* __restrict__ twtab,
                                                                 cf* __restrict__ cspec, cf* __restrict__ xspec,
                                                                 unsigned long long* __restrict__ gmax, int n_gmax, int code_blocks) {
    constexpr int B = kBlockPoints / N;
    __shared__ cf lds[B * lds_span(N)];
    if ((int)blockIdx.x < code_blocks) acq_code_body<N>(codes, P.nsamp, P.n_prn, twtab, cspec, gmax, n_gmax, lds, (int)blockIdx.x, code_blocks);
    else acq_fwd_body<N>(P, phase, twtab, xspec, lds, (int)blockIdx.x - code_blocks);
}

// grid.x = ceil(n_freq / B), grid.y = PRN.  |IFFT(Y)|^2 = |FFT(conj Y)|^2, Y = -X conj(C)  (cpxconv's product).
template <int N>
__global__ __launch_bounds__(kBlockThreads) void acq_inv_kernel(AcqParams P, int step, const int* __restrict__ done,
                                                                const cf* __restrict__ twtab,
                                                                const cf* __restrict__ xspec,
                                                                const cf* __restrict__ cspec, double* __restrict__ power) {
    constexpr int TF = N / 16, B = kBlockPoints / N;
    __shared__ cf lds[B * lds_span(N)];
    const int p = blockIdx.y;
    if (done[p]) return;                       // this PRN has been acquired at an earlier step (sdracq.c:24-27)
    const int tid = threadIdx.x, b = tid / TF, jl = tid % TF;
    const int f = blockIdx.x * B + b;
    const bool live = f < P.n_freq;
    const cf* x = xspec + ((size_t)step * P.n_freq + (live ? f : 0)) * N;
    const cf* c = cspec + (size_t)p * N;
    c2 v[16];
#pragma unroll
    for (int s = 0; s < 16; ++s) {
        // sdrcmn.c:131-135: real = -p0 q0 - p1 q1, imag = p0 q1 - p1 q0; conjugated for the forward-FFT inverse
        // = conj(X) * (-C), with conj(X) and -C / m as stored
        const cf a = x[jl + TF * s], q = c[jl + TF * s];
        v[s] = make_c2(a.x * q.x - a.y * q.y, a.x * q.y + a.y * q.x);
    }
    acq_passes<N, 0>(v, lds, b * lds_span(N), jl, twtab);
    if (live) {
        double* dst = power + ((size_t)p * P.n_freq + f) * P.nsamp;
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            const int k = jl + TF * s;
            if (k < P.nsamp) dst[k] += (double)(v[s].x * v[s].x + v[s].y * v[s].y);   // flagsum = 1 (:141-143); 1/m^2 rides on C_p
        }
    }
}

struct AcqBest {
    double val;
    int idx;
};
__device__ __forceinline__ AcqBest acq_better(AcqBest a, AcqBest b) {   // larger value, first index on a tie (maxvd)
    return (b.val > a.val || (b.val == a.val && b.idx < a.idx)) ? b : a;
}
__device__ AcqBest acq_block_best(AcqBest v, AcqBest* sh) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        AcqBest o;
        o.val = __shfl_xor(v.val, off, 64);
        o.idx = __shfl_xor(v.idx, off, 64);
        v = acq_better(v, o);
    }
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    AcqBest r = sh[0];
    for (unsigned k = 1; k < blockDim.x / 64; ++k) r = acq_better(r, sh[k]);
    return r;
}

// checkacquisition (sdracq.c:52-84) for one PRN per workgroup, with maxvd / meanvd / ind2sub as written
// (sdrcmn.c:411-440,512-515): strict '>' keeps the FIRST maximum; index 0 of a row seeds maxvd even when it lies in
// the exclusion zone; a wrapped exclusion zone (exinds > exinde) keeps only the indices strictly between.
__global__ __launch_bounds__(1024) void acq_check_kernel(AcqParams P, int step, int nsampchip, double ctime, float threshold,
                                                         const double* __restrict__ power, int* __restrict__ done,
                                                         gj_acq_result* __restrict__ out) {
    __shared__ AcqBest sh[16];
    __shared__ double shs[16];
    __shared__ int shn[16];
    const int p = blockIdx.x;
    if (done[p]) return;
    const double* pw = power + (size_t)p * P.n_freq * P.nsamp;
    const int total = P.n_freq * P.nsamp;
    AcqBest b{-1.0, 0x7fffffff};
    for (int i = threadIdx.x; i < total; i += blockDim.x) b = acq_better(b, AcqBest{pw[i], i});
    b = acq_block_best(b, sh);
    if (b.idx >= total) b.idx = 0;          // only if every power were NaN
    const int codei = b.idx % P.nsamp, freqi = b.idx / P.nsamp;
    int exinds = codei - 2 * nsampchip, exinde = codei + 2 * nsampchip;
    if (exinds < 0) exinds += P.nsamp;
    if (exinde >= P.nsamp) exinde -= P.nsamp;
    const double* row = pw + (size_t)freqi * P.nsamp;
    auto kept = [&](int i) {
        return (exinds <= exinde) ? (i < exinds || i > exinde) : (i < exinds && i > exinde);
    };
    AcqBest b2{row[0], 0};                       // maxvd: max = data[0] whatever the exclusion zone says
    double sum = 0.0;
    int cnt = 0;
    for (int i = threadIdx.x; i < P.nsamp; i += blockDim.x) {
        if (kept(i)) {
            if (i >= 1) b2 = acq_better(b2, AcqBest{row[i], i});
            sum += row[i];
            ++cnt;
        }
    }
    b2 = acq_block_best(b2, sh);
    sum = wave_sum_f64(sum);
    cnt = (int)wave_sum_u32((unsigned)cnt);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) { shs[threadIdx.x >> 6] = sum; shn[threadIdx.x >> 6] = cnt; }
    __syncthreads();
    if (threadIdx.x == 0) {
        double tot = 0.0;
        int n = 0;
        for (unsigned k = 0; k < blockDim.x / 64; ++k) { tot += shs[k]; n += shn[k]; }
        const double meanP = tot / (double)n;
        gj_acq_result r;
        r.max_power = b.val;
        r.second_power = b2.val;
        r.mean_power = meanP;
        r.peak_ratio = b.val / b2.val;
        r.cn0 = 10.0 * log10(b.val / meanP / ctime);
        r.code_index = codei;
        r.freq_index = freqi;
        r.steps = step + 1;
        r.acquired = r.peak_ratio > (double)threshold ? 1 : 0;
        out[p] = r;
        if (r.acquired) done[p] = 1;
    }
}

// ---- single-launch form (no d_power requested) ---------------------------------------------------
// One workgroup per (PRN, Doppler bin) runs ALL integration steps: the code spectrum and the running power of
// its row stay in registers, nothing of P goes through memory.  After every step it records what
// checkacquisition would need from this row IF it turned out to hold the PRN's peak: the row's first maximum,
// and -- with the exclusion zone centred on that maximum -- maxvd's second peak and meanvd's sum.  The
// summary kernel then replays the reference's step loop per PRN on those 71 x intg records.
struct AcqRow {
    double maxv, max2, sum;
    int argk, cnt;
};

// Reduction over the TF = N / 16 threads of one transform.  Inside a wave: four DPP steps per row of sixteen lanes
// (quad_perm swaps, row_half_mirror, row_mirror -- plain VALU moves, no ds_bpermute round trip per step), then the
// four row results through v_readlane (wave-uniform).  N >= 2048: the transform's waves meet through `sh` with ONE
// barrier -- each of the two reductions of a step has its own `sh`, and between two uses of it lie the other
// reduction's barrier and those of the exchanges; the barrier also orders the last gather of the X4096 exchange
// before the next step's first scatter.
__device__ __forceinline__ int acq_dpp(int v, int ctrl_sel) {
    switch (ctrl_sel) {
        case 0: return __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xf, 0xf, false);    // quad_perm [1,0,3,2]
        case 1: return __builtin_amdgcn_update_dpp(0, v, 0x4E, 0xf, 0xf, false);    // quad_perm [2,3,0,1]
        case 2: return __builtin_amdgcn_update_dpp(0, v, 0x141, 0xf, 0xf, false);   // row_half_mirror
        default: return __builtin_amdgcn_update_dpp(0, v, 0x140, 0xf, 0xf, false);  // row_mirror
    }
}
__device__ __forceinline__ double acq_dpp(double v, int ctrl_sel) {
    return __hiloint2double(acq_dpp(__double2hiint(v), ctrl_sel), acq_dpp(__double2loint(v), ctrl_sel));
}
__device__ __forceinline__ double acq_lane(double v, int lane) {
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), lane), __builtin_amdgcn_readlane(__double2loint(v), lane));
}

template <int N, typename T, typename F>
__device__ __forceinline__ T acq_group_reduce(T v, F&& op, T* sh /* [kBlockThreads / 64] */, int b) {
    constexpr int TF = N / 16, WPF = TF / 64;   // waves per transform (N >= 1024)
#pragma unroll
    for (int k = 0; k < 4; ++k) v = op(v, v.moved(k));
    v = op(op(v.lane(0), v.lane(16)), op(v.lane(32), v.lane(48)));
    if constexpr (WPF == 1) return v;
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    T r = sh[b * WPF];
#pragma unroll
    for (int k = 1; k < WPF; ++k) r = op(r, sh[b * WPF + k]);
    return r;
}

struct AcqMax {     // a row's maximum alone: what EVERY row needs every step
    double v;
    __device__ __forceinline__ AcqMax moved(int k) const { return AcqMax{acq_dpp(v, k)}; }
    __device__ __forceinline__ AcqMax lane(int l) const { return AcqMax{acq_lane(v, l)}; }
};

struct AcqBestS : AcqBest {
    __device__ __forceinline__ AcqBestS moved(int k) const {
        AcqBestS o;
        o.val = acq_dpp(val, k);
        o.idx = acq_dpp(idx, k);
        return o;
    }
    __device__ __forceinline__ AcqBestS lane(int l) const {
        AcqBestS o;
        o.val = acq_lane(val, l);
        o.idx = __builtin_amdgcn_readlane(idx, l);
        return o;
    }
};
struct AcqTail {   // what checkacquisition takes from the rest of the row: meanvd's sum and maxvd's second peak
    double sum, mx2;
    __device__ __forceinline__ AcqTail moved(int k) const { return AcqTail{acq_dpp(sum, k), acq_dpp(mx2, k)}; }
    __device__ __forceinline__ AcqTail lane(int l) const { return AcqTail{acq_lane(sum, l), acq_lane(mx2, l)}; }
};

// The three passes of one transform for the single-launch kernel.  N = 4096 takes the bank-conflict-free X4096
// schedule of fft_core.h (thread tid transforms inputs tid + 256 s in pass 0 and ends up holding bins
// jl1(tid) + 256 s); the barrier behind the last gather is left to the reductions that follow.
template <int N>
__device__ __forceinline__ void acq_transform(c2 (&v)[16], cf* lds, int tid, int base, int jl, const c2 (&tw)[3][15]) {
    if constexpr (N == 4096) {
        const InnerTw k = inner_twiddles();
        fft_pass<4096, 0, false, true>(v, tw[0], k);
        x4096_scatter<0>(v, lds, tid);
        __syncthreads();
        x4096_gather<0>(v, lds, tid);
        __syncthreads();
        fft_pass<4096, 1, false, true>(v, tw[1], k);
        x4096_scatter<1>(v, lds, tid);
        __syncthreads();
        x4096_gather<1>(v, lds, tid);
        fft_pass<4096, 2, false, true>(v, tw[2], k);
    } else {
        acq_passes_tw<N, 0>(v, lds, base, jl, tw);
    }
}

// Grid: 8 x n_prn x ceil(groups / 8) workgroups, a group = the kBlockPoints / N Doppler bins one workgroup transforms
// together.  Workgroups b and b + 8 share an XCD and its 4-MiB L2 (observed placement; a speed matter only): the bins
// are dealt over the eight residues of b and the PRN runs fastest inside a residue, so the ~96 workgroups an XCD holds
// at a time read the same three bins' data spectra (10 steps x 32 KB each) and the 1 MB of code spectra out of their
// own L2 instead of each XCD streaming all 23 MB of data spectra from the Infinity Cache.
//
// Round 4 -- the peak bookkeeping only where it can matter.  checkacquisition looks at ONE row per (PRN, step): the row
// that holds the step's largest power.  Rounds 2-3 had every one of the 71 rows work out, every step, what the check
// would need IF it were that row (first maximum with its index, then -- around that index -- the second peak and the
// mean outside the +-2-chip zone: two group reductions of (double, int) / (double, double) records, as many vector
// instructions as the transform itself).  Now every row reduces its MAXIMUM only (one value), and thread 0 of the row
// offers it to gmax[PRN][step], the running maximum over the rows seen so far (one L2 atomic, the bits of a
// non-negative double order like an unsigned integer).  A row whose maximum is below the running one cannot be the
// step's winner whatever comes later and skips the rest; a row that raises or equals it works the statistics out
// exactly as before.  The step's true winner -- and every row that ties with it -- always does (nothing larger can
// have been offered before it), so acq_summary_kernel finds the same record with the same bits; with the rows arriving
// in arbitrary order about ln 71 = 4-5 of the 71 do the full bookkeeping instead of all.  Records of skipped rows
// carry their maximum and cnt = 0.
template <int N>
__global__ __launch_bounds__(kBlockThreads, N == 2048 ? 2 : 3) void acq_inv_all_kernel(AcqParams P, int nsampchip, const cf* __restrict__ twtab,
                                                                    const cf* __restrict__ xspec,
                                                                    const cf* __restrict__ cspec,
                                                                    AcqRow* __restrict__ rows /* [n_prn][intg][n_freq] */,
                                                                    unsigned long long* __restrict__ gmax /* [n_prn][intg] */) {
    constexpr int TF = N / 16, B = kBlockPoints / N;
    constexpr bool XP = N == 4096;
    __shared__ cf lds[XP ? X4096::kSpan : B * lds_span(N)];
    __shared__ AcqMax shm[kBlockThreads / 64];
    __shared__ AcqBestS shb[kBlockThreads / 64];
    __shared__ AcqTail sht[kBlockThreads / 64];
    __shared__ int sh_need[2][B];               // [step parity][transform]: this row may hold the step's peak
    __shared__ unsigned long long sh_seen[2];   // (one transform per workgroup) the running maximum as read when the step began
    const int slot = blockIdx.x >> 3;
    const int p = slot % P.n_prn, fg = (int)(blockIdx.x & 7) + 8 * (slot / P.n_prn);
    if (fg * B >= P.n_freq) return;            // the whole workgroup: the grid is padded to a multiple of eight groups
    // one transform per workgroup (N = 4096): b is 0 by construction -- said so explicitly, the spectra's addresses are
    // then workgroup-uniform (scalar base + one per-thread offset instead of a 64-bit address pair per load)
    const int tid = threadIdx.x, b = (B == 1) ? 0 : tid / TF, jl0 = (B == 1) ? tid : tid % TF;
    const int jl = XP ? X4096::jl1(tid) : jl0;   // butterfly of the later passes = the bins jl + TF s held at the end
    const int f = fg * B + b;
    const bool live = f < P.n_freq;
    // 32 KB per PRN, re-read every step out of L2.  Keeping it in registers together with a prefetch of the next
    // step's data spectrum needs 228 VGPRs = two workgroups per CU: 0.245 ms per search against 0.236 this way
    // (2 272 workgroups are 2.96 rounds of 768 slots, but 4.44 rounds of 512)
    const cf* c = cspec + (size_t)p * N;
    c2 tw[3][15];
#pragma unroll
    for (int k = 0; k < 15; ++k) tw[0][k] = tw[1][k] = tw[2][k] = make_c2(1.f, 0.f);
    if constexpr (fft_npass(N) > 1) load_twiddles<N, 1>(tw[1], twtab, jl);
    if constexpr (fft_npass(N) > 2) load_twiddles<N, 2>(tw[2], twtab, jl);
    double acc[8];
#pragma unroll
    for (int s = 0; s < 8; ++s) acc[s] = 0.0;
    // every index outside the +-2-chip zone around the peak counts towards the mean: nsamp - (4 nsampchip + 1) of them,
    // wrapped or not (launch_acq_search checks 4 nsampchip < nsamp)
    const int cnt = P.nsamp - 4 * nsampchip - 1;
    // (`seen` lives across the steps on purpose: declared inside the loop the same code compiles to a kernel that is 11 %
    // slower -- 0.196 against 0.176 ms per search, profiles/r04_ab_acq.txt; the register allocation decides, not the idea)
    unsigned long long seen = 0ull;
    for (int step = 0; step < P.intg; ++step) {
        c2 v[16];
        // the running maximum of this (PRN, step) as it stands now, read past the L1 by the transform's first thread; used
        // after the transform, so its latency hides behind the spectra's loads.  Possibly stale by then -- gmax only grows
        if (jl == 0 && live) seen = __hip_atomic_load(&gmax[(size_t)p * P.intg + step], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned long long seen_now = seen;
        const cf* x = xspec + ((size_t)step * P.n_freq + (live ? f : 0)) * N;
        // ALL 32 loads of the step first, the products behind them: left to itself the scheduler sometimes issues the loads
        // two at a time with a full wait in between (seen in the ISA of an otherwise identical source: 0.196 against
        // 0.176 ms per search).  The scheduling barrier pins the order; the products then consume the loads as they land.
        cf xa[16], qa[16];
#pragma unroll
        for (int s = 0; s < 16; ++s) xa[s] = x[jl0 + TF * s];
#pragma unroll
        for (int s = 0; s < 16; ++s) qa[s] = c[jl0 + TF * s];
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            const cf a = xa[s], q = qa[s];
            // conj(X) * (-C / m) as stored = cpxconv's product, as in acq_inv_kernel.  Plain arithmetic on purpose: the
            // packed form (v_pk_mul + v_pk_fma per point, alone or two points interleaved) measured 18-19 % SLOWER here,
            // and so did every attempt to hold part of the next step's spectrum in registers or the running power in LDS
            // (profiles/r04_ab_acq.txt): the kernel sits at 168 VGPRs with three workgroups per CU and nothing to spare
            v[s] = make_c2(a.x * q.x - a.y * q.y, a.x * q.y + a.y * q.x);
        }
        acq_transform<N>(v, lds, tid, b * lds_span(N), jl, tw);
        AcqMax mine{-1.0};
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            acc[s] += (double)(v[s].x * v[s].x + v[s].y * v[s].y);   // 1/m^2 rides on the code spectrum (acq_code_body)
            mine.v = __builtin_fmax(mine.v, acc[s]);
        }
#ifdef GJ_ACQ_ABLATE_CHECK   // timing only (tools/ab_build.sh): no peak bookkeeping at all
        if (step + 1 == P.intg && live && jl == 0) rows[((size_t)p * P.intg + step) * P.n_freq + f].maxv = mine.v;
        continue;
#endif
        // The row's maximum (value only) AND what the transform's first thread read of the running maximum when the step
        // began, both known to every thread of the transform after ONE exchange.  The early read may be stale, but gmax
        // only grows: a row below even the stale value cannot be the step's winner and goes straight on -- no atomic,
        // no second barrier (N = 4096; the shapes with several transforms per workgroup agree on the branch through LDS).
        const int par = step & 1;
        bool mine_maybe, maybe;
        AcqMax rowmax;
        unsigned long long bits;
        if constexpr (B == 1) {
            if (tid == 0) sh_seen[par] = seen_now;   // published by the barrier inside the reduction
            rowmax = acq_group_reduce<N>(mine, [](AcqMax a, AcqMax o) { return AcqMax{__builtin_fmax(a.v, o.v)}; }, shm, b);
            bits = (unsigned long long)__double_as_longlong(rowmax.v);
            mine_maybe = bits >= sh_seen[par];       // live: a workgroup with one transform has returned already otherwise
            maybe = mine_maybe;
        } else {
            rowmax = acq_group_reduce<N>(mine, [](AcqMax a, AcqMax o) { return AcqMax{__builtin_fmax(a.v, o.v)}; }, shm, b);
            bits = (unsigned long long)__double_as_longlong(rowmax.v);
            if (jl == 0) sh_need[par][b] = (live && bits >= seen_now) ? 1 : 0;
            __syncthreads();
            mine_maybe = sh_need[par][b] != 0;
            maybe = false;
#pragma unroll
            for (int k = 0; k < B; ++k) maybe |= sh_need[par][k] != 0;
            __syncthreads();                         // sh_need[par] is written again below
        }
        if (!mine_maybe && live && jl == 0) {
            AcqRow r;
            r.maxv = rowmax.v;
            r.max2 = 0.0;
            r.sum = 0.0;
            r.argk = 0;
            r.cnt = 0;
            rows[((size_t)p * P.intg + step) * P.n_freq + f] = r;
        }
        if (!maybe) continue;
        // may be the winner: offer the maximum; only a row that raises or equals the running maximum does the bookkeeping
        if (jl == 0) {
            int need = 0;
            if (mine_maybe) {
                const unsigned long long old = atomicMax(&gmax[(size_t)p * P.intg + step], bits);
                need = bits >= old;
                if (!need) {
                    AcqRow r;
                    r.maxv = rowmax.v;
                    r.max2 = 0.0;
                    r.sum = 0.0;
                    r.argk = 0;
                    r.cnt = 0;
                    rows[((size_t)p * P.intg + step) * P.n_freq + f] = r;
                }
            }
            sh_need[par][b] = need;
        }
        __syncthreads();
        bool any = false;
#pragma unroll
        for (int k = 0; k < B; ++k) any |= sh_need[par][k] != 0;    // workgroup-uniform: the reductions below hold barriers
        if (!any) continue;
        const bool need = sh_need[par][b] != 0;
        // ---- the full bookkeeping, for the few rows that may hold the step's peak (as in rounds 2-3) ----
        AcqBestS best;
        best.val = -1.0;
        best.idx = 0x7fffffff;
#pragma unroll
        for (int s = 0; s < 8; ++s)
            if (acc[s] > best.val) { best.val = acc[s]; best.idx = jl + TF * s; }   // indices rise with s: '>' keeps the first
        best = acq_group_reduce<N>(best, [](AcqBestS a, AcqBestS o) {
            return (o.val > a.val || (o.val == a.val && o.idx < a.idx)) ? o : a; }, shb, b);
        int exinds = best.idx - 2 * nsampchip, exinde = best.idx + 2 * nsampchip;
        if (exinds < 0) exinds += P.nsamp;
        if (exinde >= P.nsamp) exinde -= P.nsamp;
        AcqTail t{0.0, (jl == 0) ? acc[0] : -1.0};   // maxvd seeds with data[0] whatever the exclusion zone says
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            const int i = jl + TF * s;
            const bool kept = (exinds <= exinde) ? (i < exinds || i > exinde) : (i < exinds && i > exinde);
            if (kept) {
                t.sum += acc[s];
                if (i >= 1) t.mx2 = acc[s] > t.mx2 ? acc[s] : t.mx2;
            }
        }
        t = acq_group_reduce<N>(t, [](AcqTail a, AcqTail o) { return AcqTail{a.sum + o.sum, a.mx2 > o.mx2 ? a.mx2 : o.mx2}; }, sht, b);
        if (live && need && jl == 0) {
            AcqRow r;
            r.maxv = best.val;
            r.max2 = t.mx2;
            r.sum = t.sum;
            r.argk = best.idx;
            r.cnt = cnt;
            rows[((size_t)p * P.intg + step) * P.n_freq + f] = r;
        }
    }
}

// the reference's loop over the integration steps (sdracq.c:15-28), replayed on the row records: wave w finds the
// winning Doppler row of steps w, w + 16, ... (their loads are in flight together), then one thread walks the steps
// in order and stops at the first that passes the peak test
__global__ __launch_bounds__(1024) void acq_summary_kernel(AcqParams P, double ctime, float threshold,
                                                           const AcqRow* __restrict__ rows, gj_acq_result* __restrict__ out) {
    __shared__ AcqRow win[64];     // intg <= 64 (launch_acq_search)
    __shared__ int winf[64];
    const int p = blockIdx.x, lane = threadIdx.x & 63;
    for (int step = threadIdx.x >> 6; step < P.intg; step += blockDim.x >> 6) {
        const AcqRow* rr = rows + ((size_t)p * P.intg + step) * P.n_freq;
        AcqBest b{-1.0, 0x7fffffff};   // idx = Doppler row; equal maxima: the smaller flat index = the smaller row
        for (int f = lane; f < P.n_freq; f += 64) b = acq_better(b, AcqBest{rr[f].maxv, f});
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            AcqBest o;
            o.val = __shfl_xor(b.val, off, 64);
            o.idx = __shfl_xor(b.idx, off, 64);
            b = acq_better(b, o);
        }
        if (b.idx >= P.n_freq) b.idx = 0;   // only if every row maximum were NaN
        if (lane == 0) {
            win[step] = rr[b.idx];
            winf[step] = b.idx;
        }
    }
    __syncthreads();
    if (threadIdx.x != 0) return;
    for (int step = 0; step < P.intg; ++step) {
        const AcqRow w = win[step];
        const double meanP = w.sum / (double)w.cnt;
        const double peakr = w.maxv / w.max2;
        const bool acquired = peakr > (double)threshold;
        if (acquired || step + 1 == P.intg) {
            gj_acq_result r;
            r.max_power = w.maxv;
            r.second_power = w.max2;
            r.mean_power = meanP;
            r.peak_ratio = peakr;
            r.cn0 = 10.0 * log10(w.maxv / meanP / ctime);
            r.code_index = w.argk;
            r.freq_index = winf[step];
            r.steps = step + 1;
            r.acquired = acquired ? 1 : 0;
            out[p] = r;
            return;
        }
    }
}

size_t acq_workspace(int nsamp, int n_freq, int n_prn, int intg, bool own_power) {
    const size_t nfft = 2 * (size_t)nsamp;
    size_t b = 256;                                                        // done flags
    b += align_up((size_t)n_prn * sizeof(int), 256);
    b += align_up((size_t)n_prn * nfft * sizeof(cf), 256);                 // code spectra
    b += align_up((size_t)intg * n_freq * nfft * sizeof(cf), 256);         // data spectra
    // no power array requested: the single-launch form keeps P in registers and needs the row records only
    if (own_power) b += align_up((size_t)n_prn * intg * n_freq * sizeof(AcqRow), 256) +
                        align_up((size_t)n_prn * intg * sizeof(unsigned long long), 256);   // row records + running maxima
    return b;
}

template <int N>
static int acq_run(gj_ctx* ctx, const AcqParams& P, const short* d_codes, const uint8_t* d_phase, int nsampchip,
                   double ctime, float threshold, gj_acq_result* d_out, double* d_power) {
    constexpr int B = kBlockPoints / N;
    unsigned char* w = ctx->ws;
    int* done = reinterpret_cast<int*>(w);
    w += 256 + align_up((size_t)P.n_prn * sizeof(int), 256);
    cf* cspec = reinterpret_cast<cf*>(w);
    w += align_up((size_t)P.n_prn * N * sizeof(cf), 256);
    cf* xspec = reinterpret_cast<cf*>(w);
    w += align_up((size_t)P.intg * P.n_freq * N * sizeof(cf), 256);
    AcqRow* rows = reinterpret_cast<AcqRow*>(w);
    unsigned long long* gmax = reinterpret_cast<unsigned long long*>(w + align_up((size_t)P.n_prn * P.intg * P.n_freq * sizeof(AcqRow), 256));
    const int code_blocks = (P.n_prn + B - 1) / B, fwd_blocks = (P.intg * P.n_freq + B - 1) / B;
    hipLaunchKernelGGL((acq_prep_kernel<N>), dim3((unsigned)(code_blocks + fwd_blocks)), dim3(kBlockThreads), 0, ctx->stream, P,
                       d_codes, d_phase, ctx->d_twiddle, cspec, xspec, d_power ? nullptr : gmax, d_power ? 0 : P.n_prn * P.intg,
                       code_blocks);
    GJ_LAUNCH_CHECK(ctx);
    if (!d_power) {
        const unsigned groups = (unsigned)((P.n_freq + B - 1) / B);
        hipLaunchKernelGGL((acq_inv_all_kernel<N>), dim3(8u * (unsigned)P.n_prn * ((groups + 7u) / 8u)), dim3(kBlockThreads), 0,
                           ctx->stream, P, nsampchip, ctx->d_twiddle, xspec, cspec, rows, gmax);
        GJ_LAUNCH_CHECK(ctx);
        const unsigned waves = (unsigned)(P.intg < 16 ? P.intg : 16);
        hipLaunchKernelGGL(acq_summary_kernel, dim3((unsigned)P.n_prn), dim3(64u * waves), 0, ctx->stream, P, ctime, threshold,
                           rows, d_out);   // writes every PRN's record
        GJ_LAUNCH_CHECK(ctx);
        return GJ_OK;
    }
    // the caller wants the power array: step by step, P in memory, a PRN's rows frozen when it acquires
    double* power = d_power;
    GJ_HIP(ctx, hipMemsetAsync(d_out, 0, (size_t)P.n_prn * sizeof(gj_acq_result), ctx->stream));
    GJ_HIP(ctx, hipMemsetAsync(done, 0, (size_t)P.n_prn * sizeof(int), ctx->stream));
    GJ_HIP(ctx, hipMemsetAsync(power, 0, (size_t)P.n_prn * P.n_freq * P.nsamp * sizeof(double), ctx->stream));   // calloc, sdrmain.c:346
    for (int step = 0; step < P.intg; ++step) {
        hipLaunchKernelGGL((acq_inv_kernel<N>), dim3((unsigned)((P.n_freq + B - 1) / B), (unsigned)P.n_prn),
                           dim3(kBlockThreads), 0, ctx->stream, P, step, done, ctx->d_twiddle, xspec, cspec, power);
        GJ_LAUNCH_CHECK(ctx);
        hipLaunchKernelGGL(acq_check_kernel, dim3((unsigned)P.n_prn), dim3(1024), 0, ctx->stream, P, step, nsampchip,
                           ctime, threshold, power, done, d_out);
        GJ_LAUNCH_CHECK(ctx);
    }
    return GJ_OK;
}

int launch_acq_search(gj_ctx* ctx, const uint8_t* d_iq, size_t nbytes, size_t first_sample, int nsamp, int intg,
                      const int16_t* d_codes, int n_prn, const uint8_t* d_phase, int n_freq, int nsampchip, double ctime,
                      float threshold, gj_acq_result* d_out, double* d_power) {
    if (nsamp != 2048 && nsamp != 1024 && nsamp != 512) return fail(ctx, GJ_ERR_UNSUPPORTED, "nsamp must be 512, 1024 or 2048");
    if (intg < 1 || intg > 64 || n_prn < 1 || n_prn > 4096 || n_freq < 1 || n_freq > 4096)
        return fail(ctx, GJ_ERR_INVALID, "intg 1..64, n_prn and n_freq 1..4096");
    if (nsampchip < 0 || 4 * nsampchip >= nsamp || !(ctime > 0.0)) return fail(ctx, GJ_ERR_INVALID, "bad nsampchip / ctime");
    if ((reinterpret_cast<uintptr_t>(d_iq) & 1) != 0) return fail(ctx, GJ_ERR_INVALID, "capture must be 2-byte aligned");
    // step s reads samples [first + s nsamp, first + s nsamp + 2 nsamp)  (rcvgetbuff of 2*nsamp, then += nsamp)
    if (first_sample > nbytes / 2) return fail(ctx, GJ_ERR_INVALID, "first_sample %zu is past the capture's %zu samples", first_sample, nbytes / 2);
    const size_t need = first_sample + (size_t)(intg + 1) * nsamp;   // cannot wrap: first_sample <= nbytes / 2
    if (need > nbytes / 2) return fail(ctx, GJ_ERR_INVALID, "search needs samples up to %zu, capture has %zu", need, nbytes / 2);
    int rc = ensure_workspace(ctx, acq_workspace(nsamp, n_freq, n_prn, intg, d_power == nullptr));
    if (rc) return rc;
    AcqParams P;
    P.iq = d_iq;
    P.first_sample = first_sample;
    P.nsamp = nsamp;
    P.nfft = 2 * nsamp;
    P.n_freq = n_freq;
    P.n_prn = n_prn;
    P.intg = intg;
    P.offset = 128;
    P.scale = (float)((1.0 / 32.0) / (double)P.nfft);   // CSCALE / m (sdrcmn.c:7,766)
    P.inv_m2 = 1.0f / ((float)P.nfft * (float)P.nfft);
    const short* codes = reinterpret_cast<const short*>(d_codes);
    switch (P.nfft) {
        case 4096: return acq_run<4096>(ctx, P, codes, d_phase, nsampchip, ctime, threshold, d_out, d_power);
        case 2048: return acq_run<2048>(ctx, P, codes, d_phase, nsampchip, ctime, threshold, d_out, d_power);
        default: return acq_run<1024>(ctx, P, codes, d_phase, nsampchip, ctime, threshold, d_out, d_power);
    }
}

}   // namespace gj
