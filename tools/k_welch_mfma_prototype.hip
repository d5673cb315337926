// MEASURED AND REJECTED (round 3): welch_kernel<4096> with pass 0 on the matrix pipe.  A reading copy, not part of the
// build: it was compiled inside csrc/k_welch.hip (after welch_kernel, selected with GPSJAM_WELCH_MFMA=1 in welch_range)
// together with welch_mfma.h (now tools/welch_mfma.h; checked on the GPU by tools/mfma_pass0_check.hip).
// Result (profiles/r03_ab_mfma_pass0.txt): PSD equal to welch_kernel's to 4.8e-7 relative on the 1-GiB capture, 159
// VGPRs, three workgroups per CU -- and 1.242-1.249 ms against 1.128 ms (+10 %).  With four of the eight
// v_mfma_f32_32x32x16_f16 removed: 1.178 ms.  The matrix instructions cost their full 32 cycles each on top of the
// packed-f32 work of the SIMD's other waves (8 x 32 = 256 of 2463 cycles per wave-step = 10.4 %): packed f32 and MFMA do
// not overlap on gfx950, so the idle matrix pipe is not free capacity for this kernel.  Without them the new front end,
// writer-side twiddle and three-tap window run level with the shipped kernel (about 1.115 against 1.128 ms).
// ---------------------------------------------------------------------------------------------------------------
// N = 4096 with pass 0 on the matrix pipe (welch_mfma.h).  Differences from welch_kernel<4096>:
//  * the raw samples go to the MFMA as fp16 integers u - 128, unwindowed; per-segment mean removal makes the offset
//    immaterial (the detrend term below is formed from the same u - 128), and the scale 2 / (2 w) cancels;
//  * the FULL inter-pass twiddle W_4096^(n1 k2) is applied by the WRITER of the first exchange (thread n1 = tid holds
//    all sixteen k2), where the periodic Hann window w[n] = 1/2 - 1/4 e^(i th n) - 1/4 e^(-i th n), n = n1 + 256 n2,
//    is the real three-tap  V[k] = U[k] - (U[k-1] + U[k+1]) / 2  over the thread's own U[k] = T[k] Z[k] (= 2 w applied),
//    with one extra factor c = e^(16 i th n1) on the two wrap-around terms;
//  * pass 1 therefore runs without input twiddles, pass 2 with W_256^(s t1) in place of W_4096^(s jl).
// Everything else -- exchange schedule, detrend bins, |X|^2 accumulation, partial-spectrum layout -- is welch_kernel's.
__global__ __launch_bounds__(kBlockThreads, 3) void welch_mfma_kernel(const uint8_t* __restrict__ iq, WelchGeom g,
                                                                      const cf* __restrict__ twtab,
                                                                      float* __restrict__ partial, unsigned wg_base) {
    constexpr int N = 4096, TF = 256;
    __shared__ cf lds0[X4096::kSpan];
    __shared__ float wsum[2][4][2];   // [step parity][wave][(sum I, sum Q) of u - 128 over the wave's 1024 samples]
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int jl = X4096::jl1(tid);   // butterfly of passes 1 and 2 = bins held at the end
    const unsigned wg = blockIdx.x + wg_base;
    const unsigned c = wg / g.splits, part = wg % g.splits;
    const unsigned nseg = (c + 1 == g.nchunks) ? g.nseg_last : g.nseg_full;
    const unsigned seg_lo = (unsigned)((unsigned long long)part * nseg / g.splits);
    const unsigned seg_hi = (unsigned)((unsigned long long)(part + 1) * nseg / g.splits);
    const InnerTw ktw = inner_twiddles();
    const MfmaDft16 A = mfma_dft16_matrix(lane);
    c2 T[15], tw2[15];
#pragma unroll
    for (int k = 1; k < 16; ++k) {
        T[k - 1] = to_c2(twtab[(tid * k) & (kTwiddleTable - 1)]);
        tw2[k - 1] = to_c2(twtab[(16 * k * (jl >> 4)) & (kTwiddleTable - 1)]);
    }
    const c2 cw = to_c2(twtab[(kTwiddleTable - 16 * tid) & (kTwiddleTable - 1)]);   // e^(+2 pi i 16 n1 / 4096)
    float accs[16];
#pragma unroll
    for (int s = 0; s < 16; ++s) accs[s] = 0.f;

    const uint8_t* chunk8 = iq + (size_t)c * g.chunk_samples * 2;
    // lane (H = lane / 32, col = lane % 32): samples n1 = 64 wave + 32 grp + col, n2 = 8 H + j of the segment
    const unsigned lane_byte = 2u * (unsigned)(64 * wave + (lane & 31)) + 4096u * (unsigned)(lane >> 5);
    auto load_step = [&](unsigned (&dst)[2][8], unsigned seg_idx) {
        const unsigned byte0 = seg_idx * (unsigned)N + lane_byte;   // N / 2 samples per hop = N bytes
#pragma unroll
        for (int grp = 0; grp < 2; ++grp)
#pragma unroll
            for (int j = 0; j < 8; ++j)
                dst[grp][j] = *reinterpret_cast<const uint16_t*>(chunk8 + (byte0 + 64u * grp + 512u * j));
    };
    const unsigned nsteps = seg_hi - seg_lo;
    unsigned raw[2][8];
    if (nsteps) load_step(raw, seg_lo);
    for (unsigned it = 0; it < nsteps; ++it) {
        const unsigned seg = seg_lo + it;
        const unsigned seg_next = (seg + 1 < seg_hi) ? seg + 1 : seg_lo;
        half2v x[2][8];
        half2v hs = {(_Float16)0.0f, (_Float16)0.0f};   // (sum I, sum Q) of u - 128 over the lane's 16 samples: |.| <= 2048, exact
#pragma unroll
        for (int grp = 0; grp < 2; ++grp)
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                x[grp][j] = unpack_f16(raw[grp][j]);
                hs += x[grp][j];
            }
        // wave totals: both sums biased to 0 .. 4096 ride in one register through the in-row steps
        float si, sq;
        wave_sum_pair_u16((float)hs[0] + 2048.0f, (float)hs[1] + 2048.0f, si, sq);
        const unsigned cur = it & 1;
        if (lane == 0) {
            wsum[cur][wave][0] = si - 64.0f * 2048.0f;
            wsum[cur][wave][1] = sq - 64.0f * 2048.0f;
        }
        c2 v[16];
        pass0_mfma(x, A, v);
        // full twiddle, then the window as a three-tap over k2 (times 2)
        c2 u[16];
        u[0] = v[0];
#pragma unroll
        for (int k = 1; k < 16; ++k) u[k] = cmul(v[k], T[k - 1]);
        const c2 wrap_lo = cmul(u[15], cw);                             // c U[15]
        const c2 wrap_hi = cmul(u[0], make_c2(cw.x, -cw.y));            // conj(c) U[0]
        const c2 mhalf = make_c2(-0.5f, -0.5f);
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const c2 lo = (k == 0) ? wrap_lo : u[k - 1];
            const c2 hi = (k == 15) ? wrap_hi : u[k + 1];
            v[k] = fma2(cadd(lo, hi), mhalf, u[k]);
        }
        x4096_scatter<0>(v, lds0, tid);
        load_step(raw, seg_next);   // next step's samples, asked for while this step's points sit in LDS
        __syncthreads();
        x4096_gather<0>(v, lds0, tid);
        __syncthreads();
        dft16_fma(v, ktw);
        x4096_scatter<1>(v, lds0, tid);
        __syncthreads();
        x4096_gather<1>(v, lds0, tid);
        __syncthreads();
        dft16_fma_tw(v, tw2, ktw);

        // detrend in the frequency domain on bins 0, 1, N-1: FFT(2 w (d - m)) = FFT(2 w d) - m 2 W, 2 W = N at bin 0, -N/2 at +-1
        if (jl <= 1 || jl == TF - 1) {
            const float Sx = (wsum[cur][0][0] + wsum[cur][1][0]) + (wsum[cur][2][0] + wsum[cur][3][0]);
            const float Sy = (wsum[cur][0][1] + wsum[cur][1][1]) + (wsum[cur][2][1] + wsum[cur][3][1]);
            if (jl == WelchBins<N>::jl(0)) {
                v[WelchBins<N>::slot(0)].x -= Sx;
                v[WelchBins<N>::slot(0)].y -= Sy;
            }
            if (jl == WelchBins<N>::jl(1)) {
                v[WelchBins<N>::slot(1)].x += 0.5f * Sx;
                v[WelchBins<N>::slot(1)].y += 0.5f * Sy;
            }
            if (jl == WelchBins<N>::jl(N - 1)) {
                v[WelchBins<N>::slot(N - 1)].x += 0.5f * Sx;
                v[WelchBins<N>::slot(N - 1)].y += 0.5f * Sy;
            }
        }
#pragma unroll
        for (int s = 0; s < 16; ++s) accs[s] = fmaf(v[s].x, v[s].x, fmaf(v[s].y, v[s].y, accs[s]));
    }
    float* out = partial + (size_t)wg * N + jl;
#pragma unroll
    for (int s = 0; s < 16; ++s) out[TF * s] = accs[s];
}

