// CPU emulation of the block-FFT schedule in csrc/fft_core.h: the same header, the same
// per-thread register arrays, the same LDS scatter/gather index maps, executed one
// "thread" at a time with an array standing in for LDS.  Checks every supported size
// against a double-precision O(N^2) DFT.  Built and run by tests/test_fft_core_host.py.
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "fft_core.h"

using namespace gj;

static std::vector<cf> g_table;

template <int N, int PASS, bool TWO = false, bool FMA = false>
static void run_passes(std::vector<cf (*)[16]>& regs, std::vector<cf>& lds) {
    constexpr int TF = N / 16, NP = fft_npass(N);
    for (int j = 0; j < kBlockThreads; ++j) {
        const int b = j / TF, jl = j % TF;
        c2 tw[15];
        for (auto& t : tw) t = make_c2(0.f, 0.f);
        constexpr bool two = TWO && PASS > 0 && fft_radix(N, PASS) == 16;
        if constexpr (two) load_twiddles6<N, PASS>(tw, g_table.data(), jl);
        else if constexpr (PASS > 0) load_twiddles<N, PASS>(tw, g_table.data(), jl);
        fft_pass<N, PASS, TWO, FMA>(*regs[j], tw, inner_twiddles());
        if constexpr (PASS + 1 < NP) lds_scatter<N, PASS>(*regs[j], lds.data(), b * lds_span(N), jl);
    }
    if constexpr (PASS + 1 < NP) {
        for (int j = 0; j < kBlockThreads; ++j) {
            const int b = j / TF, jl = j % TF;
            lds_gather<N>(*regs[j], lds.data(), b * lds_span(N), jl);
        }
        run_passes<N, PASS + 1, TWO, FMA>(regs, lds);
    }
}

template <int N, bool TWO = false, bool FMA = false>
static double check() {
    constexpr int TF = N / 16, B = kBlockPoints / N;
    std::vector<cf> in(kBlockPoints), lds(kBlockPoints + kBlockPoints / 16 + 64);
    for (auto& v : in) v = cf{(float)(rand() % 511 - 255), (float)(rand() % 511 - 255)};
    std::vector<cf> storage(kBlockThreads * 16);
    std::vector<cf (*)[16]> regs(kBlockThreads);
    for (int j = 0; j < kBlockThreads; ++j) {
        regs[j] = reinterpret_cast<cf (*)[16]>(&storage[j * 16]);
        const int b = j / TF, jl = j % TF;
        for (int s = 0; s < 16; ++s) (*regs[j])[s] = in[b * N + jl + TF * s];
    }
    run_passes<N, 0, TWO, FMA>(regs, lds);
    double worst = 0.0;
    for (int b = 0; b < B; ++b) {
        double norm = 0.0;
        std::vector<double> re(N), im(N);
        for (int k = 0; k < N; ++k) {
            double sr = 0, si = 0;
            for (int n = 0; n < N; ++n) {
                const double ang = -2.0 * M_PI * (double)((long long)k * n % N) / N;
                const double c = cos(ang), s = sin(ang);
                sr += in[b * N + n].x * c - in[b * N + n].y * s;
                si += in[b * N + n].x * s + in[b * N + n].y * c;
            }
            re[k] = sr; im[k] = si;
            norm += sr * sr + si * si;
        }
        norm = sqrt(norm / N);
        for (int j = b * TF; j < (b + 1) * TF; ++j)
            for (int s = 0; s < 16; ++s) {
                const int k = (j % TF) + TF * s;
                const double dr = (*regs[j])[s].x - re[k], di = (*regs[j])[s].y - im[k];
                const double e = sqrt(dr * dr + di * di) / norm;
                if (e > worst) worst = e;
            }
    }
    return worst;
}

// the conflict-free N = 4096 schedule (thread role changes at the first exchange)
template <bool FMA>
static double check_x4096() {
    constexpr int N = 4096;
    std::vector<cf> in(N), lds(X4096::kSpan);
    for (auto& v : in) v = cf{(float)(rand() % 511 - 255), (float)(rand() % 511 - 255)};
    // the two layouts must be injective and fit the buffer
    for (int ex = 0; ex < 2; ++ex) {
        std::vector<int> seen(X4096::kSpan, 0);
        for (int i = 0; i < N; ++i) {
            const int sl = ex ? X4096::slot1(i) : X4096::slot0(i);
            if (sl < 0 || sl >= X4096::kSpan || seen[sl]++) return 1.0;
        }
    }
    std::vector<cf> storage(kBlockThreads * 16);
    auto regs = [&](int j) -> cf (&)[16] { return *reinterpret_cast<cf (*)[16]>(&storage[j * 16]); };
    const InnerTw k = inner_twiddles();
    c2 none[15];
    for (int j = 0; j < 256; ++j) {
        for (int s = 0; s < 16; ++s) regs(j)[s] = in[j + 256 * s];
        fft_pass<N, 0, false, FMA>(regs(j), none, k);
        // the scatter must put leg t at the layout slot of its logical index
        for (int t = 0; t < 16; ++t)
            if (286 * (j >> 4) + 17 * (j & 15) + X4096::c0(t) != X4096::slot0(out_index<N, 0>(j, 0, t))) return 2.0;
        x4096_scatter<0>(regs(j), lds.data(), j);
    }
    for (int pass = 1; pass <= 2; ++pass) {
        for (int j = 0; j < 256; ++j) {
            if (pass == 1) x4096_gather<0>(regs(j), lds.data(), j);
            else x4096_gather<1>(regs(j), lds.data(), j);
        }
        for (int j = 0; j < 256; ++j) {
            const int jl = X4096::jl1(j);
            c2 tw[15];
            if (pass == 1) {
                load_twiddles<N, 1>(tw, g_table.data(), jl);
                fft_pass<N, 1, false, FMA>(regs(j), tw, k);
                for (int t = 0; t < 16; ++t)
                    if (287 * (j & 15) + (j >> 4) + 18 * t != X4096::slot1(out_index<N, 1>(jl, 0, t))) return 3.0;
                x4096_scatter<1>(regs(j), lds.data(), j);
            } else {
                load_twiddles<N, 2>(tw, g_table.data(), jl);
                fft_pass<N, 2, false, FMA>(regs(j), tw, k);
            }
        }
    }
    double worst = 0.0, norm = 0.0;
    std::vector<double> re(N), im(N);
    for (int kk = 0; kk < N; ++kk) {
        double sr = 0, si = 0;
        for (int n = 0; n < N; ++n) {
            const double ang = -2.0 * M_PI * (double)((long long)kk * n % N) / N;
            sr += in[n].x * cos(ang) - in[n].y * sin(ang);
            si += in[n].x * sin(ang) + in[n].y * cos(ang);
        }
        re[kk] = sr; im[kk] = si;
        norm += sr * sr + si * si;
    }
    norm = sqrt(norm / N);
    for (int j = 0; j < 256; ++j)
        for (int s = 0; s < 16; ++s) {
            const int kk = X4096::jl1(j) + 256 * s;
            const double dr = regs(j)[s].x - re[kk], di = regs(j)[s].y - im[kk];
            worst = std::max(worst, sqrt(dr * dr + di * di) / norm);
        }
    return worst;
}

int main() {
    g_table.resize(kTwiddleTable);
    for (int m = 0; m < kTwiddleTable; ++m) {
        const double a = -2.0 * M_PI * m / kTwiddleTable;
        g_table[m] = cf{(float)cos(a), (float)sin(a)};
    }
    srand(7);
    int bad = 0;
#define CHECK(N)                                                    \
    {                                                               \
        const double e = check<N>();                                \
        printf("N=%5d passes=%d rel_err=%.3e\n", N, fft_npass(N), e); \
        if (!(e < 2e-6)) ++bad;                                     \
    }
    {
        const double e1 = check<4096, true>(), e2 = check<1024, true>(), e3 = check<256, true>();
        printf("two-step twiddles: N=4096 %.3e  N=1024 %.3e  N=256 %.3e\n", e1, e2, e3);
        if (!(e1 < 2e-6 && e2 < 2e-6 && e3 < 2e-6)) ++bad;
    }
    {
        const double e1 = check<4096, false, true>(), e2 = check<1024, false, true>(), e3 = check<16, false, true>();
        printf("FMA-form butterflies: N=4096 %.3e  N=1024 %.3e  N=16 %.3e\n", e1, e2, e3);
        if (!(e1 < 2e-6 && e2 < 2e-6 && e3 < 2e-6)) ++bad;
    }
    {
        const double e1 = check_x4096<false>(), e2 = check_x4096<true>();
        printf("conflict-free N=4096 schedule: %.3e  (FMA form %.3e)\n", e1, e2);
        if (!(e1 < 2e-6 && e2 < 2e-6)) ++bad;
    }
    CHECK(16) CHECK(32) CHECK(64) CHECK(128) CHECK(256) CHECK(512) CHECK(1024) CHECK(2048) CHECK(4096)
    printf(bad ? "FAIL\n" : "OK\n");
    return bad;
}
