/* A plain C11 consumer of include/gpsjam.h (the reference's backend, gnssdec, is C): links
 * libgpsjam_hip.so, runs K1, K3 and K4 on a small deterministic capture through the host-buffer
 * entry points and checks them against the same arithmetic done here in integers/doubles.
 * Built and run by tests/test_c_abi_gpu.py (gcc, no HIP headers needed). */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "gpsjam.h"

#define CHECK(call)                                                                      \
    do {                                                                                 \
        int rc_ = (call);                                                                \
        if (rc_ != GJ_OK) {                                                              \
            fprintf(stderr, "%s -> %d %s: %s\n", #call, rc_, gj_strerror(rc_), gj_last_error(ctx)); \
            return 2;                                                                    \
        }                                                                                \
    } while (0)

int main(void) {
    gj_ctx* ctx = NULL;
    if (gj_version() != GJ_VERSION) { fprintf(stderr, "version mismatch\n"); return 2; }
    CHECK(gj_create(0, &ctx));

    /* quiet floor, then a loud burst from sample 260000 on (xorshift noise, no libc rand) */
    const size_t nsamp = 400000, nbytes = 2 * nsamp, chunk_bytes = 65536;
    uint8_t* iq = (uint8_t*)malloc(nbytes);
    uint32_t x = 2463534242u;
    for (size_t i = 0; i < nbytes; ++i) {
        x ^= x << 13; x ^= x >> 17; x ^= x << 5;
        const int amp = (i / 2 >= 260000) ? 100 : 9;
        iq[i] = (uint8_t)(128 + (int)(x % (2u * (unsigned)amp + 1u)) - amp);
    }

    /* K1 */
    const size_t nchunks = gj_chunk_count(nbytes, chunk_bytes);
    float* power = (float*)malloc(nchunks * sizeof(float));
    size_t n_out = 0;
    float ms = 0.f;
    CHECK(gj_chunk_power_u8(ctx, iq, nbytes, chunk_bytes, 1e-10f, 0, power, nchunks, &n_out, &ms));
    if (n_out != nchunks) { fprintf(stderr, "chunk count %zu != %zu\n", n_out, nchunks); return 1; }
    for (size_t c = 0; c < nchunks; ++c) {
        const size_t off = c * chunk_bytes, len = (nbytes - off < chunk_bytes) ? nbytes - off : chunk_bytes;
        long long s = 0;
        for (size_t i = 0; i < (len & ~(size_t)1); ++i) { const long long v = 2 * (long long)iq[off + i] - 255; s += v * v; }
        const float want = (float)((double)s / (4.0 * (double)(len / 2))) + 1e-10f;
        if (fabsf(power[c] - want) > 1e-6f * want) { fprintf(stderr, "K1 chunk %zu: %g vs %g\n", c, power[c], want); return 1; }
    }

    /* K3, threshold 0.5: first sample whose amplitude exceeds it, mean amplitude from there on */
    gj_amp_stats st;
    CHECK(gj_amp_stats_u8(ctx, iq, nbytes, 0.5f, &st, &ms));
    long long first = -1;
    double sum = 0.0;
    for (size_t n = 0; n < nsamp; ++n) {
        const float re = ((float)iq[2 * n] - 127.5f) / 127.5f, im = ((float)iq[2 * n + 1] - 127.5f) / 127.5f;
        const float a = sqrtf(re * re + im * im);
        if (first < 0 && a > 0.5f) first = (long long)n;
        if (first >= 0) sum += a;
    }
    if (st.first_index != first || st.count != nsamp - (size_t)first ||
        fabs(st.sum - sum) > 1e-6 * sum) {
        fprintf(stderr, "K3: first %lld (%lld) count %llu sum %.9g (%.9g)\n", (long long)st.first_index, first,
                (unsigned long long)st.count, st.sum, sum);
        return 1;
    }

    /* K4: the onset must sit inside one window of the burst start */
    gj_onset on;
    CHECK(gj_onset_u8(ctx, iq, nbytes, 200000, 1000, 50.0f, &on, &ms));
    if (on.start_index < 260000 - 1000 || on.start_index > 260000 + 1000) {
        fprintf(stderr, "K4 onset %lld not near 260000\n", (long long)on.start_index);
        return 1;
    }
    if (sizeof(gj_onset) != 32 || !(on.margin_hit > 0.f) || !(on.margin_before > 0.f)) {
        fprintf(stderr, "K4 margins %g %g (struct %zu bytes)\n", on.margin_hit, on.margin_before, sizeof(gj_onset));
        return 1;
    }

    /* error path: a too-small output buffer is reported, not overrun */
    if (gj_chunk_power_u8(ctx, iq, nbytes, chunk_bytes, 0.f, 0, power, nchunks - 1, &n_out, &ms) != GJ_ERR_CAPACITY) {
        fprintf(stderr, "capacity error not reported\n");
        return 1;
    }
    CHECK(gj_destroy(ctx));
    printf("c_abi_smoke OK: %zu chunks, first %lld, onset %lld\n", nchunks, first, (long long)on.start_index);
    free(power);
    free(iq);
    return 0;
}
