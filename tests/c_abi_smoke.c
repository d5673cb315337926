/* A plain C11 consumer of include/gpsjam.h (the reference's backend, gnssdec, is C): links
 * libgpsjam_hip.so, runs K1, K3 and K4 on a small deterministic capture through the host-buffer
 * entry points and checks them against the same arithmetic done here in integers/doubles, then
 * the device-resident per-stream flow with the library's own collectives.
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
    /* The per-stream flow of INTEGRATION.md section C from a host with no torch: resident capture, one-pass scan,
     * Welch rows, TDOA slot, the library's own RCCL collectives (a communicator of one rank here), pair solve, result
     * vector, gather.  With one antenna the only pair is (0, 0): lag 0. */
    {
        void *d_iq = NULL, *d_power = NULL, *d_stats = NULL, *d_amp = NULL, *d_on = NULL, *d_psd = NULL, *d_slot = NULL,
             *d_slots = NULL, *d_pairs = NULL, *d_lags = NULL, *d_peaks = NULL, *d_margins = NULL, *d_res = NULL, *d_all = NULL;
        const int nperseg = 1024;
        const size_t chunk_samples = 100000, nslice = 50000;
        const size_t rows = gj_welch_rows(nbytes, chunk_samples, nperseg), sb = gj_tdoa_slot_bytes(nslice);
        const size_t rlen = GJ_RESULT_HEADER + nchunks + (size_t)nperseg + GJ_RESULT_PAIR_FIELDS * 1;
        CHECK(gj_upload(ctx, iq, nbytes, &d_iq));
        CHECK(gj_malloc(ctx, nchunks * 4, &d_power));
        CHECK(gj_malloc(ctx, 16, &d_stats));
        CHECK(gj_malloc(ctx, sizeof(gj_amp_stats), &d_amp));
        CHECK(gj_malloc(ctx, sizeof(gj_onset), &d_on));
        CHECK(gj_malloc(ctx, rows * nperseg * 4, &d_psd));
        CHECK(gj_malloc(ctx, sb, &d_slot));
        CHECK(gj_malloc(ctx, sb, &d_slots));
        CHECK(gj_malloc(ctx, 8, &d_pairs));
        CHECK(gj_malloc(ctx, 4, &d_lags));
        CHECK(gj_malloc(ctx, 4, &d_peaks));
        CHECK(gj_malloc(ctx, 4, &d_margins));
        CHECK(gj_malloc(ctx, rlen * 8, &d_res));
        CHECK(gj_malloc(ctx, rlen * 8, &d_all));
        CHECK(gj_stream_scan_dev(ctx, d_iq, nbytes, chunk_bytes, 1e-10f, 0, d_power, 0.5f, d_amp, 200000, 1000, 50.0f, d_on));
        CHECK(gj_power_threshold_dev(ctx, d_power, nchunks, 5.0f, 6.0f, d_stats, NULL));
        CHECK(gj_welch_dev(ctx, d_iq, nbytes, chunk_samples, nperseg, 2.048e6, GJ_WELCH_SHIFT, d_psd, NULL));
        CHECK(gj_tdoa_slot_dev(ctx, d_iq, nbytes, (const int64_t*)d_on /* &start_index */, nslice, d_slot));
        /* round 5: the same scan + threshold + slot in TWO launches, K2 of two captures in one launch, both result vectors in
         * one launch -- every byte equal to the calls above (only gj_onset.margin_before, a bound, is left out) */
        {
            void *p2, *s2, *a2, *o2, *sl2, *psd_a, *psd_b, *res_a, *res_b;
            CHECK(gj_malloc(ctx, nchunks * 4, &p2));
            CHECK(gj_malloc(ctx, 16, &s2));
            CHECK(gj_malloc(ctx, sizeof(gj_amp_stats), &a2));
            CHECK(gj_malloc(ctx, sizeof(gj_onset), &o2));
            CHECK(gj_malloc(ctx, sb, &sl2));
            CHECK(gj_malloc(ctx, rows * nperseg * 4, &psd_a));
            CHECK(gj_malloc(ctx, rows * nperseg * 4, &psd_b));
            CHECK(gj_malloc(ctx, rlen * 8, &res_a));
            CHECK(gj_malloc(ctx, rlen * 8, &res_b));
            gj_scan_extra extra = {5.0f, 6.0f, (float*)s2, NULL, nslice, (uint8_t*)sl2};
            CHECK(gj_capture_scan_dev(ctx, d_iq, nbytes, chunk_bytes, 1e-10f, 0, p2, 0.5f, a2, 200000, 1000, 50.0f, o2, &extra));
            const uint8_t* caps2[2] = {d_iq, d_iq};
            float* psds2[2] = {(float*)psd_a, (float*)psd_b};
            CHECK(gj_welch_batch_dev(ctx, caps2, 2, nbytes, chunk_samples, nperseg, 2.048e6, GJ_WELCH_SHIFT, psds2));
            struct { void *got, *want; size_t n; const char* what; } cmp[] = {
                {p2, d_power, nchunks * 4, "power map"}, {s2, d_stats, 12, "threshold statistics"}, {a2, d_amp, sizeof(gj_amp_stats), "amplitude record"},
                {sl2, d_slot, sb, "TDOA slot"}, {psd_a, d_psd, rows * nperseg * 4, "PSD (batch, capture 0)"}, {psd_b, d_psd, rows * nperseg * 4, "PSD (batch, capture 1)"}};
            for (size_t k = 0; k < sizeof(cmp) / sizeof(cmp[0]); ++k) {
                unsigned char *g = (unsigned char*)malloc(cmp[k].n), *w = (unsigned char*)malloc(cmp[k].n);
                CHECK(gj_memcpy_d2h(ctx, g, cmp[k].got, cmp[k].n));
                CHECK(gj_memcpy_d2h(ctx, w, cmp[k].want, cmp[k].n));
                if (memcmp(g, w, cmp[k].n) != 0) { fprintf(stderr, "two-launch path: %s differs\n", cmp[k].what); return 1; }
                free(g); free(w);
            }
            gj_onset og, ow;
            CHECK(gj_memcpy_d2h(ctx, &og, o2, sizeof(og)));
            CHECK(gj_memcpy_d2h(ctx, &ow, d_on, sizeof(ow)));
            if (og.start_index != ow.start_index || og.guard_index != ow.guard_index || og.noise_power != ow.noise_power ||
                og.threshold != ow.threshold || og.margin_hit != ow.margin_hit || og.start_index != on.start_index) {
                fprintf(stderr, "two-launch path: onset %lld / %lld guard %lld / %lld\n", (long long)og.start_index, (long long)ow.start_index,
                        (long long)og.guard_index, (long long)ow.guard_index);
                return 1;
            }
            gj_combine_capture two[2];
            memset(two, 0, sizeof(two));
            for (int a = 0; a < 2; ++a) {
                two[a].n_chunks = nchunks; two[a].rows = rows; two[a].antenna = a; two[a].n_pairs = 0; two[a].pair_cap = 1;
                two[a].d_power = (float*)d_power; two[a].d_stats = (float*)d_stats; two[a].d_amp = (gj_amp_stats*)d_amp;
                two[a].d_onset = (gj_onset*)d_on; two[a].d_psd = (float*)d_psd; two[a].d_out = (double*)(a ? res_b : res_a);
            }
            CHECK(gj_pack_results_dev(ctx, two, 2, nperseg, NULL, NULL, NULL, NULL));
            CHECK(gj_pack_result_dev(ctx, nchunks, d_power, d_stats, d_amp, d_on, d_psd, rows, nperseg, 1, 0, 1, NULL, NULL, NULL, NULL, d_res));
            double *rb = (double*)malloc(rlen * 8), *rw = (double*)malloc(rlen * 8);
            CHECK(gj_memcpy_d2h(ctx, rb, res_b, rlen * 8));
            CHECK(gj_memcpy_d2h(ctx, rw, d_res, rlen * 8));
            if (memcmp(rb, rw, rlen * 8) != 0) { fprintf(stderr, "gj_pack_results_dev differs from gj_pack_result_dev\n"); return 1; }
            free(rb); free(rw);
            void* mine[] = {p2, s2, a2, o2, sl2, psd_a, psd_b, res_a, res_b};
            for (size_t k = 0; k < sizeof(mine) / sizeof(mine[0]); ++k) CHECK(gj_free(ctx, mine[k]));
        }
        unsigned char id[GJ_COMM_ID_BYTES];
        gj_comm* comm = NULL;
        const int rc_id = gj_comm_unique_id(id);
        if (rc_id != GJ_OK) { fprintf(stderr, "gj_comm_unique_id -> %d (librccl missing?)\n", rc_id); return 2; }
        CHECK(gj_comm_init_rank(ctx, id, 0, 1, &comm));
        int r = -1, nr = -1, cdev = -1;
        CHECK(gj_comm_rank(comm, &r, &nr));          /* read from the live communicator (ncclCommUserRank / ncclCommCount) */
        CHECK(gj_comm_device(comm, &cdev));
        char ident[160];
        CHECK(gj_device_identity(ctx, ident, sizeof(ident)));
        if (cdev != 0 || strncmp(ident, "pci=", 4) != 0 || !strstr(ident, " uuid=") || !strstr(ident, " hip=0")) {
            fprintf(stderr, "identity '%s', communicator device %d\n", ident, cdev);
            return 1;
        }
        CHECK(gj_comm_allgather_dev(comm, d_slot, sb, d_slots));
        const int32_t pair[2] = {0, 0};
        CHECK(gj_memcpy_h2d(ctx, d_pairs, pair, sizeof(pair)));
        CHECK(gj_xcorr_slots_dev(ctx, d_slots, sb, 1, nslice, pair, 1, d_lags, d_peaks, d_margins));
        CHECK(gj_pack_result_dev(ctx, nchunks, d_power, d_stats, d_amp, d_on, d_psd, rows, nperseg, 0, 1, 1, d_pairs, d_lags,
                                 d_peaks, d_margins, d_res));
        CHECK(gj_comm_gather_dev(comm, d_res, rlen * 8, d_all, 0));
        double* res = (double*)malloc(rlen * 8);
        CHECK(gj_memcpy_d2h(ctx, res, d_all, rlen * 8));
        const double* blk = res + GJ_RESULT_HEADER + nchunks + nperseg;
        int bad = r != 0 || nr != 1 || res[0] != (double)nchunks || res[4] != (double)st.first_index ||
                  res[5] != (double)st.count || res[7] != (double)on.start_index || res[11] != (double)rows ||
                  res[12] != (double)nperseg || res[13] != 0.0 || res[14] != 1.0 || res[15] != 1.0 || blk[0] != 0.0 ||
                  blk[1] != 0.0 || blk[2] != 0.0 /* lag of a slice against itself */ || !(blk[3] > 0.0);
        for (size_t c = 0; c < nchunks && !bad; ++c) bad = (float)res[GJ_RESULT_HEADER + c] != power[c];
        if (bad) {
            fprintf(stderr, "result vector: n %g first %g count %g onset %g rows %g nperseg %g rank %g pairs %g/%g lag %g peak %g\n",
                    res[0], res[4], res[5], res[7], res[11], res[12], res[13], res[14], res[15], blk[2], blk[3]);
            return 1;
        }
        free(res);
        CHECK(gj_comm_destroy(comm));
        void* all[] = {d_iq, d_power, d_stats, d_amp, d_on, d_psd, d_slot, d_slots, d_pairs, d_lags, d_peaks, d_margins, d_res, d_all};
        for (size_t k = 0; k < sizeof(all) / sizeof(all[0]); ++k) CHECK(gj_free(ctx, all[k]));
    }
    CHECK(gj_destroy(ctx));
    printf("c_abi_smoke OK: %zu chunks, first %lld, onset %lld\n", nchunks, first, (long long)on.start_index);
    free(power);
    free(iq);
    return 0;
}
