/* Plain C consumer of the round-3 entry points of include/gpsjam.h -- no Python, no torch:
 *  (1) gj_ingest_u8: upload + fused scan + Welch rows in one call, bit-identical to gj_upload followed by the
 *      separate calls; the "*_u8" entry points handed the resident DEVICE pointer;
 *  (2) one capture cut into three parts: gj_part_scan_dev / gj_part_welch_dev / gj_part_slot_dev per
 *      part, gj_amp_combine_dev / gj_onset_combine_dev / gj_slots_pick_dev for the capture -- every number equal,
 *      bit for bit, to what the unsplit capture gives.
 * Built and run by tests/test_c_abi_gpu.py (gcc, no HIP headers needed). */
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
#define FAIL(...) do { fprintf(stderr, __VA_ARGS__); return 1; } while (0)

static gj_ctx* ctx;

static void* dmalloc(size_t n) {
    void* p = NULL;
    if (gj_malloc(ctx, n ? n : 16, &p) != GJ_OK) { fprintf(stderr, "gj_malloc(%zu) failed\n", n); exit(2); }
    return p;
}

int main(void) {
    CHECK(gj_create(0, &ctx));
    /* 70 MiB + a ragged odd tail: above the 64-MiB threshold, so the ingest really runs piece by piece */
    const size_t nbytes = ((size_t)70 << 20) + 12345;
    const size_t chunk_bytes = 65536, chunk_samples = 262144; /* unit = lcm(65536, 524288) = 512 KiB */
    const int nperseg = 1024, noise = 200000, window = 1000;
    const size_t burst = 21000000;                            /* sample where the loud part starts */
    uint8_t* iq = (uint8_t*)malloc(nbytes);
    uint32_t x = 2463534242u;
    for (size_t i = 0; i < nbytes; ++i) {
        x ^= x << 13; x ^= x >> 17; x ^= x << 5;
        const int amp = (i / 2 >= burst) ? 100 : 9;
        iq[i] = (uint8_t)(128 + (int)(x % (2u * (unsigned)amp + 1u)) - amp);
    }
    const size_t nchunks = gj_chunk_count(nbytes, chunk_bytes), rows = gj_welch_rows(nbytes, chunk_samples, nperseg);
    const size_t nfl = rows * (size_t)nperseg;

    /* ---- the unsplit capture: upload, then the separate calls on the resident DEVICE pointer ---- */
    void* d_cap = NULL;
    CHECK(gj_upload(ctx, iq, nbytes, &d_cap));
    float* power = (float*)malloc(nchunks * 4);
    float* psd = (float*)malloc(nfl * 4);
    gj_amp_stats amp;
    gj_onset on;
    size_t n_out = 0, rows_out = 0;
    float ms = 0.f;
    CHECK(gj_chunk_power_u8(ctx, (const uint8_t*)d_cap, nbytes, chunk_bytes, 1e-10f, 0, power, nchunks, &n_out, &ms));
    CHECK(gj_amp_stats_u8(ctx, (const uint8_t*)d_cap, nbytes, 0.3f, &amp, &ms));
    CHECK(gj_onset_u8(ctx, (const uint8_t*)d_cap, nbytes, noise, window, 50.0f, &on, &ms));
    CHECK(gj_welch_u8(ctx, (const uint8_t*)d_cap, nbytes, chunk_samples, nperseg, 2.048e6, GJ_WELCH_SHIFT, psd, NULL, nfl, &rows_out, &ms));
    if (n_out != nchunks || rows_out != rows || on.start_index < (long long)burst - 1000 || on.start_index > (long long)burst + 1000 ||
        on.guard_index != on.start_index || amp.first_index < 0)
        FAIL("unsplit: chunks %zu rows %zu onset %lld guard %lld first %lld\n", n_out, rows_out, (long long)on.start_index,
             (long long)on.guard_index, (long long)amp.first_index);
    {   /* the same through host buffers: the device-pointer form must not differ */
        float* p2 = (float*)malloc(nchunks * 4);
        CHECK(gj_chunk_power_u8(ctx, iq, nbytes, chunk_bytes, 1e-10f, 0, p2, nchunks, &n_out, &ms));
        if (memcmp(p2, power, nchunks * 4)) FAIL("device-pointer and host-buffer power maps differ\n");
        free(p2);
    }

    /* ---- (1) gj_ingest_u8: everything while the capture uploads ---- */
    {
        gj_ingest_plan plan;
        memset(&plan, 0, sizeof(plan));
        plan.chunk_bytes = chunk_bytes; plan.eps = 1e-10f; plan.rssi_threshold = 0.3f; plan.noise_samples = noise;
        plan.window = window; plan.factor = 50.0f; plan.chunk_samples = chunk_samples; plan.nperseg = nperseg;
        plan.welch_flags = GJ_WELCH_SHIFT; plan.fs = 2.048e6;
        gj_ingest_result res;
        float* p2 = (float*)malloc(nchunks * 4);
        float* s2 = (float*)malloc(nfl * 4);
        void* d2 = NULL;
        CHECK(gj_ingest_u8(ctx, iq, nbytes, &plan, p2, nchunks, s2, NULL, nfl, &res, &d2));
        if (res.nbytes != nbytes || res.n_chunks != nchunks || res.rows != rows || memcmp(p2, power, nchunks * 4) ||
            memcmp(s2, psd, nfl * 4) || memcmp(&res.amp, &amp, sizeof(amp)) || memcmp(&res.onset, &on, sizeof(on)))
            FAIL("ingest differs from upload-then-run (sum %.17g vs %.17g, onset %lld vs %lld)\n", res.amp.sum, amp.sum,
                 (long long)res.onset.start_index, (long long)on.start_index);
        uint8_t back[64];
        CHECK(gj_memcpy_d2h(ctx, back, (const uint8_t*)d2 + nbytes - 64, 64));
        if (memcmp(back, iq + nbytes - 64, 64)) FAIL("the ingested capture is not the capture\n");
        printf("ingest: upload %.2f ms, total %.2f ms, identical to upload-then-run\n", res.upload_ms, res.total_ms);
        CHECK(gj_free(ctx, d2));
        free(p2);
        free(s2);
    }

    /* ---- (2) the capture in three parts ---- */
    const size_t unit = 524288, units = (nbytes + unit - 1) / unit, tile = 65536, nslice = 1 << 15;
    const size_t cut[4] = {0, (units / 3) * unit, (2 * units / 3) * unit, nbytes};
    const size_t sb = gj_tdoa_slot_bytes(nslice);
    const size_t ntiles_total = gj_amp_tile_count(nbytes);
    float* d_power_all = (float*)dmalloc(nchunks * 4);
    float* d_psd_all = (float*)dmalloc(nfl * 4);
    uint8_t* d_tiles_all = (uint8_t*)dmalloc(ntiles_total * 16);
    gj_amp_part* d_amp_parts = (gj_amp_part*)dmalloc(3 * sizeof(gj_amp_part));
    gj_onset* d_on_parts = (gj_onset*)dmalloc(3 * sizeof(gj_onset));
    uint8_t* d_slots = (uint8_t*)dmalloc(3 * sb);
    void* d_noise = NULL;
    CHECK(gj_upload(ctx, iq, 2 * (size_t)noise, &d_noise));
    void* bufs[3];
    for (int g = 0; g < 3; ++g) {
        gj_part_view v;
        memset(&v, 0, sizeof(v));
        const size_t halo = g ? tile : 0;
        v.own_first_byte = cut[g];
        v.own_bytes = cut[g + 1] - cut[g];
        v.total_bytes = nbytes;
        v.buf_first_byte = cut[g] - halo;
        size_t end = cut[g + 1] + 2 * nslice + tile;            /* a tail of one slice behind the own range */
        if (end > nbytes) end = nbytes;
        v.buf_bytes = end - v.buf_first_byte;
        CHECK(gj_upload(ctx, iq + v.buf_first_byte, v.buf_bytes, &bufs[g]));   /* each "rank" reads its own range */
        v.d_buf = (const uint8_t*)bufs[g];
        v.d_noise = g ? (const uint8_t*)d_noise : NULL;
        CHECK(gj_part_scan_dev(ctx, &v, chunk_bytes, 1e-10f, 0, d_power_all + cut[g] / chunk_bytes, 0.3f,
                               d_tiles_all + (cut[g] / tile) * 16, d_amp_parts + g, noise, window, 50.0f, d_on_parts + g));
        CHECK(gj_part_welch_dev(ctx, &v, chunk_samples, nperseg, 2.048e6, GJ_WELCH_SHIFT,
                                d_psd_all + (cut[g] / (2 * chunk_samples)) * (size_t)nperseg, NULL));
        CHECK(gj_part_slot_dev(ctx, &v, &d_on_parts[g].start_index, nslice, d_slots + (size_t)g * sb));
    }
    gj_amp_stats* d_amp = (gj_amp_stats*)dmalloc(sizeof(gj_amp_stats));
    gj_onset* d_on = (gj_onset*)dmalloc(sizeof(gj_onset));
    uint8_t* d_slot_c = (uint8_t*)dmalloc(sb);
    uint8_t* d_slot_w = (uint8_t*)dmalloc(sb);
    int32_t* d_groups = (int32_t*)dmalloc(32);
    const int32_t groups[5] = {0, 3, /* members */ 0, 1, 2};
    CHECK(gj_memcpy_h2d(ctx, d_groups, groups, sizeof(groups)));
    CHECK(gj_amp_combine_dev(ctx, d_tiles_all, ntiles_total, d_amp_parts, 3, nbytes, d_amp));
    CHECK(gj_onset_combine_dev(ctx, d_on_parts, 3, d_on));
    CHECK(gj_slots_pick_dev(ctx, d_slots, sb, d_groups, d_groups + 2, 1, d_slot_c));
    /* the unsplit capture's slot, for comparison */
    gj_onset* d_on_w = (gj_onset*)dmalloc(sizeof(gj_onset));
    CHECK(gj_memcpy_h2d(ctx, d_on_w, &on, sizeof(on)));
    CHECK(gj_tdoa_slot_dev(ctx, (const uint8_t*)d_cap, nbytes, &d_on_w->start_index, nslice, d_slot_w));
    float* p3 = (float*)malloc(nchunks * 4);
    float* s3 = (float*)malloc(nfl * 4);
    uint8_t* slot_c = (uint8_t*)malloc(sb);
    uint8_t* slot_w = (uint8_t*)malloc(sb);
    gj_amp_stats amp3;
    gj_onset on3;
    CHECK(gj_memcpy_d2h(ctx, p3, d_power_all, nchunks * 4));
    CHECK(gj_memcpy_d2h(ctx, s3, d_psd_all, nfl * 4));
    CHECK(gj_memcpy_d2h(ctx, &amp3, d_amp, sizeof(amp3)));
    CHECK(gj_memcpy_d2h(ctx, &on3, d_on, sizeof(on3)));
    CHECK(gj_memcpy_d2h(ctx, slot_c, d_slot_c, sb));
    CHECK(gj_memcpy_d2h(ctx, slot_w, d_slot_w, sb));
    if (memcmp(p3, power, nchunks * 4)) FAIL("split: power map differs\n");
    if (memcmp(s3, psd, nfl * 4)) FAIL("split: PSD rows differ\n");
    if (memcmp(&amp3, &amp, sizeof(amp))) FAIL("split: amplitude statistics differ (sum %.17g vs %.17g, first %lld vs %lld)\n", amp3.sum, amp.sum,
                                                (long long)amp3.first_index, (long long)amp.first_index);
    if (on3.start_index != on.start_index || on3.guard_index != on.guard_index || on3.margin_hit != on.margin_hit ||
        on3.noise_power != on.noise_power || on3.threshold != on.threshold)
        FAIL("split: onset differs (%lld vs %lld)\n", (long long)on3.start_index, (long long)on.start_index);
    if (memcmp(slot_c, slot_w, sb)) FAIL("split: the picked TDOA slot is not the unsplit capture's slot\n");
    printf("split: 3 parts, %zu chunks, %zu rows, %zu tiles, onset %lld, first %lld: identical to the unsplit capture\n", nchunks,
           rows, ntiles_total, (long long)on3.start_index, (long long)amp3.first_index);
    void* all[] = {d_cap, d_power_all, d_psd_all, d_tiles_all, d_amp_parts, d_on_parts, d_slots, d_noise, bufs[0], bufs[1], bufs[2],
                   d_amp, d_on, d_slot_c, d_slot_w, d_groups, d_on_w};
    for (size_t k = 0; k < sizeof(all) / sizeof(all[0]); ++k) CHECK(gj_free(ctx, all[k]));
    CHECK(gj_destroy(ctx));
    printf("c_split_ingest OK\n");
    return 0;
}
