/* A caller that is killed inside the library must not take the context with it (the GUI stops
 * an analysis with QThread.terminate(), GpsJammerApp/app/ui_mainwindow.py:818-826).  Threads are ended with a raw
 * exit system call -- no unwinding, no destructors, exactly what a hard kill leaves behind -- at the library's own
 * wait sites (gj_debug_set_wait_hook), and the main thread then keeps using the same context:
 *   A dies inside a 256-MiB gj_upload (staged copy)          -> its lane is taken back, the next call is correct
 *   A' the same, and the owner probe is made to answer "alive" once (gj_debug_inject: the wrong answer a sampled
 *      probe gave in GPUTEST_r04)                            -> a second lane is made; the missed lane still comes back at
 *      the NEXT check-out (and, in a second run of the step, at the next gj_debug_counters call)
 *   B dies inside gj_chunk_power_u8, waiting for its event   -> same
 *   C dies as the OWNER of the context mutex                 -> the next locker recovers the mutex
 *   D is pthread_cancel'ed (QThread.terminate() on POSIX) while inside gj_upload -> the call completes, the
 *     cancellation takes effect after it has returned (the library defers it), nothing is unwound half way
 * gcc tests/c_abandoned_caller.c -Iinclude -Lgps-jamming_amd/csrc -lgpsjam_hip -lpthread */
#define _GNU_SOURCE
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/syscall.h>
#include <unistd.h>

#include "gpsjam.h"

static __thread int doomed_site = 0;
static volatile int inside_upload = 0, uploads_done = 0;
static __thread int announce = 0;
static void hook(void* arg, int site) {
    (void)arg;
    if (announce && site == 3) inside_upload = 1;
    if (doomed_site && site == doomed_site) syscall(SYS_exit, 0); /* this thread only; nothing is unwound */
}

static gj_ctx* ctx;
static uint8_t* big;
static const size_t BIG = (size_t)256 << 20;

static void* cancelled_in_upload(void* p) {
    (void)p;
    announce = 1;                               /* the hook below raises `inside_upload` at the staged-copy site */
    for (;;) {
        void* d = NULL;
        if (gj_upload(ctx, big, BIG, &d) == GJ_OK) {
            gj_free(ctx, d);
            ++uploads_done;
        }
        pthread_testcancel();                   /* outside the library: here a pending cancellation may act */
    }
    return (void*)1;
}

static void* die_in_upload(void* p) {
    (void)p;
    doomed_site = 3;
    void* d = NULL;
    gj_upload(ctx, big, BIG, &d);
    return (void*)1; /* not reached */
}
static void* die_in_event_wait(void* p) {
    (void)p;
    doomed_site = 1;
    float pw[4096], ms;
    size_t n;
    gj_chunk_power_u8(ctx, big, 64u << 20, 65536, 0.f, 0, pw, 4096, &n, &ms);
    return (void*)1;
}
static void* die_holding_lock(void* p) {
    (void)p;
    doomed_site = 5;
    int a, b, c, d;
    gj_debug_counters(ctx, &a, &b, &c, &d);
    return (void*)1;
}

static int check_power(const char* what) {
    /* bytes alternate 100, 150: (100-127.5)^2 + (150-127.5)^2 = 1262.5 per I/Q pair */
    static float pw[64];
    size_t n = 0;
    float ms = 0.f;
    int rc = gj_chunk_power_u8(ctx, big, (size_t)4 << 20, 65536, 0.f, 0, pw, 64, &n, &ms);
    if (rc != GJ_OK || n != 64) {
        fprintf(stderr, "%s: rc %d (%s) n %zu\n", what, rc, gj_last_error(ctx), n);
        return 1;
    }
    for (size_t k = 0; k < n; ++k)
        if (pw[k] != 1262.5f) {
            fprintf(stderr, "%s: power[%zu] = %f\n", what, k, pw[k]);
            return 1;
        }
    return 0;
}

int main(void) {
    if (gj_create(0, &ctx) != GJ_OK) return 2;
    big = (uint8_t*)malloc(BIG);
    for (size_t i = 0; i < BIG; ++i) big[i] = (i & 1) ? 150 : 100;
    gj_debug_set_wait_hook(ctx, hook, NULL);
    int lanes, busy, reclaimed, deaths;
    pthread_t t;
    void* (*killers[3])(void*) = {die_in_upload, die_in_event_wait, die_holding_lock};
    const char* names[3] = {"after a caller died inside gj_upload", "after a caller died waiting for its event",
                            "after a caller died holding the context mutex"};
    for (int k = 0; k < 3; ++k) {
        void* ret = NULL;
        pthread_create(&t, NULL, killers[k], NULL);
        pthread_join(t, &ret);
        if (ret == (void*)1) {
            fprintf(stderr, "the doomed thread of step %d came back\n", k);
            return 3;
        }
        if (check_power(names[k])) return 4;
        if (check_power(names[k])) return 4;
        gj_debug_counters(ctx, &lanes, &busy, &reclaimed, &deaths);
        printf("%s: lanes %d busy %d reclaimed %d owner_deaths %d\n", names[k], lanes, busy, reclaimed, deaths);
        if (busy != 0) return 5;
        /* the dead caller's lane is re-used, not replaced: ownership is the kernel's verdict (a robust mutex), so the
         * first check-out after the join already sees it -- whatever the timing of the thread's exit */
        if (k == 0 && (lanes != 1 || reclaimed != 1)) return 9;
    }
    if (reclaimed < 2 || deaths < 1) return 6;
    /* A': the probe misses the dead owner once.  Round 4's state (lanes 2, busy 1, reclaimed unchanged) is reproduced on
     * purpose, then the lane must come back: first at the next check-out, second time at the next counters call. */
    for (int variant = 0; variant < 2; ++variant) {
        void* ret = NULL;
        int lanes0, busy0, reclaimed0;
        gj_debug_counters(ctx, &lanes0, &busy0, &reclaimed0, &deaths);
        pthread_create(&t, NULL, die_in_upload, NULL);
        pthread_join(t, &ret);
        if (ret == (void*)1) return 3;
        /* variant 0: two wrong answers (the check-out's sweep and the counters' sweep) so that the missed state can be read */
        if (gj_debug_inject(ctx, GJ_INJECT_OWNER_ALIVE, variant == 0 ? 2 : 1) != GJ_OK) return 10;
        if (check_power("with the dead owner reported alive")) return 4;   /* served from another lane */
        if (variant == 0) {
            gj_debug_counters(ctx, &lanes, &busy, &reclaimed, &deaths);
            printf("dead owner reported alive: lanes %d busy %d reclaimed %d (the state GPUTEST_r04 ended in)\n", lanes, busy, reclaimed);
            if (busy != 1 || reclaimed != reclaimed0 || lanes < 2) return 11;
            if (check_power("at the next check-out")) return 4;             /* its sweep takes the lane back */
        }
        gj_debug_counters(ctx, &lanes, &busy, &reclaimed, &deaths);        /* variant 1: THIS call's sweep takes it back */
        printf("missed once, then %s: lanes %d busy %d reclaimed %d\n", variant == 0 ? "the next check-out" : "the next counters call",
               lanes, busy, reclaimed);
        if (busy != 0 || reclaimed != reclaimed0 + 1) return 12;
    }
    /* more dead callers than there are lanes (8): every one of them comes back, nothing spins */
    for (int k = 0; k < 12; ++k) {
        void* ret = NULL;
        pthread_create(&t, NULL, die_in_event_wait, NULL);
        pthread_join(t, &ret);
        if (ret == (void*)1) return 3;
    }
    if (check_power("after twelve more dead callers")) return 4;
    gj_debug_counters(ctx, &lanes, &busy, &reclaimed, &deaths);
    printf("after twelve more dead callers: lanes %d busy %d reclaimed %d\n", lanes, busy, reclaimed);
    if (busy != 0 || lanes > 8) return 13;
    /* D: cancelled while inside gj_upload */
    {
        void* ret = NULL;
        pthread_create(&t, NULL, cancelled_in_upload, NULL);
        while (!inside_upload) usleep(100);
        pthread_cancel(t);                      /* the request arrives while the thread is inside the library */
        pthread_join(t, &ret);
        if (ret != PTHREAD_CANCELED) { fprintf(stderr, "the cancelled thread was not cancelled\n"); return 8; }
        if (uploads_done < 1) { fprintf(stderr, "the upload was torn down half way (%d done)\n", uploads_done); return 8; }
        if (check_power("after a caller was cancelled inside gj_upload")) return 4;
        gj_debug_counters(ctx, &lanes, &busy, &reclaimed, &deaths);
        printf("after a caller was cancelled inside gj_upload: %d upload(s) completed first, lanes %d busy %d\n", uploads_done, lanes, busy);
        if (busy != 0) return 5;
    }
    /* the context still uploads and frees */
    void* d = NULL;
    if (gj_upload(ctx, big, BIG, &d) != GJ_OK || gj_free(ctx, d) != GJ_OK) return 7;
    gj_debug_set_wait_hook(ctx, NULL, NULL);
    gj_destroy(ctx);
    free(big);
    printf("abandoned callers: ok\n");
    return 0;
}
