// Host-side scenarios of libgpsjam_hip's C-ABI for ThreadSanitizer and ASan + UBSan (VERDICT r05 "next" 1).
// The library's eight .hip files are compiled HOST-ONLY with the sanitizer and linked against tests/hip_stub (streams
// with worker threads, asynchronous copies, no-op kernels, an in-process RCCL stand-in) -- no GPU, no numbers: what is
// exercised is the threaded host logic the reference's callers drive from three host threads and kill one of
// (GpsJammerApp/app/worker.py:488-490,610-611, ui_mainwindow.py:818-826): lanes with robust owner mutexes, the fill-
// thread pool, gj_ingest_files' own threads, the grow-only workspace and its retired arenas, the in-flight counting of
// collectives against gj_comm_destroy / gj_destroy.  Each scenario returns 0; the sanitizer's verdict is the test.
//   san_scenarios <scenario> [tmpdir]
#include <dlfcn.h>
#include <pthread.h>
#include <sys/syscall.h>
#include <unistd.h>

#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "gpsjam.h"
#include "hip_stub.h"

#define CHECK(cond)                                                                      \
    do {                                                                                 \
        if (!(cond)) {                                                                   \
            fprintf(stderr, "%s:%d: CHECK(%s) failed\n", __FILE__, __LINE__, #cond);     \
            return 1;                                                                    \
        }                                                                                \
    } while (0)
#define OK(call)                                                                                           \
    do {                                                                                                   \
        const int rc__ = (call);                                                                           \
        if (rc__ != GJ_OK) {                                                                               \
            fprintf(stderr, "%s:%d: %s -> %d (%s)\n", __FILE__, __LINE__, #call, rc__, gj_strerror(rc__)); \
            return 1;                                                                                      \
        }                                                                                                  \
    } while (0)

static std::vector<uint8_t> make_capture(size_t n, unsigned seed) {
    std::vector<uint8_t> v(n);
    unsigned x = seed * 2654435761u + 1u;
    for (size_t i = 0; i < n; ++i) {
        x = x * 1664525u + 1013904223u;
        v[i] = (uint8_t)(96 + ((x >> 24) & 63));
    }
    return v;
}

static gj_ingest_plan default_plan(int nperseg = 1024) {
    gj_ingest_plan p;
    memset(&p, 0, sizeof(p));
    p.chunk_bytes = 65536;
    p.eps = 1e-10f;
    p.rssi_threshold = 0.f;
    p.noise_samples = 200000;
    p.window = 1000;
    p.factor = 50.f;
    p.chunk_samples = 2048000;
    p.nperseg = nperseg;
    p.welch_flags = GJ_WELCH_SHIFT;
    p.fs = 2.048e6;
    return p;
}

// ------------------------------------------------------------------------------------------------------------------
// 1. eight threads making *_u8 / upload / ingest calls on ONE context while a ninth reads the counters and a tenth
//    moves the context between streams (AntennaStream / SplitStreams do that at construction: gj_set_stream)
// ------------------------------------------------------------------------------------------------------------------
static int caller(gj_ctx* ctx, int t, int rounds, std::atomic<int>* bad) {
    const size_t small = (size_t)(600000 + 4096 * t), big = (size_t)(6u << 20) + 65536 * (size_t)t + 2 * (size_t)t;
    std::vector<uint8_t> a = make_capture(small, (unsigned)t), b = make_capture(big, 100u + (unsigned)t);
    std::vector<float> power(4096), psd(8 * 4096);
    for (int r = 0; r < rounds; ++r) {
        size_t n = 0, rows = 0;
        float ms = 0.f;
        gj_amp_stats amp;
        gj_onset on;
        switch ((r + t) % 8) {
            case 0: OK(gj_chunk_power_u8(ctx, a.data(), small, 65536, 0.f, 0, power.data(), power.size(), &n, &ms)); break;
            case 1: OK(gj_welch_u8(ctx, a.data(), small, 200000, 1024, 2.048e6, GJ_WELCH_SHIFT, psd.data(), nullptr, psd.size(), &rows, &ms)); break;
            case 2: OK(gj_amp_stats_u8(ctx, a.data(), small, 0.1f, &amp, &ms)); break;
            case 3: OK(gj_onset_u8(ctx, a.data(), small, 200000, 1000, 50.f, &on, &ms)); break;
            case 4: {
                const uint8_t* sl[3] = {a.data(), a.data() + 2000, a.data() + 4000};
                const int32_t pairs[6] = {0, 1, 0, 2, 1, 2};
                int32_t lags[3];
                float peaks[3], margins[3];
                OK(gj_xcorr_lags_u8(ctx, sl, 3, 4096, pairs, 3, lags, peaks, margins, &ms));
                break;
            }
            case 5: {   // staged copy (>= 4 MiB: fill threads, bounce buffers) + a *_u8 call on the resident capture
                void* d = nullptr;
                OK(gj_upload(ctx, b.data(), big, &d));
                OK(gj_chunk_power_u8(ctx, static_cast<const uint8_t*>(d), big, 65536, 0.f, 0, power.data(), power.size(), &n, &ms));
                OK(gj_free(ctx, d));
                break;
            }
            case 6: {   // overlapped ingest: the dispatcher launches on pieces while the fill threads copy
                gj_ingest_plan plan = default_plan();
                gj_ingest_result res;
                void* d = nullptr;
                OK(gj_ingest_u8(ctx, b.data(), big, &plan, power.data(), power.size(), psd.data(), nullptr, psd.size(), &res, &d));
                CHECK(res.nbytes == big && res.n_chunks == gj_chunk_count(big, 65536));
                OK(gj_free(ctx, d));
                break;
            }
            case 7: {   // small ingest (one piece) with a chunk size the fused pass does not take
                gj_ingest_plan plan = default_plan(0);
                plan.chunk_bytes = 1000;
                gj_ingest_result res;
                void* d = nullptr;
                std::vector<float> pw(small / 1000 + 2);
                OK(gj_ingest_u8(ctx, a.data(), small, &plan, pw.data(), pw.size(), nullptr, nullptr, 0, &res, &d));
                OK(gj_free(ctx, d));
                break;
            }
        }
    }
    (void)bad;
    return 0;
}

static int scenario_threads() {
    gj_ctx* ctx = nullptr;
    OK(gj_create(0, &ctx));
    std::atomic<int> bad{0}, stop{0};
    std::vector<std::thread> pool;
    for (int t = 0; t < 8; ++t)
        pool.emplace_back([&, t] {
            if (caller(ctx, t, 24, &bad)) bad.fetch_add(1);
        });
    std::thread counters([&] {
        while (!stop.load()) {
            int lanes, busy, reclaimed, deaths;
            if (gj_debug_counters(ctx, &lanes, &busy, &reclaimed, &deaths) != GJ_OK || lanes > 8 || busy > lanes) bad.fetch_add(1);
            double off, sc;
            if (gj_get_unpack(ctx, &off, &sc) != GJ_OK) bad.fetch_add(1);
            usleep(300);
        }
    });
    std::thread mover([&] {   // the context hops between its own stream and the legacy default stream
        int k = 0;
        while (!stop.load()) {
            if (gj_set_stream(ctx, nullptr, (k++ & 1)) != GJ_OK) bad.fetch_add(1);
            if (gj_set_fill_threads(ctx, k % 5) != GJ_OK) bad.fetch_add(1);
            usleep(700);
        }
    });
    for (auto& th : pool) th.join();
    stop.store(1);
    counters.join();
    mover.join();
    OK(gj_set_stream(ctx, nullptr, 0));
    OK(gj_synchronize(ctx));
    int lanes, busy, reclaimed, deaths;
    OK(gj_debug_counters(ctx, &lanes, &busy, &reclaimed, &deaths));
    CHECK(bad.load() == 0 && busy == 0 && reclaimed == 0 && deaths == 0);
    OK(gj_destroy(ctx));
    CHECK(hip_stub_live_allocations() == 0);
    return 0;
}

// ------------------------------------------------------------------------------------------------------------------
// 2. gj_ingest_files with three files (the library's own threads), twice at once from two caller threads, with
//    gj_set_fill_threads changed meanwhile (the per-call override must not touch the context's setting)
// ------------------------------------------------------------------------------------------------------------------
static int scenario_ingest_files(const char* tmpdir) {
    const size_t nbytes = (9u << 20) + 12346;
    std::vector<std::string> paths;
    for (int k = 0; k < 3; ++k) {
        paths.push_back(std::string(tmpdir) + "/cap" + std::to_string(k) + ".bin");
        std::vector<uint8_t> v = make_capture(nbytes, 7u + (unsigned)k);
        FILE* f = fopen(paths.back().c_str(), "wb");
        CHECK(f && fwrite(v.data(), 1, nbytes, f) == nbytes);
        fclose(f);
    }
    gj_ctx* ctx = nullptr;
    OK(gj_create(0, &ctx));
    OK(gj_set_fill_threads(ctx, 3));
    std::atomic<int> bad{0};
    auto run = [&](int rounds) {
        const gj_ingest_plan plan = default_plan();
        for (int r = 0; r < rounds; ++r) {
            gj_ingest_job jobs[3];
            std::vector<float> power[3], psd[3];
            memset(jobs, 0, sizeof(jobs));
            for (int k = 0; k < 3; ++k) {
                power[k].resize(gj_chunk_count(nbytes, 65536));
                psd[k].resize(3 * 1024);
                jobs[k].path = paths[(size_t)k].c_str();
                jobs[k].power = power[k].data();
                jobs[k].power_cap = power[k].size();
                jobs[k].psd = psd[k].data();
                jobs[k].psd_cap_floats = psd[k].size();
            }
            if (gj_ingest_files(ctx, jobs, 3, &plan) != GJ_OK) bad.fetch_add(1);
            for (int k = 0; k < 3; ++k) {
                if (jobs[k].status != GJ_OK || jobs[k].result.nbytes != nbytes) bad.fetch_add(1);
                if (jobs[k].dptr && gj_free(ctx, jobs[k].dptr) != GJ_OK) bad.fetch_add(1);
            }
        }
    };
    std::thread a(run, 4), b(run, 4);
    std::thread setter([&] {
        for (int k = 0; k < 40; ++k) {
            (void)gj_set_fill_threads(ctx, 3);   // always the same value: whatever interleaving, it must still be 3 at the end
            usleep(500);
        }
    });
    a.join();
    b.join();
    setter.join();
    // a missing file: the status and the message come home from the library's own thread
    {
        const gj_ingest_plan plan = default_plan();
        gj_ingest_job jobs[2];
        memset(jobs, 0, sizeof(jobs));
        std::vector<float> power(gj_chunk_count(nbytes, 65536)), psd(3 * 1024);
        std::string missing = std::string(tmpdir) + "/does_not_exist.bin";
        jobs[0].path = paths[0].c_str();
        jobs[1].path = missing.c_str();
        for (int k = 0; k < 2; ++k) {
            jobs[k].power = power.data(); jobs[k].power_cap = power.size();
            jobs[k].psd = psd.data(); jobs[k].psd_cap_floats = psd.size();
        }
        jobs[1].power = nullptr;
        CHECK(gj_ingest_files(ctx, jobs, 2, &plan) == GJ_ERR_INVALID);
        CHECK(strstr(gj_last_error(ctx), "does_not_exist") != nullptr);
        CHECK(jobs[0].status == GJ_OK && jobs[1].status == GJ_ERR_INVALID);
        OK(gj_free(ctx, jobs[0].dptr));
    }
    CHECK(bad.load() == 0);
    OK(gj_destroy(ctx));
    CHECK(hip_stub_live_allocations() == 0);
    for (auto& p : paths) unlink(p.c_str());
    return 0;
}

// ------------------------------------------------------------------------------------------------------------------
// 3. the grow-only workspace: *_dev calls of growing size from four threads (ensure_workspace retires the old arena
//    behind an event), gj_reserve and gj_synchronize (reap_retired) from two more
// ------------------------------------------------------------------------------------------------------------------
static int scenario_workspace() {
    gj_ctx* ctx = nullptr;
    OK(gj_create(0, &ctx));
    const size_t cap_bytes = 24u << 20;
    void* d_cap = nullptr;
    {
        std::vector<uint8_t> v = make_capture(cap_bytes, 3);
        OK(gj_upload(ctx, v.data(), cap_bytes, &d_cap));
    }
    std::atomic<int> bad{0}, stop{0};
    std::vector<std::thread> pool;
    for (int t = 0; t < 4; ++t)
        pool.emplace_back([&, t] {
            void *d_pow = nullptr, *d_psd = nullptr, *d_amp = nullptr, *d_on = nullptr, *d_st = nullptr;
            if (gj_malloc(ctx, 4 * 4096, &d_pow) || gj_malloc(ctx, 4 * 16 * 4096, &d_psd) || gj_malloc(ctx, 64, &d_amp) ||
                gj_malloc(ctx, 64, &d_on) || gj_malloc(ctx, 64, &d_st)) { bad.fetch_add(1); return; }
            for (int r = 0; r < 40; ++r) {
                const size_t nbytes = ((size_t)(r + 1) * (cap_bytes / 40)) & ~(size_t)1;   // growing: the workspace follows
                int rc = 0;
                switch ((r + t) % 5) {
                    case 0: rc = gj_welch_dev(ctx, static_cast<uint8_t*>(d_cap), nbytes, 2048000, 4096, 2.048e6, 0, static_cast<float*>(d_psd), nullptr); break;
                    case 1: rc = gj_stream_scan_dev(ctx, static_cast<uint8_t*>(d_cap), nbytes, 65536, 0.f, 0, static_cast<float*>(d_pow), 0.f,
                                                    static_cast<gj_amp_stats*>(d_amp), 200000, 1000, 50.f, static_cast<gj_onset*>(d_on)); break;
                    case 2: rc = gj_onset_dev(ctx, static_cast<uint8_t*>(d_cap) + 2, nbytes - 2, 200000, 1000, 50.f, static_cast<gj_onset*>(d_on)); break;   // unaligned: copied into the workspace
                    case 3: rc = gj_chunk_power_dev(ctx, static_cast<uint8_t*>(d_cap), nbytes, 131072, 0.f, 0, static_cast<float*>(d_pow)); break;
                    case 4: {
                        gj_scan_extra x;
                        memset(&x, 0, sizeof(x));
                        x.pct = 5.f; x.rise_db = 6.f; x.d_stats = static_cast<float*>(d_st);
                        rc = gj_capture_scan_dev(ctx, static_cast<uint8_t*>(d_cap), nbytes | 1, 65536, 0.f, GJ_CP_ODD_CHUNK_ZERO, static_cast<float*>(d_pow), 0.f,
                                                 static_cast<gj_amp_stats*>(d_amp), 200000, 1000, 50.f, static_cast<gj_onset*>(d_on), &x);
                        break;
                    }
                }
                if (rc != GJ_OK) { fprintf(stderr, "thread %d round %d: rc %d (%s)\n", t, r, rc, gj_last_error(ctx)); bad.fetch_add(1); }
            }
            if (gj_free(ctx, d_pow) || gj_free(ctx, d_psd) || gj_free(ctx, d_amp) || gj_free(ctx, d_on) || gj_free(ctx, d_st)) bad.fetch_add(1);
        });
    std::thread reserver([&] {
        size_t want = 1u << 20;
        while (!stop.load()) {
            if (gj_reserve(ctx, want) != GJ_OK) bad.fetch_add(1);
            want += 3u << 20;
            usleep(400);
        }
    });
    std::thread syncer([&] {
        while (!stop.load()) {
            if (gj_synchronize(ctx) != GJ_OK) bad.fetch_add(1);
            usleep(150);
        }
    });
    for (auto& th : pool) th.join();
    stop.store(1);
    reserver.join();
    syncer.join();
    OK(gj_free(ctx, d_cap));
    CHECK(bad.load() == 0);
    OK(gj_destroy(ctx));
    CHECK(hip_stub_live_allocations() == 0);
    return 0;
}

// ------------------------------------------------------------------------------------------------------------------
// 4. communicators: (a) two ranks (two contexts, two threads) exchanging slots and result vectors like
//    gpsjam/sharded.py; (b) gj_comm_destroy racing a collective that another thread is enqueueing;
//    (c) gj_destroy of the context racing it -- the handle stays valid for gj_comm_destroy afterwards
// ------------------------------------------------------------------------------------------------------------------
static int scenario_comm() {
    unsigned char id[GJ_COMM_ID_BYTES];
    OK(gj_comm_unique_id(id));
    std::atomic<int> bad{0};
    auto rank_main = [&](int rank) {
        gj_ctx* ctx = nullptr;
        gj_comm* c = nullptr;
        if (gj_create(rank, &ctx) != GJ_OK || gj_comm_init_rank(ctx, id, rank, 2, &c) != GJ_OK) { bad.fetch_add(1); return; }
        int r = -1, n = 0, dev = -1;
        if (gj_comm_rank(c, &r, &n) != GJ_OK || r != rank || n != 2 || gj_comm_device(c, &dev) != GJ_OK || dev != rank) bad.fetch_add(1);
        void *d_send = nullptr, *d_recv = nullptr;
        const size_t bytes = 4096;
        if (gj_malloc(ctx, bytes, &d_send) || gj_malloc(ctx, 2 * bytes, &d_recv)) { bad.fetch_add(1); return; }
        std::vector<uint8_t> mine(bytes, (uint8_t)(0x10 + rank)), got(2 * bytes);
        for (int round = 0; round < 20; ++round) {
            if (gj_memcpy_h2d(ctx, d_send, mine.data(), bytes)) bad.fetch_add(1);
            if (gj_comm_allgather_dev(c, d_send, bytes, d_recv) != GJ_OK) bad.fetch_add(1);
            if (gj_comm_gather_dev(c, d_send, bytes, d_recv, 0) != GJ_OK) bad.fetch_add(1);
            if (gj_comm_bcast_dev(c, d_recv, bytes, 0) != GJ_OK) bad.fetch_add(1);
            if (gj_memcpy_d2h(ctx, got.data(), d_recv, 2 * bytes)) bad.fetch_add(1);
            if (got[0] != 0x10 || got[bytes] != 0x11) bad.fetch_add(1);   // the stand-in really moves the bytes
        }
        if (gj_free(ctx, d_send) || gj_free(ctx, d_recv)) bad.fetch_add(1);
        if (rank == 0) {   // one rank drops the communicator first, the other the context first
            if (gj_comm_destroy(c) != GJ_OK || gj_destroy(ctx) != GJ_OK) bad.fetch_add(1);
        } else {
            if (gj_destroy(ctx) != GJ_OK || gj_comm_destroy(c) != GJ_OK) bad.fetch_add(1);
        }
    };
    {
        std::thread a(rank_main, 0), b(rank_main, 1);
        a.join();
        b.join();
    }
    CHECK(bad.load() == 0);
    fprintf(stderr, "comm (a) two ranks: done\n");
    // (b) a collective that is INSIDE its RCCL call when gj_comm_destroy (variant 0) or gj_destroy of its context
    //     (variant 1) arrives: the stand-in holds the call at a gate, the destroyer must wait for it (in_flight), and the
    //     call must return GJ_OK on a communicator that is still alive
    void* stub = dlopen(getenv("GPSJAM_RCCL"), RTLD_NOW | RTLD_NOLOAD);
    CHECK(stub != nullptr);
    auto gate = reinterpret_cast<void (*)(int)>(dlsym(stub, "rccl_stub_gate"));
    auto waiting = reinterpret_cast<int (*)()>(dlsym(stub, "rccl_stub_waiting"));
    CHECK(gate && waiting);
    for (int variant = 0; variant < 2; ++variant) {
        for (int rep = 0; rep < 4; ++rep) {
            OK(gj_comm_unique_id(id));
            gj_ctx* ctx = nullptr;
            gj_comm* c = nullptr;
            OK(gj_create(0, &ctx));
            OK(gj_comm_init_rank(ctx, id, 0, 1, &c));
            void *d_a = nullptr, *d_b = nullptr;
            OK(gj_malloc(ctx, 1024, &d_a));
            OK(gj_malloc(ctx, 1024, &d_b));
            fprintf(stderr, "comm (b) variant %d rep %d\n", variant, rep);
            gate(1);
            std::atomic<int> rc_call{-99}, rc_destroy{-99};
            std::thread caller_t([&] { rc_call.store(rep & 1 ? gj_comm_gather_dev(c, d_a, 1024, d_b, 0) : gj_comm_allgather_dev(c, d_a, 1024, d_b)); });
            while (waiting() == 0) sched_yield();             // the call is inside RCCL now, holding no lock
            std::thread destroyer([&] { rc_destroy.store(variant == 0 ? gj_comm_destroy(c) : gj_destroy(ctx)); });
            usleep(1500);                                        // the destroyer is spinning in comm_quiesce
            CHECK(rc_destroy.load() == -99);                     // ... and has not gone past the call in flight
            gate(0);
            caller_t.join();
            destroyer.join();
            CHECK(rc_call.load() == GJ_OK && rc_destroy.load() == GJ_OK);
            if (variant == 0) {
                OK(gj_free(ctx, d_a));
                OK(gj_free(ctx, d_b));
                OK(gj_destroy(ctx));
            } else {
                CHECK(gj_comm_allgather_dev(c, d_a, 1024, d_b) == GJ_ERR_INVALID);   // its context is gone: refused, not crashed
                OK(gj_comm_destroy(c));                          // frees the handle only
                OK(gj_create(0, &ctx));
                OK(gj_free(ctx, d_a));
                OK(gj_free(ctx, d_b));
                OK(gj_destroy(ctx));
            }
        }
    }
    // (c) gj_destroy of the context while another thread keeps STARTING collectives on its communicator (allowed: the
    //     handle outlives the context); the hammer is told to stop by the GJ_ERR_INVALID of the detached communicator
    for (int rep = 0; rep < 8; ++rep) {
        fprintf(stderr, "comm (c) rep %d\n", rep);
        OK(gj_comm_unique_id(id));
        gj_ctx* ctx = nullptr;
        gj_comm* c = nullptr;
        OK(gj_create(0, &ctx));
        OK(gj_comm_init_rank(ctx, id, 0, 1, &c));
        void *d_a = nullptr, *d_b = nullptr;
        OK(gj_malloc(ctx, 1024, &d_a));
        OK(gj_malloc(ctx, 1024, &d_b));
        std::atomic<int> calls{0}, refused{0}, go{0};
        std::thread hammer([&] {
            go.store(1);
            for (;;) {
                const int rc = gj_comm_allgather_dev(c, d_a, 1024, d_b);
                if (rc == GJ_ERR_INVALID) { refused.fetch_add(1); break; }
                if (rc != GJ_OK) { bad.fetch_add(1); break; }
                calls.fetch_add(1);
            }
        });
        while (!go.load()) sched_yield();
        usleep(100 + 120 * (unsigned)rep);
        OK(gj_destroy(ctx));   // detaches, waits for the call in flight AND for the collectives already queued, then destroys
        hammer.join();
        CHECK(refused.load() == 1);
        OK(gj_comm_destroy(c));
        OK(gj_create(0, &ctx));
        OK(gj_free(ctx, d_a));
        OK(gj_free(ctx, d_b));
        OK(gj_destroy(ctx));
    }
    CHECK(bad.load() == 0);
    CHECK(hip_stub_live_allocations() == 0);
    return 0;
}

// ------------------------------------------------------------------------------------------------------------------
// 5. abandoned callers: threads that leave the library at one of its wait sites WITHOUT unwinding and then end -- the
//    lane's owner token and (site 5) the context mutex still locked, the lane still marked busy: the state a killed
//    caller leaves behind (tests/c_abandoned_caller.c produces it on the GPU with a raw exit system call; the
//    sanitizer runtimes keep per-thread state that a raw exit corrupts -- ASan aborts in asan_thread.cpp -- so here
//    the thread longjmps out of the hook to its start function and returns: no destructor runs, the kernel finds the
//    robust mutexes held at thread exit exactly as after a kill).  Other threads keep calling; the lanes come back
//    through the sweep of a later check-out or of gj_debug_counters -- unlocked by the sweeping thread that took them
//    over (lane_owner_take -> lane_recover -> lane_checkin).
//    SAN_LANES_PAUSE=1 (the ThreadSanitizer run): the live callers pause while a doomed thread dies and is joined.
//    TSan models a mutex hand-over by its unlock; a dead owner never unlocks, so what it wrote to its lane looks
//    unordered to the thread that inherits the mutex with EOWNERDEAD -- the kernel orders it (the robust list is walked
//    at thread exit, before the futex wake), TSan cannot know.  With the pause the order also runs through the join.
// ------------------------------------------------------------------------------------------------------------------
#include <csetjmp>
static thread_local int doomed_site = 0;
static thread_local jmp_buf doomed_jb;
static void dying_hook(void*, int site) {
    if (doomed_site && site == doomed_site) longjmp(doomed_jb, 1);
}

static int scenario_lanes() {
    gj_ctx* ctx = nullptr;
    OK(gj_create(0, &ctx));
    OK(gj_debug_set_wait_hook(ctx, dying_hook, nullptr));
    const size_t big = 8u << 20;
    std::vector<uint8_t> cap = make_capture(big, 11);
    std::atomic<int> bad{0}, stop{0}, pause{0}, paused{0};
    const bool pausing = getenv("SAN_LANES_PAUSE") != nullptr;
    // live callers beside the dying ones
    std::vector<std::thread> live;
    for (int t = 0; t < 3; ++t)
        live.emplace_back([&, t] {
            std::vector<float> power(256);
            while (!stop.load()) {
                if (pause.load()) {
                    paused.fetch_add(1);
                    while (pause.load()) usleep(50);
                    paused.fetch_sub(1);
                }
                size_t n = 0;
                float ms = 0.f;
                if (gj_chunk_power_u8(ctx, cap.data(), (size_t)(1u << 20) + 2 * (size_t)t, 65536, 0.f, 0, power.data(), power.size(), &n, &ms) != GJ_OK) bad.fetch_add(1);
                int lanes, busy, reclaimed, deaths;
                if ((t == 0) && gj_debug_counters(ctx, &lanes, &busy, &reclaimed, &deaths) != GJ_OK) bad.fetch_add(1);
            }
        });
    struct Doomed {
        gj_ctx* ctx;
        const uint8_t* cap;
        size_t big;
        int site;
    };
    auto doomed = [](void* p) -> void* {
        Doomed* d = static_cast<Doomed*>(p);
        doomed_site = d->site;
        if (setjmp(doomed_jb)) return nullptr;    // came out of the hook: end the thread with everything still held
        if (d->site == 3) {                       // inside a staged upload, before its fill threads start
            void* dev = nullptr;
            (void)gj_upload(d->ctx, d->cap, d->big, &dev);
        } else if (d->site == 1) {                // waiting for its event
            float pw[256], ms;
            size_t n;
            (void)gj_chunk_power_u8(d->ctx, d->cap, 1u << 20, 65536, 0.f, 0, pw, 256, &n, &ms);
        } else {                                  // as the owner of the context mutex
            int a, b, c, e;
            (void)gj_debug_counters(d->ctx, &a, &b, &c, &e);
        }
        return (void*)1;                          // not reached
    };
    const int sites[3] = {3, 1, 5};
    for (int k = 0; k < 18; ++k) {
        Doomed d{ctx, cap.data(), big, sites[k % 3]};
        pthread_t t;
        void* ret = nullptr;
        if (pausing) {
            pause.store(1);
            while (paused.load() < 3) usleep(50);
        }
        CHECK(pthread_create(&t, nullptr, doomed, &d) == 0);
        CHECK(pthread_join(t, &ret) == 0);
        CHECK(ret != (void*)1);
        pause.store(0);
        if (k % 6 == 5) OK(gj_debug_inject(ctx, GJ_INJECT_OWNER_ALIVE, 1));   // one probe answers "alive": the lane comes back a sweep later
    }
    stop.store(1);
    for (auto& th : live) th.join();
    int lanes, busy, reclaimed, deaths;
    OK(gj_debug_counters(ctx, &lanes, &busy, &reclaimed, &deaths));
    OK(gj_debug_counters(ctx, &lanes, &busy, &reclaimed, &deaths));
    fprintf(stderr, "lanes %d busy %d reclaimed %d owner_deaths %d\n", lanes, busy, reclaimed, deaths);
    CHECK(busy == 0 && lanes <= 8 && reclaimed >= 12 && deaths >= 1 && bad.load() == 0);
    OK(gj_debug_set_wait_hook(ctx, nullptr, nullptr));
    OK(gj_destroy(ctx));
    return 0;
}

// ------------------------------------------------------------------------------------------------------------------
// 6. allocation failures: the n-th hipMalloc / hipHostMalloc of a call sequence fails, for n = 0 .. until the sequence
//    passes -- every entry point must come back with an error code (no crash, no use of what it did not get), the
//    context must stay usable, and gj_destroy must leave nothing allocated
// ------------------------------------------------------------------------------------------------------------------
static int scenario_alloc_failures(const char* tmpdir) {
    const size_t nbytes = (5u << 20) + 2;
    std::vector<uint8_t> cap = make_capture(nbytes, 21);
    const std::string path = std::string(tmpdir) + "/fail_cap.bin";
    {
        FILE* f = fopen(path.c_str(), "wb");
        CHECK(f && fwrite(cap.data(), 1, nbytes, f) == nbytes);
        fclose(f);
    }
    int passed_at = -1, failures = 0;
    for (int n = 0; n < 200 && passed_at < 0; ++n) {
        gj_ctx* ctx = nullptr;
        hip_stub_fail_alloc_after(n);
        int rc = gj_create(0, &ctx);
        std::vector<float> power(256), psd(4 * 1024);
        void* d = nullptr;
        gj_ingest_result res;
        const gj_ingest_plan plan = default_plan();
        size_t cnt = 0, rows = 0;
        float ms;
        gj_onset on;
        if (!rc) rc = gj_chunk_power_u8(ctx, cap.data(), nbytes, 65536, 0.f, 0, power.data(), power.size(), &cnt, &ms);
        if (!rc) rc = gj_welch_u8(ctx, cap.data(), nbytes, 2048000, 1024, 2.048e6, 0, psd.data(), nullptr, psd.size(), &rows, &ms);
        if (!rc) rc = gj_onset_u8(ctx, cap.data() + 1, nbytes - 1, 200000, 1000, 50.f, &on, &ms);
        if (!rc) rc = gj_ingest_u8(ctx, cap.data(), nbytes, &plan, power.data(), power.size(), psd.data(), nullptr, psd.size(), &res, &d);
        if (!rc) rc = gj_free(ctx, d), d = nullptr;
        if (!rc) rc = gj_ingest_file(ctx, path.c_str(), 0, 0, &plan, power.data(), power.size(), psd.data(), nullptr, psd.size(), &res, &d);
        if (!rc) rc = gj_free(ctx, d), d = nullptr;
        if (!rc) rc = gj_reserve(ctx, 64u << 20);
        hip_stub_fail_alloc_after(-1);
        if (rc == GJ_OK) passed_at = n;
        else {
            ++failures;
            CHECK(rc == GJ_ERR_NOMEM || rc == GJ_ERR_HIP);
            // the context (if it came to life) still works once memory is back
            if (ctx) OK(gj_chunk_power_u8(ctx, cap.data(), nbytes, 65536, 0.f, 0, power.data(), power.size(), &cnt, &ms));
        }
        if (d && ctx) OK(gj_free(ctx, d));
        OK(gj_destroy(ctx));
        CHECK(hip_stub_live_allocations() == 0);
    }
    fprintf(stderr, "allocation failures injected: %d, sequence passes from allocation %d on\n", failures, passed_at);
    CHECK(passed_at > 5);
    unlink(path.c_str());
    return 0;
}

// ------------------------------------------------------------------------------------------------------------------
// 7. every remaining entry point once, with valid arguments and a few invalid ones (ASan / UBSan over the host side of
//    the plans: a part's view, the combine plan's bounds-by-division, the batched K2 and pack descriptors, K5 over slots,
//    the acquisition search, the file upload) -- two threads at once so that TSan sees the same calls share a context
// ------------------------------------------------------------------------------------------------------------------
static int api_sweep_once(gj_ctx* ctx, const char* tmpdir, int who) {
    const size_t unit = 8192000, nbytes = 2 * unit + 65536 * 3 + 100;     // two whole 2-s units + a ragged end
    std::vector<uint8_t> cap = make_capture(nbytes, 40u + (unsigned)who);
    void* d_cap = nullptr;
    OK(gj_upload(ctx, cap.data(), nbytes, &d_cap));
    const uint8_t* dc = static_cast<const uint8_t*>(d_cap);
    char name[256];
    int cus = 0;
    uint64_t hbm = 0;
    OK(gj_device_info(ctx, name, sizeof(name), &cus, &hbm));
    OK(gj_device_identity(ctx, name, sizeof(name)));
    CHECK(strstr(name, "pci=") != nullptr);
    // a file upload with an offset that is not a page multiple
    {
        const std::string path = std::string(tmpdir) + "/sweep" + std::to_string(who) + ".bin";
        FILE* f = fopen(path.c_str(), "wb");
        CHECK(f && fwrite(cap.data(), 1, nbytes, f) == nbytes);
        fclose(f);
        void* d = nullptr;
        size_t got = 0;
        OK(gj_upload_file(ctx, path.c_str(), 4097, 5u << 20, &d, &got));
        CHECK(got == (5u << 20));
        OK(gj_free(ctx, d));
        CHECK(gj_upload_file(ctx, (path + ".missing").c_str(), 0, 0, &d, &got) == GJ_ERR_INVALID);
        unlink(path.c_str());
    }
    const size_t nch = gj_chunk_count(nbytes, 65536), rows = gj_welch_rows(nbytes, 2048000, 1024);
    void *d_pow, *d_psd, *d_psd2, *d_amp, *d_on, *d_st, *d_mask, *d_hist, *d_slots, *d_l, *d_p, *d_m, *d_vec, *d_tiles, *d_amp_part, *d_pairs;
    const size_t sb = gj_tdoa_slot_bytes(4096);
    OK(gj_malloc(ctx, 4 * nch, &d_pow));
    OK(gj_malloc(ctx, 4 * (rows + 1) * 1024, &d_psd));
    OK(gj_malloc(ctx, 4 * (rows + 1) * 1024, &d_psd2));
    OK(gj_malloc(ctx, 64, &d_amp));
    OK(gj_malloc(ctx, 64, &d_on));
    OK(gj_malloc(ctx, 64, &d_st));
    OK(gj_malloc(ctx, nch, &d_mask));
    OK(gj_malloc(ctx, 256 * 8, &d_hist));
    OK(gj_malloc(ctx, 3 * sb, &d_slots));
    OK(gj_malloc(ctx, 64, &d_l));
    OK(gj_malloc(ctx, 64, &d_p));
    OK(gj_malloc(ctx, 64, &d_m));
    OK(gj_malloc(ctx, 64, &d_pairs));
    const size_t vec_len = GJ_RESULT_HEADER + nch + 1024 + GJ_RESULT_PAIR_FIELDS * 3;
    OK(gj_malloc(ctx, 8 * vec_len * 3, &d_vec));
    OK(gj_malloc(ctx, 16 * (gj_amp_tile_count(nbytes) + 1), &d_tiles));
    OK(gj_malloc(ctx, 64, &d_amp_part));
    const int32_t pairs[6] = {0, 1, 0, 2, 1, 2};
    OK(gj_memcpy_h2d(ctx, d_pairs, pairs, sizeof(pairs)));
    OK(gj_timer_start(ctx));
    OK(gj_chunk_power_dev(ctx, dc, nbytes, 65536, 0.f, 0, static_cast<float*>(d_pow)));
    OK(gj_power_threshold_dev(ctx, static_cast<float*>(d_pow), nch, 5.f, 6.f, static_cast<float*>(d_st), static_cast<uint8_t*>(d_mask)));
    OK(gj_amp_stats_dev(ctx, dc + 3, nbytes - 3, 0.2f, static_cast<gj_amp_stats*>(d_amp)));
    OK(gj_onset_dev(ctx, dc, nbytes, 200000, 1000, 50.f, static_cast<gj_onset*>(d_on)));
    OK(gj_byte_histogram_dev(ctx, dc, nbytes, 2048000, 1024, 100, static_cast<uint64_t*>(d_hist)));
    float kms = 0.f, fms = 0.f;
    OK(gj_welch_timed_dev(ctx, dc, nbytes, 2048000, 1024, 2.048e6, GJ_WELCH_SHIFT, static_cast<float*>(d_psd), nullptr, &kms, &fms));
    {
        const uint8_t* caps[3] = {dc, dc, dc};
        float* out[3] = {static_cast<float*>(d_psd), static_cast<float*>(d_psd2), static_cast<float*>(d_psd)};
        OK(gj_welch_batch_dev(ctx, caps, 2, nbytes, 2048000, 1024, 2.048e6, 0, out));
        CHECK(gj_welch_batch_dev(ctx, caps, 17, nbytes, 2048000, 1024, 2.048e6, 0, out) == GJ_ERR_INVALID);
    }
    for (int a = 0; a < 3; ++a)
        OK(gj_tdoa_slot_dev(ctx, dc, nbytes, &static_cast<gj_onset*>(d_on)->start_index, 4096, static_cast<uint8_t*>(d_slots) + a * sb));
    OK(gj_xcorr_slots_dev(ctx, static_cast<uint8_t*>(d_slots), sb, 3, 4096, pairs, 3, static_cast<int32_t*>(d_l), static_cast<float*>(d_p),
                          static_cast<float*>(d_m)));
    CHECK(gj_xcorr_slots_dev(ctx, static_cast<uint8_t*>(d_slots), sb, 3, 4096, pairs, 0, static_cast<int32_t*>(d_l), static_cast<float*>(d_p),
                             static_cast<float*>(d_m)) == GJ_ERR_INVALID);
    OK(gj_pack_result_dev(ctx, nch, static_cast<float*>(d_pow), static_cast<float*>(d_st), static_cast<gj_amp_stats*>(d_amp),
                          static_cast<gj_onset*>(d_on), static_cast<float*>(d_psd), rows, 1024, 0, 3, 3, static_cast<int32_t*>(d_pairs),
                          static_cast<int32_t*>(d_l), static_cast<float*>(d_p), static_cast<float*>(d_m), static_cast<double*>(d_vec)));
    {
        gj_combine_capture c[3];
        memset(c, 0, sizeof(c));
        for (int a = 0; a < 3; ++a) {
            c[a].n_chunks = nch; c[a].rows = rows; c[a].antenna = a; c[a].n_pairs = a == 0 ? 3 : 0; c[a].pair_cap = 3;
            c[a].d_power = static_cast<float*>(d_pow); c[a].d_stats = static_cast<float*>(d_st); c[a].d_amp = static_cast<gj_amp_stats*>(d_amp);
            c[a].d_onset = static_cast<gj_onset*>(d_on); c[a].d_psd = static_cast<float*>(d_psd);
            c[a].d_out = static_cast<double*>(d_vec) + a * vec_len;
        }
        OK(gj_pack_results_dev(ctx, c, 3, 1024, static_cast<int32_t*>(d_pairs), static_cast<int32_t*>(d_l), static_cast<float*>(d_p), static_cast<float*>(d_m)));
        c[1].n_pairs = 4;
        CHECK(gj_pack_results_dev(ctx, c, 3, 1024, static_cast<int32_t*>(d_pairs), static_cast<int32_t*>(d_l), static_cast<float*>(d_p), static_cast<float*>(d_m)) == GJ_ERR_INVALID);
        // the combine plan's validation: one good list, then counts and strides chosen to wrap 64-bit products
        c[1].n_pairs = 0;
        for (int a = 0; a < 3; ++a) { c[a].n_tiles = gj_amp_tile_count(nbytes); c[a].total_bytes = nbytes; c[a].n_parts = 1;
            c[a].d_tiles = d_tiles; c[a].d_amp_parts = static_cast<gj_amp_part*>(d_amp_part); c[a].d_onset_parts = static_cast<gj_onset*>(d_on); }
        // (arena = d_vec only: the descriptors above point outside it, which the check must say)
        gj_combine_copy cp[2];
        memset(cp, 0, sizeof(cp));
        cp[0].src_byte = 0; cp[0].dst = (uint64_t)(uintptr_t)d_vec; cp[0].count = 16; cp[0].src_stride = 8; cp[0].kind = GJ_COPY_F64;
        cp[1] = cp[0];
        CHECK(gj_combine_plan_check(cp, 2, c, 3, 4096, d_vec, 8 * vec_len * 3, 1024, 1) != GJ_OK);       // arrays outside the arena
        cp[1].count = ~0ull / 8 + 2;                                                                       // count * 8 wraps
        CHECK(gj_combine_plan_check(cp, 2, c, 3, 4096, d_vec, 8 * vec_len * 3, 1024, 1) != GJ_OK);
        cp[1].count = 4; cp[1].src_stride = 0xffffffffu; cp[1].src_byte = ~0ull - 8;                       // offset + stride wraps
        CHECK(gj_combine_plan_check(cp, 2, c, 3, 4096, d_vec, 8 * vec_len * 3, 1024, 1) != GJ_OK);
        CHECK(gj_combine_plan_check(cp, 0, c, 3, 4096, d_vec, 8 * vec_len * 3, 1024, 1) == GJ_ERR_INVALID);
    }
    {   // a part of a split capture: its own range is the second 2-s unit, one tile of halo in front
        gj_part_view v;
        memset(&v, 0, sizeof(v));
        v.d_buf = dc + unit - 65536; v.buf_first_byte = unit - 65536; v.buf_bytes = nbytes - (unit - 65536);
        v.own_first_byte = unit; v.own_bytes = unit; v.total_bytes = nbytes; v.d_noise = dc;
        gj_scan_extra x;
        memset(&x, 0, sizeof(x));
        x.pct = 5.f; x.rise_db = 6.f; x.d_slot = static_cast<uint8_t*>(d_slots); x.slice_samples = 4096;
        OK(gj_part_capture_scan_dev(ctx, &v, 65536, 0.f, 0, static_cast<float*>(d_pow), 0.f, d_tiles, static_cast<gj_amp_part*>(d_amp_part), 200000,
                                    1000, 50.f, static_cast<gj_onset*>(d_on), &x));
        OK(gj_part_scan_dev(ctx, &v, 65536, 0.f, 0, static_cast<float*>(d_pow), 0.f, d_tiles, static_cast<gj_amp_part*>(d_amp_part), 200000, 1000,
                            50.f, static_cast<gj_onset*>(d_on)));
        CHECK(gj_part_welch_workspace(ctx, &v, 2048000, 1024) > 0);
        OK(gj_part_welch_dev(ctx, &v, 2048000, 1024, 2.048e6, 0, static_cast<float*>(d_psd), nullptr));
        OK(gj_part_slot_dev(ctx, &v, &static_cast<gj_onset*>(d_on)->start_index, 4096, static_cast<uint8_t*>(d_slots)));
        v.own_first_byte = unit + 2;                       // not on a chunk boundary
        CHECK(gj_part_scan_dev(ctx, &v, 65536, 0.f, 0, static_cast<float*>(d_pow), 0.f, d_tiles, static_cast<gj_amp_part*>(d_amp_part), 200000, 1000,
                               50.f, static_cast<gj_onset*>(d_on)) == GJ_ERR_UNSUPPORTED);
        OK(gj_amp_combine_dev(ctx, d_tiles, gj_amp_tile_count(nbytes), static_cast<gj_amp_part*>(d_amp_part), 1, nbytes, static_cast<gj_amp_stats*>(d_amp)));
        OK(gj_onset_combine_dev(ctx, static_cast<gj_onset*>(d_on), 1, static_cast<gj_onset*>(d_on)));
    }
    {
        gj_synth_params sp;
        memset(&sp, 0, sizeof(sp));
        sp.key_noise = 1; sp.key_common = 2; sp.jam_start = 100; sp.jam_end = 200; sp.noise_k = 100; sp.jam_k = 50;
        void* d_syn = nullptr;
        OK(gj_malloc(ctx, 2 << 20, &d_syn));
        OK(gj_synth_u8_dev(ctx, &sp, 0, 1 << 20, static_cast<uint8_t*>(d_syn)));
        OK(gj_free(ctx, d_syn));
    }
    OK(gj_probe_busy_dev(ctx, 0.1f));
    CHECK(gj_probe_busy_dev(ctx, 1000.f) == GJ_ERR_INVALID);
    float ms = 0.f;
    OK(gj_timer_stop(ctx, &ms));
    OK(gj_synchronize(ctx));
    for (void* p : {d_pow, d_psd, d_psd2, d_amp, d_on, d_st, d_mask, d_hist, d_slots, d_l, d_p, d_m, d_vec, d_tiles, d_amp_part, d_pairs, d_cap}) OK(gj_free(ctx, p));
    return 0;
}

static int scenario_api_sweep(const char* tmpdir) {
    gj_ctx* ctx = nullptr;
    OK(gj_create(0, &ctx));
    std::atomic<int> bad{0};
    std::thread a([&] { if (api_sweep_once(ctx, tmpdir, 0)) bad.fetch_add(1); });
    std::thread b([&] { if (api_sweep_once(ctx, tmpdir, 1)) bad.fetch_add(1); });
    a.join();
    b.join();
    CHECK(bad.load() == 0);
    OK(gj_destroy(ctx));
    CHECK(hip_stub_live_allocations() == 0);
    return 0;
}

int main(int argc, char** argv) {
    if (argc < 2) {
        fprintf(stderr, "usage: %s threads|ingest_files|workspace|comm|lanes|alloc_failures|api_sweep [tmpdir]\n", argv[0]);
        return 2;
    }
    const std::string s = argv[1];
    const char* tmp = argc > 2 ? argv[2] : "/tmp";
    int rc = 2;
    if (s == "threads") rc = scenario_threads();
    else if (s == "ingest_files") rc = scenario_ingest_files(tmp);
    else if (s == "workspace") rc = scenario_workspace();
    else if (s == "comm") rc = scenario_comm();
    else if (s == "lanes") rc = scenario_lanes();
    else if (s == "alloc_failures") rc = scenario_alloc_failures(tmp);
    else if (s == "api_sweep") rc = scenario_api_sweep(tmp);
    else fprintf(stderr, "unknown scenario %s\n", s.c_str());
    if (rc == 0) printf("%s: ok (%llu launches, %llu copies through the stand-in runtime)\n", s.c_str(), hip_stub_launches(), hip_stub_copies());
    return rc;
}
