// Host->HBM ingest options for the *_u8 entry points: pageable hipMemcpy, hipHostRegister +
// copy, staged through two pinned buffers (host memcpy overlapped with the DMA).
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <vector>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
    const size_t n = 1ull << 30;
    std::vector<unsigned char> host(n);
    for (size_t i = 0; i < n; i += 4096) host[i] = (unsigned char)i;
    void* d; (void)hipMalloc(&d, n);
    hipStream_t s; (void)hipStreamCreate(&s);
    for (int rep = 0; rep < 2; ++rep) {
        double t0 = now();
        (void)hipMemcpy(d, host.data(), n, hipMemcpyHostToDevice);
        double t1 = now();
        printf("pageable hipMemcpy          : %.1f ms  %.1f GB/s\n", (t1 - t0) * 1e3, n / (t1 - t0) / 1e9);
        t0 = now();
        (void)hipHostRegister(host.data(), n, hipHostRegisterDefault);
        double tr = now();
        (void)hipMemcpyAsync(d, host.data(), n, hipMemcpyHostToDevice, s);
        (void)hipStreamSynchronize(s);
        double tc = now();
        (void)hipHostUnregister(host.data());
        t1 = now();
        printf("register %.1f + copy %.1f + unregister %.1f = %.1f ms  %.1f GB/s\n", (tr - t0) * 1e3, (tc - tr) * 1e3,
               (t1 - tc) * 1e3, (t1 - t0) * 1e3, n / (t1 - t0) / 1e9);
        const size_t B = 32 << 20;
        void *p0, *p1; (void)hipHostMalloc(&p0, B); (void)hipHostMalloc(&p1, B);
        void* pin[2] = {p0, p1};
        hipEvent_t ev[2]; (void)hipEventCreate(&ev[0]); (void)hipEventCreate(&ev[1]);
        t0 = now();
        for (size_t off = 0, k = 0; off < n; off += B, ++k) {
            const size_t len = (n - off < B) ? n - off : B;
            if (k >= 2) (void)hipEventSynchronize(ev[k & 1]);
            memcpy(pin[k & 1], host.data() + off, len);
            (void)hipMemcpyAsync((char*)d + off, pin[k & 1], len, hipMemcpyHostToDevice, s);
            (void)hipEventRecord(ev[k & 1], s);
        }
        (void)hipStreamSynchronize(s);
        t1 = now();
        printf("staged 2 x 32 MiB pinned    : %.1f ms  %.1f GB/s\n", (t1 - t0) * 1e3, n / (t1 - t0) / 1e9);
        (void)hipHostFree(p0); (void)hipHostFree(p1);
    }
    return 0;
}
