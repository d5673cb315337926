// Host-only probe behind profiles/r02_ingest.txt: N threads copy a file out of the page cache in 16-MiB pieces,
// with pread (mode 0) or through a mapping (mode 1), four times in a row; the mapping's munmap is timed apart.
//   g++ -O2 -pthread -o /tmp/probe tools/ingest_probe.cpp && /tmp/probe <file> <threads> <mode>
#include <fcntl.h>
#include <unistd.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main(int argc, char** argv) {
    const char* path = argv[1];
    int nthreads = atoi(argv[2]);
    int mode = atoi(argv[3]);   // 0 pread, 1 mmap+memcpy
    int fd = open(path, O_RDONLY);
    struct stat st; fstat(fd, &st);
    size_t n = st.st_size, piece = 16u << 20;
    std::vector<unsigned char*> bufs(nthreads);
    for (auto& b : bufs) { b = (unsigned char*)aligned_alloc(4096, piece); memset(b, 1, piece); }
    for (int rep = 0; rep < 4; ++rep) {
        double t0 = now();
        unsigned char* m = nullptr;
        if (mode == 1) m = (unsigned char*)mmap(nullptr, n, PROT_READ, MAP_PRIVATE, fd, 0);
        std::vector<std::thread> th;
        size_t np = (n + piece - 1) / piece;
        for (int t = 0; t < nthreads; ++t) th.emplace_back([&, t] {
            for (size_t p = t; p < np; p += nthreads) {
                size_t off = p * piece, len = n - off < piece ? n - off : piece;
                if (mode == 0) { size_t d = 0; while (d < len) { ssize_t k = pread(fd, bufs[t] + d, len - d, off + d); if (k <= 0) break; d += k; } }
                else memcpy(bufs[t], m + off, len);
            }
        });
        for (auto& x : th) x.join();
        double t1 = now();
        if (m) munmap(m, n);
        double t2 = now();
        printf("mode %d rep %d: copy %.1f ms, munmap %.1f ms\n", mode, rep, (t1 - t0) * 1e3, (t2 - t1) * 1e3);
    }
}
