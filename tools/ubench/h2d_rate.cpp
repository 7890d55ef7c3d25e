// Host -> device rates for a file in the page cache (what `matchy match` does with its input): (1) one hipMemcpy straight from
// the mapping (pageable memory: the runtime stages / pins it), (2) the same after MADV_POPULATE_READ, (3) T threads, each with a
// stream and two pinned 16 MiB buffers of its own, copying its share chunk by chunk (memcpy into a pinned buffer, async DMA
// from it). Build on the GPU box: hipcc -O2 -o /tmp/h2d_rate tools/ubench/h2d_rate.cpp -lpthread ; run: /tmp/h2d_rate <file>
#include <hip/hip_runtime.h>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <thread>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main(int argc, char** argv) {
    const char* path = argv[1];
    int fd = open(path, O_RDONLY);
    struct stat sb; fstat(fd, &sb);
    const size_t n = (size_t)sb.st_size;
    uint8_t* dev; CK(hipMalloc(&dev, n));
    CK(hipMemset(dev, 0, n));
    for (int mode = 0; mode < 2; ++mode) {
        void* m = mmap(nullptr, n, PROT_READ, MAP_PRIVATE, fd, 0);
        double t0 = now();
#ifdef MADV_POPULATE_READ
        if (mode == 1) madvise(m, n, MADV_POPULATE_READ);
#endif
        double t1 = now();
        CK(hipMemcpy(dev, m, n, hipMemcpyHostToDevice));
        double t2 = now();
        printf("hipMemcpy from the mapping%s: %.2f GB/s (populate %.3f s, copy %.3f s)\n", mode ? " after MADV_POPULATE_READ" : "", n / (t2 - t0) / 1e9, t1 - t0, t2 - t1);
        munmap(m, n);
    }
    for (int T : {1, 2, 4, 8}) {
        for (int use_read = 0; use_read < 2; ++use_read) {
            void* m = mmap(nullptr, n, PROT_READ, MAP_PRIVATE, fd, 0);
            const size_t CH = (size_t)16 << 20;
            std::vector<std::thread> th;
            double t0 = now();
            for (int t = 0; t < T; ++t) th.emplace_back([&, t] {
                (void)hipSetDevice(0);
                hipStream_t s; (void)hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
                uint8_t* pin[2]; hipEvent_t ev[2];
                for (int k = 0; k < 2; ++k) { (void)hipHostMalloc((void**)&pin[k], CH, hipHostMallocDefault); (void)hipEventCreateWithFlags(&ev[k], hipEventDisableTiming); }
                const size_t lo = n / T * t, hi = t == T - 1 ? n : n / T * (t + 1);
                int i = 0;
                for (size_t off = lo; off < hi; off += CH, ++i) {
                    const size_t len = std::min(CH, hi - off);
                    const int k = i & 1;
                    if (i >= 2) (void)hipEventSynchronize(ev[k]);
                    if (use_read) { size_t have = 0; while (have < len) { ssize_t r = pread(fd, pin[k] + have, len - have, (off_t)(off + have)); if (r <= 0) break; have += (size_t)r; } }
                    else memcpy(pin[k], (const uint8_t*)m + off, len);
                    (void)hipMemcpyAsync(dev + off, pin[k], len, hipMemcpyHostToDevice, s);
                    (void)hipEventRecord(ev[k], s);
                }
                (void)hipStreamSynchronize(s);
                for (int k = 0; k < 2; ++k) { (void)hipHostFree(pin[k]); (void)hipEventDestroy(ev[k]); }
                (void)hipStreamDestroy(s);
            });
            for (auto& x : th) x.join();
            double t1 = now();
            printf("%d thread(s), pinned 16 MiB chunks filled by %s: %.2f GB/s\n", T, use_read ? "pread" : "memcpy from the mapping", n / (t1 - t0) / 1e9);
            munmap(m, n);
        }
    }
    return 0;
}
