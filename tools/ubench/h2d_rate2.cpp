// Round 3: can a file in the page cache reach device memory faster than the ~20 GB/s of tools/ubench/h2d_rate.cpp (one extra copy
// through pinned staging)? Tried here: (A) the ceiling — hipMemcpy from hipHostMalloc'ed memory; (B) hipHostRegister on the
// populated mapping (whole file, then DMA straight from the page-cache pages); (C) the same pipelined in chunks, registration of
// chunk k+1 on a helper thread under the DMA of chunk k; (D) many reader threads with small pinned chunks; (E) O_DIRECT reads into
// pinned chunks (storage, not page cache). Build on the GPU box: hipcc -O2 -o /tmp/h2d_rate2 tools/ubench/h2d_rate2.cpp -lpthread
#include <hip/hip_runtime.h>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
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
    {   // A: ceiling
        uint8_t* pin; CK(hipHostMalloc((void**)&pin, n, hipHostMallocDefault));
        memset(pin, 1, n);
        CK(hipMemcpy(dev, pin, n, hipMemcpyHostToDevice));
        double t0 = now();
        CK(hipMemcpy(dev, pin, n, hipMemcpyHostToDevice));
        double t1 = now();
        printf("A  hipMemcpy from pinned memory (ceiling): %.2f GB/s\n", n / (t1 - t0) / 1e9);
        CK(hipHostFree(pin));
    }
    for (int shared = 0; shared < 2; ++shared) {   // B: register the whole mapping
        void* m = mmap(nullptr, n, PROT_READ, shared ? MAP_SHARED : MAP_PRIVATE, fd, 0);
        double t0 = now();
#ifdef MADV_POPULATE_READ
        madvise(m, n, MADV_POPULATE_READ);
#endif
        double t1 = now();
        hipError_t e = hipHostRegister(m, n, hipHostRegisterDefault);
        double t2 = now();
        if (e != hipSuccess) { printf("B  hipHostRegister(%s mapping) failed: %s\n", shared ? "MAP_SHARED" : "MAP_PRIVATE", hipGetErrorString(e)); (void)hipGetLastError(); munmap(m, n); continue; }
        CK(hipMemcpy(dev, m, n, hipMemcpyHostToDevice));
        double t3 = now();
        CK(hipHostUnregister(m));
        double t4 = now();
        printf("B  %s mapping: populate %.3f s, hipHostRegister %.3f s (%.1f GB/s), hipMemcpy %.3f s (%.1f GB/s), unregister %.3f s; all %.2f GB/s\n",
               shared ? "MAP_SHARED" : "MAP_PRIVATE", t1 - t0, t2 - t1, n / (t2 - t1) / 1e9, t3 - t2, n / (t3 - t2) / 1e9, t4 - t3, n / (t4 - t0) / 1e9);
        munmap(m, n);
    }
    for (size_t chunk_mb : {64, 256}) for (int helpers : {1, 2, 4}) {   // C: pipelined registration
        void* m = mmap(nullptr, n, PROT_READ, MAP_PRIVATE, fd, 0);
        const size_t CH = chunk_mb << 20, nch = (n + CH - 1) / CH;
        std::vector<std::atomic<int>> ready(nch);
        for (auto& r : ready) r = 0;
        std::atomic<size_t> next{0};
        std::atomic<int> failed{0};
        double t0 = now();
        std::vector<std::thread> th;
        for (int h = 0; h < helpers; ++h) th.emplace_back([&] {
            (void)hipSetDevice(0);
            for (;;) {
                const size_t c = next.fetch_add(1);
                if (c >= nch) break;
                uint8_t* p = (uint8_t*)m + c * CH;
                const size_t len = std::min(CH, n - c * CH);
#ifdef MADV_POPULATE_READ
                madvise(p, len, MADV_POPULATE_READ);
#endif
                if (hipHostRegister(p, len, hipHostRegisterDefault) != hipSuccess) { failed = 1; (void)hipGetLastError(); }
                ready[c] = 1;
            }
        });
        hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
        for (size_t c = 0; c < nch; ++c) {
            while (!ready[c]) std::this_thread::yield();
            const size_t len = std::min(CH, n - c * CH);
            CK(hipMemcpyAsync(dev + c * CH, (uint8_t*)m + c * CH, len, hipMemcpyHostToDevice, s));
        }
        CK(hipStreamSynchronize(s));
        double t1 = now();
        for (auto& x : th) x.join();
        for (size_t c = 0; c < nch; ++c) (void)hipHostUnregister((uint8_t*)m + c * CH);
        double t2 = now();
        printf("C  %zu MiB chunks registered by %d helper thread(s) ahead of the DMA: %.2f GB/s (with unregister %.2f GB/s)%s\n", chunk_mb, helpers,
               n / (t1 - t0) / 1e9, n / (t2 - t0) / 1e9, failed ? "  [some registrations FAILED]" : "");
        CK(hipStreamDestroy(s));
        munmap(m, n);
    }
    for (int T : {4, 8, 16}) for (size_t chunk_mb : {4, 16}) {   // D: reader threads, pinned chunks allocated before the clock starts
        const size_t CH = chunk_mb << 20;
        std::vector<uint8_t*> pins(T * 2);
        for (auto& p : pins) CK(hipHostMalloc((void**)&p, CH, hipHostMallocDefault));
        std::vector<std::thread> th;
        double t0 = now();
        for (int t = 0; t < T; ++t) th.emplace_back([&, t] {
            (void)hipSetDevice(0);
            hipStream_t s; (void)hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
            hipEvent_t ev[2];
            for (int k = 0; k < 2; ++k) (void)hipEventCreateWithFlags(&ev[k], hipEventDisableTiming);
            const size_t lo = n / T * t, hi = t == T - 1 ? n : n / T * (t + 1);
            int i = 0;
            for (size_t off = lo; off < hi; off += CH, ++i) {
                const size_t len = std::min(CH, hi - off);
                const int k = i & 1;
                uint8_t* pin = pins[2 * t + k];
                if (i >= 2) (void)hipEventSynchronize(ev[k]);
                size_t have = 0;
                while (have < len) { ssize_t r = pread(fd, pin + have, len - have, (off_t)(off + have)); if (r <= 0) break; have += (size_t)r; }
                (void)hipMemcpyAsync(dev + off, pin, len, hipMemcpyHostToDevice, s);
                (void)hipEventRecord(ev[k], s);
            }
            (void)hipStreamSynchronize(s);
            for (int k = 0; k < 2; ++k) (void)hipEventDestroy(ev[k]);
            (void)hipStreamDestroy(s);
        });
        for (auto& x : th) x.join();
        double t1 = now();
        printf("D  %d reader thread(s), pread into pinned %zu MiB chunks: %.2f GB/s\n", T, chunk_mb, n / (t1 - t0) / 1e9);
        for (auto& p : pins) (void)hipHostFree(p);
    }
    {   // E: O_DIRECT
        int fdd = open(path, O_RDONLY | O_DIRECT);
        if (fdd < 0) printf("E  O_DIRECT open failed\n");
        else {
            const int T = 8; const size_t CH = (size_t)16 << 20;
            std::vector<uint8_t*> pins(T * 2);
            for (auto& p : pins) CK(hipHostMalloc((void**)&p, CH, hipHostMallocDefault));
            std::vector<std::thread> th;
            std::atomic<int> bad{0};
            double t0 = now();
            for (int t = 0; t < T; ++t) th.emplace_back([&, t] {
                (void)hipSetDevice(0);
                hipStream_t s; (void)hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
                hipEvent_t ev[2];
                for (int k = 0; k < 2; ++k) (void)hipEventCreateWithFlags(&ev[k], hipEventDisableTiming);
                const size_t per = (n / T) & ~(CH - 1);
                const size_t lo = per * t, hi = t == T - 1 ? n : per * (t + 1);
                int i = 0;
                for (size_t off = lo; off < hi; off += CH, ++i) {
                    const size_t len = std::min(CH, hi - off);
                    const int k = i & 1;
                    uint8_t* pin = pins[2 * t + k];
                    if (i >= 2) (void)hipEventSynchronize(ev[k]);
                    size_t have = 0;
                    while (have < len) { ssize_t r = pread(fdd, pin + have, (len - have + 4095) & ~(size_t)4095, (off_t)(off + have)); if (r <= 0) { if (r < 0) bad = 1; break; } have += (size_t)r; }
                    (void)hipMemcpyAsync(dev + off, pin, len, hipMemcpyHostToDevice, s);
                    (void)hipEventRecord(ev[k], s);
                }
                (void)hipStreamSynchronize(s);
            });
            for (auto& x : th) x.join();
            double t1 = now();
            printf("E  8 reader threads, O_DIRECT pread into pinned 16 MiB chunks: %.2f GB/s%s\n", n / (t1 - t0) / 1e9, bad ? "  [read errors]" : "");
        }
    }
    return 0;
}
