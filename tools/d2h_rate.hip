// Development: rate of device -> host copies into PAGEABLE memory (what a numpy result array is): the runtime's own path into
// fresh (never touched) and into touched pages, against pinned double buffers drained into the destination by T host threads.
//   hipcc --offload-arch=gfx950 -O2 -o /tmp/d2h_rate tools/d2h_rate.hip -lpthread && /tmp/d2h_rate [MB]
#include <hip/hip_runtime.h>
#include <sys/mman.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

static void *fresh(size_t bytes) { return mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0); }

static void copy_threads(char *dst, const char *src, size_t bytes, int T) {
    if (T <= 1) { memcpy(dst, src, bytes); return; }
    std::vector<std::thread> th;
    const size_t per = ((bytes / T) + 4095) & ~(size_t)4095;
    for (int t = 0; t < T; ++t) {
        const size_t a = (size_t)t * per, b = a + per < bytes ? a + per : bytes;
        if (a < b) th.emplace_back([=] { memcpy(dst + a, src + a, b - a); });
    }
    for (auto &x : th) x.join();
}

int main(int argc, char **argv) {
    const size_t bytes = (size_t)(argc > 1 ? atoi(argv[1]) : 237) << 20;
    char *dev;
    CK(hipMalloc((void **)&dev, bytes));
    CK(hipMemset(dev, 7, bytes));
    hipStream_t st;
    CK(hipStreamCreate(&st));
    for (int rep = 0; rep < 2; ++rep) {
        char *h = (char *)fresh(bytes);
        double t = now();
        CK(hipMemcpyAsync(h, dev, bytes, hipMemcpyDeviceToHost, st));
        CK(hipStreamSynchronize(st));
        const double fresh_s = now() - t;
        t = now();
        CK(hipMemcpyAsync(h, dev, bytes, hipMemcpyDeviceToHost, st));
        CK(hipStreamSynchronize(st));
        const double touched_s = now() - t;
        printf("runtime copy into pageable memory: fresh pages %.1f ms (%.1f GB/s), touched pages %.1f ms (%.1f GB/s)\n", fresh_s * 1e3,
               bytes / fresh_s / 1e9, touched_s * 1e3, bytes / touched_s / 1e9);
        munmap(h, bytes);
    }
    const size_t CH = (size_t)8 << 20;
    char *pin[2];
    CK(hipHostMalloc((void **)&pin[0], CH, hipHostMallocDefault));
    CK(hipHostMalloc((void **)&pin[1], CH, hipHostMallocDefault));
    hipEvent_t ev[2];
    CK(hipEventCreate(&ev[0]));
    CK(hipEventCreate(&ev[1]));
    for (int T : {1, 2, 4, 8}) {
        for (int kind = 0; kind < 2; ++kind) {
            char *h = (char *)fresh(bytes);
            if (kind) memset(h, 1, bytes);
            const double t = now();
            const size_t nch = (bytes + CH - 1) / CH;
            CK(hipMemcpyAsync(pin[0], dev, bytes < CH ? bytes : CH, hipMemcpyDeviceToHost, st));
            CK(hipEventRecord(ev[0], st));
            for (size_t c = 0; c < nch; ++c) {
                const size_t off = c * CH, n = bytes - off < CH ? bytes - off : CH;
                if (c + 1 < nch) {
                    const size_t off2 = off + CH, n2 = bytes - off2 < CH ? bytes - off2 : CH;
                    CK(hipMemcpyAsync(pin[(c + 1) & 1], dev + off2, n2, hipMemcpyDeviceToHost, st));
                    CK(hipEventRecord(ev[(c + 1) & 1], st));
                }
                CK(hipEventSynchronize(ev[c & 1]));
                copy_threads(h + off, pin[c & 1], n, T);
            }
            const double s = now() - t;
            printf("pinned 2 x 8 MB + %d thread(s), %s pages: %.1f ms (%.1f GB/s)\n", T, kind ? "touched" : "fresh", s * 1e3, bytes / s / 1e9);
            if (h[bytes - 1] != 7 || h[12345] != 7) printf("  WRONG DATA\n");
            munmap(h, bytes);
        }
    }
    // all of it into one pinned buffer: the link alone
    char *big;
    CK(hipHostMalloc((void **)&big, bytes, hipHostMallocDefault));
    double t = now();
    CK(hipMemcpyAsync(big, dev, bytes, hipMemcpyDeviceToHost, st));
    CK(hipStreamSynchronize(st));
    double s = now() - t;
    printf("into pinned memory: %.1f ms (%.1f GB/s)\n", s * 1e3, bytes / s / 1e9);
    return 0;
}
