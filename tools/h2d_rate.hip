// Host-to-device rates on the GPU box: pageable hipMemcpy, pinned hipMemcpy, and T threads copying
// pageable memory into pinned staging (what a pipelined upload can sustain).  Dev tool, not shipped.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
int main() {
    const size_t N = (size_t)1200 << 20;
    char *src = (char *)malloc(N);
    memset(src, 1, N);
    char *dev = nullptr, *pin = nullptr;
    CK(hipMalloc((void **)&dev, N));
    double t0 = now();
    CK(hipHostMalloc((void **)&pin, (size_t)256 << 20, hipHostMallocDefault));
    printf("hipHostMalloc 256 MB: %.1f ms\n", (now() - t0) * 1e3);
    memset(pin, 2, (size_t)256 << 20);
    for (int rep = 0; rep < 3; ++rep) {
        t0 = now();
        CK(hipMemcpy(dev, src, N, hipMemcpyHostToDevice));
        printf("pageable hipMemcpy 1.2 GB: %.1f GB/s\n", N / (now() - t0) / 1e9);
    }
    for (int rep = 0; rep < 3; ++rep) {
        t0 = now();
        for (int k = 0; k < 4; ++k) CK(hipMemcpyAsync(dev + ((size_t)k * 256 << 20), pin, (size_t)256 << 20, hipMemcpyHostToDevice, 0));
        CK(hipStreamSynchronize(0));
        printf("pinned hipMemcpyAsync 4 x 256 MB: %.1f GB/s\n", 4.0 * (256 << 20) / (now() - t0) / 1e9);
    }
    for (size_t chunk : {(size_t)4 << 20, (size_t)16 << 20, (size_t)48 << 20}) {
        t0 = now();
        const int reps = (int)(((size_t)1 << 30) / chunk);
        for (int k = 0; k < reps; ++k) CK(hipMemcpyAsync(dev + (size_t)k * chunk, pin + (k & 3) * chunk, chunk, hipMemcpyHostToDevice, 0));
        CK(hipStreamSynchronize(0));
        printf("pinned chunks of %zu MB: %.1f GB/s\n", chunk >> 20, (double)reps * chunk / (now() - t0) / 1e9);
    }
    for (int T : {1, 2, 4, 6, 8, 12, 16}) {
        const size_t per = ((size_t)256 << 20) / T;
        double best = 0;
        for (int rep = 0; rep < 3; ++rep) {
            t0 = now();
            for (int round = 0; round < 4; ++round) {
                std::vector<std::thread> th;
                for (int t = 0; t < T; ++t)
                    th.emplace_back([=] { memcpy(pin + t * per, src + ((size_t)round * 256 << 20) + t * per, per); });
                for (auto &x : th) x.join();
            }
            const double r = 4.0 * (256 << 20) / (now() - t0) / 1e9;
            best = r > best ? r : best;
        }
        printf("%2d threads pageable -> pinned: %.1f GB/s\n", T, best);
    }
    // hipHostRegister of the caller's buffer
    t0 = now();
    CK(hipHostRegister(src, N, hipHostRegisterDefault));
    const double treg = now() - t0;
    t0 = now();
    CK(hipMemcpy(dev, src, N, hipMemcpyHostToDevice));
    const double tc = now() - t0;
    t0 = now();
    CK(hipHostUnregister(src));
    printf("hipHostRegister 1.2 GB: %.1f ms, copy %.1f ms (%.1f GB/s), unregister %.1f ms\n", treg * 1e3, tc * 1e3, N / tc / 1e9, (now() - t0) * 1e3);
    printf("hardware threads: %u\n", std::thread::hardware_concurrency());
    return 0;
}
