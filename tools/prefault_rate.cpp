// Development: how fast can the pages of a fresh anonymous mapping (what a large np.empty is) be brought into existence --
// one thread touching every page, T threads touching their share, T threads calling madvise(MADV_POPULATE_WRITE) on their share,
// the same after madvise(MADV_HUGEPAGE)?
//   g++ -O2 -o /tmp/prefault_rate tools/prefault_rate.cpp -lpthread && /tmp/prefault_rate [MB]
#include <sys/mman.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>
#ifndef MADV_POPULATE_WRITE
#define MADV_POPULATE_WRITE 23
#endif
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main(int argc, char **argv) {
    const size_t bytes = (size_t)(argc > 1 ? atoi(argv[1]) : 800) << 20;
    for (int mode = 0; mode < 3; ++mode)
        for (int T : {1, 2, 4, 8, 16}) {
            char *p = (char *)mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
            const double t = now();
            if (mode == 2) madvise(p, bytes, MADV_HUGEPAGE);          // transparent huge pages where the kernel's policy is "madvise"
            std::vector<std::thread> th;
            int failed = 0;
            for (int k = 0; k < T; ++k)
                th.emplace_back([=, &failed] {
                    const size_t a = (bytes / T * k) & ~(size_t)4095, b = k == T - 1 ? bytes : (bytes / T * (k + 1)) & ~(size_t)4095;
                    if (mode == 0) for (size_t i = a; i < b; i += 4096) p[i] = 1;
                    else if (madvise(p + a, b - a, MADV_POPULATE_WRITE) != 0) failed = 1;
                });
            for (auto &x : th) x.join();
            const double s = now() - t;
            printf("%s, %2d thread(s): %.1f ms (%.1f GB/s)%s\n", mode == 2 ? "MADV_HUGEPAGE + POPULATE" : mode ? "MADV_POPULATE_WRITE" : "touch every page   ", T, s * 1e3, bytes / s / 1e9,
                   failed ? " -- madvise FAILED" : "");
            munmap(p, bytes);
        }
    return 0;
}
