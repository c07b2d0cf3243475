// Development micro-benchmark: throughput of 64-bit global atomicMax by address pattern within a
// wave-instruction (frame-sized buffer, 21 M words).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

// MODE 0: lane l -> consecutive words (8 lanes per 64-B line), wave base pseudo-random
// MODE 1: lane l -> one word in each of 64 different lines of a 64-line neighbourhood
// MODE 2: fully random words over the buffer
// MODE 4-8: alignment and lines per instruction (the printed rate counts 64 lanes per instruction)
// MODE 3: like 0 but only 8 lanes active per instruction (8 instructions for the same 64 words)
template <int MODE>
__global__ __launch_bounds__(256) void k(unsigned long long *buf, uint32_t nwords, int iters) {
    const uint32_t lane = threadIdx.x & 63;
    uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    uint32_t s = wave * 2654435761u + 12345u;
    for (int it = 0; it < iters; ++it) {
        s = s * 1664525u + 1013904223u;
        const uint32_t base = (s >> 4) % (nwords - 8192);
        const unsigned long long key = ((unsigned long long)(s | 1u) << 32) | lane;
        if (MODE == 0) atomicMax(&buf[base + lane], key);
        if (MODE == 1) atomicMax(&buf[base + lane * 8 + (s & 7)], key);
        if (MODE == 2) { uint32_t r = (s ^ (lane * 2246822519u)) * 3266489917u; atomicMax(&buf[(r >> 3) % nwords], key); }
        if (MODE == 4) atomicMax(&buf[(base & ~15u) + lane], key);            // 128-byte aligned
        if (MODE == 5) atomicMax(&buf[(base & ~7u) + lane], key);             // 64-byte aligned
        if (MODE == 6) { if (lane < 16) atomicMax(&buf[(base & ~15u) + lane], key); }   // one aligned 128-byte line per instruction
        if (MODE == 7) { if (lane < 8) atomicMax(&buf[(base & ~7u) + lane], key); }     // one aligned 64-byte line per instruction
        if (MODE == 8) { if (lane < 1) atomicMax(&buf[base], key); }                    // one word per instruction
        if (MODE == 3) for (int g = 0; g < 8; ++g) if ((lane >> 3) == g) atomicMax(&buf[base + lane], key);
    }
}

template <int MODE>
void run(const char *name, unsigned long long *buf, uint32_t nwords) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int blocks = 4096, iters = 64;
    k<MODE><<<blocks, 256>>>(buf, nwords, 4);
    hipEventRecord(e0);
    k<MODE><<<blocks, 256>>>(buf, nwords, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double n = (double)blocks * 256 * iters;
    printf("%-44s %.3f ms  %.1f G atomics/s\n", name, ms, n / ms / 1e6);
}

int main() {
    const uint32_t nwords = 5616u * 3744u;
    unsigned long long *buf; hipMalloc(&buf, (size_t)nwords * 8); hipMemset(buf, 0, (size_t)nwords * 8);
    run<0>("64 consecutive words per instruction", buf, nwords);
    run<1>("64 words in 64 neighbouring lines", buf, nwords);
    run<2>("64 random words", buf, nwords);
    run<3>("same 64 words, 8 lanes per instruction", buf, nwords);
    run<4>("64 consecutive words, 128-B aligned", buf, nwords);
    run<5>("64 consecutive words, 64-B aligned", buf, nwords);
    run<6>("16 lanes: one aligned 128-B line (x4 = instr/s)", buf, nwords);
    run<7>("8 lanes: one aligned 64-B line (x8 = instr/s)", buf, nwords);
    run<8>("1 lane (x64 = instr/s)", buf, nwords);
    return 0;
}
