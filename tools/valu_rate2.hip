// Development micro-benchmark: issue rate of the integer / compare / select instructions the render
// kernels are made of, next to v_fma_f32 (8 independent chains per lane, 8 waves per SIMD).
// hipcc --offload-arch=gfx950 -O3 tools/valu_rate2.hip -o build/valu_rate2 && build/valu_rate2
#include <hip/hip_runtime.h>
#include <cstdio>

template <int MODE>
__global__ __launch_bounds__(256) void k(int *out, int iters, int a, int b, float fa, float fb) {
    int x[8];
    float f[8];
    for (int i = 0; i < 8; ++i) { x[i] = threadIdx.x * 3 + i; f[i] = threadIdx.x * 1e-3f + i; }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if (MODE == 0) f[i] = __builtin_fmaf(f[i], fa, fb);
                if (MODE == 1) x[i] = x[i] + a;                                  // v_add_u32
                if (MODE == 2) x[i] = min(x[i], a) + b;                          // v_min_i32 + v_add (2 instr)
                if (MODE == 3) x[i] = __mul24(x[i], a) + b;                      // v_mad_i32_i24
                if (MODE == 4) x[i] = x[i] * a + b;                              // v_mul_lo_u32 + add (or mad_u64)
                if (MODE == 5) x[i] = (x[i] > a) ? x[i] - b : x[i] + b;          // cmp + cndmask + ...
                if (MODE == 6) x[i] = (x[i] >> 3) ^ a;                           // shift + xor
                if (MODE == 7) f[i] = __builtin_fminf(f[i], fa) + fb;            // v_min_f32 + v_add_f32
                if (MODE == 8) x[i] = (int)__builtin_rintf((float)x[i] * fa);    // cvt, mul, rndne, cvt
            }
    }
    int s = 0;
    for (int i = 0; i < 8; ++i) s += x[i] + (int)f[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int MODE>
void run(const char *name, double instr_per_op) {
    const int blocks = 2048, iters = 20000;
    int *out;
    (void)hipMalloc(&out, (size_t)blocks * 256 * 4);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    k<MODE><<<blocks, 256>>>(out, 100, 3, 7, 1.0001f, 0.5f);
    (void)hipEventRecord(e0);
    k<MODE><<<blocks, 256>>>(out, iters, 3, 7, 1.0001f, 0.5f);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    const double ops = (double)blocks * 4 * iters * 32 / 1024.0;       // per SIMD
    printf("%-34s %.3f ms  %.2f cycles per op (%.1f instr/op assumed -> %.2f cycles/instr)\n", name, ms,
           2.4e9 * ms * 1e-3 / ops, instr_per_op, 2.4e9 * ms * 1e-3 / ops / instr_per_op);
    (void)hipFree(out);
}

int main() {
    run<0>("v_fma_f32", 1);
    run<1>("v_add_u32", 1);
    run<2>("v_min_i32 + v_add_u32", 2);
    run<3>("v_mad_i32_i24", 1);
    run<4>("v_mul_lo_u32 + add", 2);
    run<5>("cmp + select + add/sub", 4);
    run<6>("shift + xor", 2);
    run<7>("v_min_f32 + v_add_f32", 2);
    run<8>("cvt + mul + rndne + cvt", 4);
    return 0;
}
