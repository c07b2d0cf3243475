// libalproj_hip.so -- the HIP-free translation unit: error state, pose folding and the threaded host helpers declared in
// host/alp_host.h, plus the three alp_host_* entry points of the ABI.  Compiled by hipcc (as plain C++) into the library
// and by g++ under the sanitizers into build/host_san/ (alproj_amd/_build.py: build_host).
#include "host/alp_host.h"

#include <sys/mman.h>

namespace alp {

// ------------------------------------------------------------------ errors
static thread_local char g_err[2048] = "";

void set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

int fail(int code, const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

// ------------------------------------------------------------------ pose folding
// Reference arithmetic being folded (all float64, src/alproj/optimize.py):
//   intrinsic_mat :35-38   fov_x = fov*pi/180; fov_y = fov_x*h/w (Q5);
//                          fx = w/(2 tan(fov_x/2)); fy = h/(2 tan(fov_y/2))
//   extrinsic_mat :71-95   R = Rx(-(tilt+90)) . Ry(-roll) . Rz(pan);  t = R.(-cam)
//   project :144-149       cam = R.p + t;  (x,y,z) = K.cam;  u = w - x/z (Q4);  v = y/z
//   _distort :104-106      c = float32((w-1)/2, (h-1)/2);  x1 = (u-c0)/c0;  y1 = (v-c1)/c1
// With p = origin + q:  cam = R.q + R.(origin - cam_pos), and
//   x1 = ((w-c0)/c0) - (x/z)/c0 = ( ((w-c0)/c0).rowZ - rowx/c0 ) . [q;1] / (rowZ.[q;1])
//   y1 = (y/z)/c1 - 1          = ( rowy/c1 - rowZ ) . [q;1] / (rowZ.[q;1])
// where rowx = fx.R0 + cx.R2, rowy = fy.R1 + cy.R2, rowZ = R2 (4-vectors incl. translation).
// The principal-point cancellation (cx.Z against c0.Z) therefore happens here in float64.
void fold_pose(const double p[ALP_NPARAM], const double origin[3], double rec[POSE_WORDS]) {
    const double X = p[0], Y = p[1], Z = p[2], fov = p[3], pan_d = p[4], tilt_d = p[5],
                 roll_d = p[6];
    const double w = p[21], h = p[22], cx = p[23], cy = p[24];
    const double pi = M_PI;

    const double fov_x = fov * pi / 180;
    const double fov_y = fov_x * h / w;
    const double fx = w / (2 * std::tan(fov_x / 2));
    const double fy = h / (2 * std::tan(fov_y / 2));

    const double a = pan_d * pi / 180;
    const double b = -(tilt_d + 90) * pi / 180;
    const double c = -roll_d * pi / 180;
    const double rz[3][3] = {{std::cos(a), -std::sin(a), 0}, {std::sin(a), std::cos(a), 0}, {0, 0, 1}};
    const double rx[3][3] = {{1, 0, 0}, {0, std::cos(b), -std::sin(b)}, {0, std::sin(b), std::cos(b)}};
    const double ry[3][3] = {{std::cos(c), 0, std::sin(c)}, {0, 1, 0}, {-std::sin(c), 0, std::cos(c)}};
    double rxy[3][3], R[3][3];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            double s = 0;
            for (int k = 0; k < 3; ++k) s += rx[i][k] * ry[k][j];
            rxy[i][j] = s;
        }
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            double s = 0;
            for (int k = 0; k < 3; ++k) s += rxy[i][k] * rz[k][j];
            R[i][j] = s;
        }
    const double d[3] = {origin[0] - X, origin[1] - Y, origin[2] - Z};
    double row[3][4];
    for (int i = 0; i < 3; ++i) {
        for (int j = 0; j < 3; ++j) row[i][j] = R[i][j];
        row[i][3] = R[i][0] * d[0] + R[i][1] * d[1] + R[i][2] * d[2];
    }
    const double c0 = (double)(float)((w - 1) / 2);
    const double c1 = (double)(float)((h - 1) / 2);
    const double A = (w - c0) / c0;
    for (int j = 0; j < 4; ++j) {
        const double rowx = fx * row[0][j] + cx * row[2][j];
        const double rowy = fy * row[1][j] + cy * row[2][j];
        rec[0 + j] = A * row[2][j] - rowx / c0;
        rec[4 + j] = rowy / c1 - row[2][j];
        rec[8 + j] = row[2][j];
    }
    for (int i = 0; i < 6; ++i) rec[12 + i] = p[9 + i];   // k1..k6
    rec[18] = 1 + p[7];                                    // 1 + a1
    rec[19] = 1 + p[8];                                    // 1 + a2
    rec[20] = 2 * p[15];                                   // 2 p1
    rec[21] = 2 * p[16];                                   // 2 p2
    for (int i = 0; i < 4; ++i) rec[22 + i] = p[17 + i];  // s1..s4
    rec[26] = c0;
    rec[27] = c1;
    rec[28] = -c0;                                         // residual = (uo - c0) + (-c0) * x1_d
    rec[29] = -c1;
    rec[30] = rec[31] = 0;
}

bool pose_is_lens_free(const double p[ALP_NPARAM]) {
    for (int i = 9; i <= 20; ++i)
        if (p[i] != 0.0) return false;             // k1..k6, p1, p2, s1..s4 (NaN compares unequal: not lens-free)
    return true;
}

void lens_free_from_general(const double g[POSE_WORDS], double rec[POSE_WORDS]) {
    const double sy = g[18] / g[19];               // (1 + a1) / (1 + a2)  (a2 = -1: inf / NaN rows, and losses, like the reference's division)
    for (int i = 0; i < POSE_WORDS; ++i) rec[i] = 0;
    for (int j = 0; j < 4; ++j) {
        rec[0 + j] = g[28] * g[0 + j];             // -c0 X'
        rec[4 + j] = (g[29] * sy) * g[4 + j];      // -c1 (1 + a1) / (1 + a2) Y'
        rec[8 + j] = g[8 + j];
    }
    rec[26] = g[26];
    rec[27] = g[27];
}

void fold_pose_lens_free(const double p[ALP_NPARAM], const double origin[3], double rec[POSE_WORDS]) {
    double g[POSE_WORDS];
    fold_pose(p, origin, g);
    lens_free_from_general(g, rec);
}

namespace host {

// ------------------------------------------------------------------ content hash of a host array
// Four independent lanes of the xxHash64 round (acc = rotl(acc + w * P2, 31) * P1: a bijection of acc for a fixed
// word and of the word for a fixed acc, so a change of ONE 8-byte word always changes the digest; several changed
// words collide with probability 2^-64), one contiguous slice per thread, slice digests chained in order.
namespace {
constexpr uint64_t HP1 = 0x9E3779B185EBCA87ull, HP2 = 0xC2B2AE3D27D4EB4Full, HP3 = 0x165667B19E3779F9ull;
inline uint64_t rotl64(uint64_t x, int r) { return (x << r) | (x >> (64 - r)); }
inline uint64_t hround(uint64_t acc, uint64_t w) { return rotl64(acc + w * HP2, 31) * HP1; }
inline uint64_t avalanche(uint64_t h) {
    h ^= h >> 33; h *= HP2; h ^= h >> 29; h *= HP3; h ^= h >> 32;
    return h;
}
uint64_t hash_slice(const unsigned char *p, size_t n, uint64_t seed) {
    uint64_t a0 = seed + HP1 + HP2, a1 = seed + HP2, a2 = seed, a3 = seed - HP1;
    size_t i = 0;
    for (; i + 32 <= n; i += 32) {
        uint64_t w[4];
        memcpy(w, p + i, 32);
        a0 = hround(a0, w[0]); a1 = hround(a1, w[1]); a2 = hround(a2, w[2]); a3 = hround(a3, w[3]);
    }
    uint64_t h = rotl64(a0, 1) + rotl64(a1, 7) + rotl64(a2, 12) + rotl64(a3, 18);
    h = (h ^ hround(0, a0)) * HP1 + HP3; h = (h ^ hround(0, a1)) * HP1 + HP3;
    h = (h ^ hround(0, a2)) * HP1 + HP3; h = (h ^ hround(0, a3)) * HP1 + HP3;
    for (; i + 8 <= n; i += 8) {
        uint64_t w;
        memcpy(&w, p + i, 8);
        h = rotl64(h ^ hround(0, w), 27) * HP1 + HP3;
    }
    for (; i < n; ++i) h = rotl64(h ^ (p[i] * HP3), 11) * HP1;
    return avalanche(h + (uint64_t)n);
}
}  // namespace

uint64_t hash64(const void *buf, int64_t bytes, int threads) {
    const unsigned char *p = (const unsigned char *)buf;
    int T = threads > 0 ? threads : (int)std::thread::hardware_concurrency();
    if (T < 1) T = 1;
    if (T > 64) T = 64;
    const int64_t SL = HASH_SLICE_BYTES;                        // fixed slices: the digest does not depend on the thread count
    const int64_t ns = bytes > 0 ? (bytes + SL - 1) / SL : 1;
    if ((int64_t)T > ns) T = (int)ns;
    std::vector<uint64_t> part((size_t)ns);
    auto run = [&](int t) {
        for (int64_t s = t; s < ns; s += T) {
            const int64_t lo = s * SL, hi = (lo + SL < bytes) ? lo + SL : bytes;
            part[(size_t)s] = hash_slice(p + lo, (size_t)(hi > lo ? hi - lo : 0), (uint64_t)s);
        }
    };
    std::vector<std::thread> th;
    for (int t = 1; t < T; ++t) th.emplace_back(run, t);
    run(0);
    for (auto &x : th) x.join();
    uint64_t h = HP3 ^ (uint64_t)bytes;
    for (int64_t s = 0; s < ns; ++s) h = rotl64(h ^ hround(0, part[(size_t)s]), 27) * HP1 + HP3;
    return avalanche(h);
}

void minmax(const double *values, int64_t n, int threads, double out[2]) {
    int T = threads > 0 ? threads : (int)std::min(8u, std::max(1u, std::thread::hardware_concurrency()));
    if (T > 64) T = 64;
    const int64_t per = MINMAX_VALUES_PER_THREAD;                // a thread is worth starting for a million values
    if ((int64_t)T > (n + per - 1) / per) T = (int)((n + per - 1) / per);
    std::vector<double> lo((size_t)T, INFINITY), hi((size_t)T, -INFINITY);
    std::vector<char> nan((size_t)T, 0);
    auto run = [&](int t) {
        const int64_t a = n * t / T, b = n * (t + 1) / T;
        // eight independent chains (the compare-and-select is a dependency; the compiler turns the inner loop into vector min / max)
        double l[8], h[8];
        for (int k = 0; k < 8; ++k) { l[k] = INFINITY; h[k] = -INFINITY; }
        int bad = 0;
        int64_t i = a;
        for (; i + 8 <= b; i += 8)
            for (int k = 0; k < 8; ++k) {
                const double v = values[i + k];
                bad |= v != v;
                l[k] = v < l[k] ? v : l[k];
                h[k] = v > h[k] ? v : h[k];
            }
        for (; i < b; ++i) {
            const double v = values[i];
            bad |= v != v;
            l[0] = v < l[0] ? v : l[0];
            h[0] = v > h[0] ? v : h[0];
        }
        double l0 = l[0], h0 = h[0];
        for (int k = 1; k < 8; ++k) { l0 = l[k] < l0 ? l[k] : l0; h0 = h[k] > h0 ? h[k] : h0; }
        lo[(size_t)t] = l0;
        hi[(size_t)t] = h0;
        nan[(size_t)t] = (char)bad;
    };
    std::vector<std::thread> th;
    for (int t = 1; t < T; ++t) th.emplace_back(run, t);
    run(0);
    for (auto &x : th) x.join();
    double l = lo[0], h = hi[0];
    bool bad = nan[0];
    for (int t = 1; t < T; ++t) {
        l = lo[(size_t)t] < l ? lo[(size_t)t] : l;
        h = hi[(size_t)t] > h ? hi[(size_t)t] : h;
        bad |= nan[(size_t)t] != 0;
    }
    out[0] = bad ? NAN : l;
    out[1] = bad ? NAN : h;
}

#ifndef MADV_POPULATE_WRITE
#define MADV_POPULATE_WRITE 23          // Linux 5.14
#endif
void prefault(void *buf, int64_t bytes, int threads) {
    const uintptr_t PG = 4096, HP = (uintptr_t)2 << 20;
    const uintptr_t a = ((uintptr_t)buf + PG - 1) & ~(PG - 1), b = ((uintptr_t)buf + (uintptr_t)bytes) & ~(PG - 1);
    if (b <= a) return;
    // advice only: where the kernel refuses either call (huge pages off, a kernel before 5.14) the pages come into being
    // one fault at a time during the copy, as they did before
    madvise((void *)a, b - a, MADV_HUGEPAGE);
    int T = threads > 0 ? threads : 4;                           // 4: 94 GB/s on the bench host; 8 and 16 fall back to 35 (tools/prefault_rate.cpp)
    if (T > 64) T = 64;
    const uintptr_t span = (((b - a) / (uintptr_t)T) + HP - 1) & ~(HP - 1);      // shares end on 2 MB boundaries of the address space
    auto run = [&](int t) {
        uintptr_t lo = t == 0 ? a : ((a + span * (uintptr_t)t) & ~(HP - 1)), hi = t == T - 1 ? b : ((a + span * (uintptr_t)(t + 1)) & ~(HP - 1));
        if (lo < a) lo = a;
        if (hi > b) hi = b;
        if (lo < hi) madvise((void *)lo, hi - lo, MADV_POPULATE_WRITE);
    };
    std::vector<std::thread> th;
    for (int t = 1; t < T; ++t) th.emplace_back(run, t);
    run(0);
    for (auto &x : th) x.join();
}

}  // namespace host
}  // namespace alp

using namespace alp;

extern "C" {

const char *alp_last_error(void) { return g_err; }

int alp_host_hash64(const void *buf, int64_t bytes, int threads, uint64_t *digest) {
    ALP_REQUIRE(digest && bytes >= 0 && (bytes == 0 || buf), "bad argument");
    *digest = host::hash64(buf, bytes, threads);
    return ALP_OK;
}

int alp_host_minmax(const double *values, int64_t n, int threads, double out[2]) {
    ALP_REQUIRE(values && out && n >= 1, "bad argument");
    host::minmax(values, n, threads, out);
    return ALP_OK;
}

int alp_host_prefault(void *buf, int64_t bytes, int threads) {
    ALP_REQUIRE(bytes >= 0 && (bytes == 0 || buf), "bad argument");
    host::prefault(buf, bytes, threads);
    return ALP_OK;
}

}  // extern "C"
