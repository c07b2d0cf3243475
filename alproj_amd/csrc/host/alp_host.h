// Host-side code of libalproj_hip.so that needs no HIP: error reporting, the float64 folding of a camera pose, the
// threaded helpers behind alp_host_hash64 / alp_host_minmax / alp_host_prefault, the threads that recognise a regular
// grid in a host index array, the workers that widen / narrow a fetched plane, and the argmin / confirmation-band
// selection of alp_eval_population_wait.
//
// It is one header + host/alp_host.cpp so that the SAME code is compiled twice: by hipcc into the library, and -- by the
// same clang++ and by g++ -- under -fsanitize=address,undefined and -fsanitize=thread into build/host_san/
// (alproj_amd/_build.py: build_host; tests/test_host_sanitized.py drives it).  Nothing here may include a HIP header.
#pragma once

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

#include "alproj_hip.h"

#if defined(__SANITIZE_THREAD__)
#define ALP_TSAN_BUILD 1
#elif defined(__has_feature)
#if __has_feature(thread_sanitizer)
#define ALP_TSAN_BUILD 1
#endif
#endif
#ifndef ALP_TSAN_BUILD
#define ALP_TSAN_BUILD 0
#endif

namespace alp {

// ------------------------------------------------------------------ errors (thread-local message; host/alp_host.cpp)
void set_error(const char *fmt, ...);
int fail(int code, const char *fmt, ...);

#define ALP_REQUIRE(cond, msg)                                                            \
    do {                                                                                  \
        if (!(cond)) return ::alp::fail(ALP_EINVAL, "%s: %s", __func__, msg);             \
    } while (0)

inline int64_t round_up(int64_t a, int64_t b) { return (a + b - 1) / b * b; }

// ------------------------------------------------------------------ pose record
// One camera pose folded, in float64 on the host, into the 32 numbers the kernels use.
// Layout (index):
//   0..3   row X' : x1 = (X'.[q;1]) / (Z.[q;1])   normalised, centred image x (see fold_pose)
//   4..7   row Y' : y1 = (Y'.[q;1]) / (Z.[q;1])
//   8..11  row Z  : depth along the optical axis
//   12..17 k1..k6
//   18,19  1+a1, 1+a2
//   20,21  2*p1, 2*p2
//   22..25 s1..s4
//   26,27  c0, c1 : float32-rounded image centre (w-1)/2, (h-1)/2
//   28,29  -c0, -c1
//   30,31  unused (0)
constexpr int POSE_WORDS = 32;

template <typename T>
struct alignas(16) PoseRec {
    T v[POSE_WORDS];
};

// params: the 25 ABI parameters; origin: local origin of the point set (absolute coords).
void fold_pose(const double params[ALP_NPARAM], const double origin[3], double rec[POSE_WORDS]);

// A LENS-FREE pose (k1..k6 = p1 = p2 = s1..s4 = 0: every candidate of the reference's first optimisation phase, example.py:51-54,
// whose targets are position, angles, fov, a1, a2) folded further: with those coefficients optimize.py:112-118 is
//   u = c0 x1 + c0,   v = c1 y1 (1 + a1) / (1 + a2) + c1
// so the residual needs no lens arithmetic at all:  uo - u = (uo - c0) + (X''.[q;1]) / (Z.[q;1])  with X'' = -c0 X',
// vo - v = (vo - c1) + (Y''.[q;1]) / (Z.[q;1])  with Y'' = -c1 (1 + a1) / (1 + a2) Y'  (scaled here in float64).
// rec[0..3] = X'', rec[4..7] = Y'', rec[8..11] = Z, rec[26], rec[27] = c0, c1 as in fold_pose, everything else 0.
bool pose_is_lens_free(const double params[ALP_NPARAM]);
void fold_pose_lens_free(const double params[ALP_NPARAM], const double origin[3], double rec[POSE_WORDS]);
// the same from the general record `g` of fold_pose (the sines and cosines are not formed twice)
void lens_free_from_general(const double g[POSE_WORDS], double rec[POSE_WORDS]);

template <typename T>
inline void fold_pose_t(const double params[ALP_NPARAM], const double origin[3], PoseRec<T> *out) {
    double r[POSE_WORDS];
    fold_pose(params, origin, r);
    for (int i = 0; i < POSE_WORDS; ++i) out->v[i] = (T)r[i];
}

namespace host {

// ------------------------------------------------------------------ threaded array helpers (alp_host_* entry points)
constexpr int64_t HASH_SLICE_BYTES = (int64_t)8 << 20;
constexpr int64_t MINMAX_VALUES_PER_THREAD = (int64_t)1 << 20;
// digest of `bytes` bytes: fixed 8 MB slices hashed by up to `threads` threads, slice digests chained in order -- the
// result does not depend on the thread count
uint64_t hash64(const void *buf, int64_t bytes, int threads);
// out = {min, max} of n >= 1 doubles, {NaN, NaN} if any is NaN (numpy's answer)
void minmax(const double *values, int64_t n, int threads, double out[2]);
// make the whole pages inside [buf, buf + bytes) exist (MADV_HUGEPAGE + MADV_POPULATE_WRITE; advice only)
void prefault(void *buf, int64_t bytes, int threads);

// ------------------------------------------------------------------ element-type conversion of a fetched plane
// (alp_projected_fetch with a change of type: a chunk arrives in pinned staging, host threads widen / narrow it into the
// caller's array while the next chunk crosses PCIe)
template <typename S, typename D>
inline void convert_slice(const S *__restrict__ src, D *__restrict__ dst, int64_t n) {
#if defined(__has_builtin) && __has_builtin(__builtin_nontemporal_store) && !ALP_TSAN_BUILD
    for (int64_t i = 0; i < n; ++i) __builtin_nontemporal_store((D)src[i], dst + i);     // the result is not read back here
#else   // g++ has no such builtin; and ThreadSanitizer does not instrument clang's (measured: two workers writing one range
        // went unreported), so its build keeps the workers' shares but stores plainly
    for (int64_t i = 0; i < n; ++i) dst[i] = (D)src[i];
#endif
}

constexpr int64_t CONVERT_SERIAL_BELOW = 1 << 16;       // a thread is not worth starting for fewer values

template <typename S, typename D>
inline void convert_threads(const S *src, D *dst, int64_t n, int T) {
    if (T <= 1 || n < CONVERT_SERIAL_BELOW) return convert_slice(src, dst, n);
    std::vector<std::thread> th;
    const int64_t per = ((n + T - 1) / T + 15) & ~(int64_t)15;
    for (int t = 1; t < T; ++t) {
        const int64_t a = std::min(n, per * t), b = std::min(n, per * (t + 1));
        if (b > a) th.emplace_back([=] { convert_slice(src + a, dst + a, b - a); });
    }
    convert_slice(src, dst, std::min(n, per));
    for (auto &x : th) x.join();
}

inline int fetch_threads() {
    if (const char *e = getenv("ALP_HOST_THREADS")) return std::max(1, std::min(64, atoi(e)));
    return (int)std::max(1u, std::min(16u, std::thread::hardware_concurrency()));
}

// ------------------------------------------------------------------ regular-grid recognition in a host index array
// Is a host index array exactly the regular grid of surface.py:194-201 with gw columns?  Answered by a few host threads
// WHILE the vertices and colours cross PCIe: a full-grid array then never crosses it at all (4.8 GB of int64 at 100 M
// vertices = 86 ms of PCIe time, a third of the reference-typed first call).  Each thread walks whole grid rows of its
// share of the cells (a streaming compare against a + {0, gw, gw+1, 0, gw+1, 1}); the first mismatch stops everybody.
template <typename I>
inline void grid_rows_check(const I *ind, long long gw, long long row0, long long row1, std::atomic<bool> *bad) {
    const long long gc = gw - 1;
    for (long long r = row0; r < row1 && !bad->load(std::memory_order_relaxed); ++r) {
        const I *p = ind + (size_t)r * gc * 6;
        long long a = r * gw;
        bool diff = false;
        for (long long c = 0; c < gc; ++c, ++a, p += 6)
            diff |= (long long)p[0] != a || (long long)p[1] != a + gw || (long long)p[2] != a + gw + 1 || (long long)p[3] != a ||
                    (long long)p[4] != a + gw + 1 || (long long)p[5] != a + 1;
        if (diff) bad->store(true, std::memory_order_relaxed);
    }
}

struct HostGridCheck {
    std::vector<std::thread> threads;
    std::atomic<bool> bad{false};
    bool started = false;
    void start(const void *ind, int ind_dtype, long long gh, long long gw, int n_threads) {
        const long long rows = gh - 1;
        const int T = (int)std::min<long long>(n_threads, rows);
        try {
            for (int t = 0; t < T; ++t) {
                const long long r0 = rows * t / T, r1 = rows * (t + 1) / T;
                if (ind_dtype == ALP_I32) threads.emplace_back(grid_rows_check<int>, (const int *)ind, gw, r0, r1, &bad);
                else threads.emplace_back(grid_rows_check<long long>, (const long long *)ind, gw, r0, r1, &bad);
            }
            started = true;
        } catch (...) {               // no threads to be had: the caller falls back to the check on the device
            bad.store(true);
            join();
            bad.store(false);
            started = false;
        }
    }
    void join() {
        for (auto &t : threads)
            if (t.joinable()) t.join();
        threads.clear();
    }
    bool is_grid() {                   // joins
        join();
        return started && !bad.load();
    }
    ~HostGridCheck() { bad.store(true); join(); }
};

// host threads for HostGridCheck: ALP_HOST_THREADS (0 = check on the device instead), else up to 8 of the machine's
inline int host_check_threads(int64_t n_tri) {
    if (const char *e = getenv("ALP_HOST_THREADS")) return std::max(0, std::min(64, atoi(e)));     // tests: either path at any size
    if (n_tri < (1 << 18)) return 0;           // small arrays: the staged check costs nothing
    const unsigned hc = std::thread::hardware_concurrency();
    return hc >= 4 ? (int)std::min(8u, hc / 2) : 0;
}

// Does an index array start like the regular grid of n_vert vertices and have exactly its triangle count?  Then
// (*gh, *gw) is the candidate HostGridCheck (or the device) confirms.  ind must hold at least 3 entries when n_tri >= 1.
inline bool grid_candidate(const void *ind, int ind_dtype, int64_t n_tri, int64_t n_vert, long long *gh_out, long long *gw_out) {
    if (n_tri < 2 || (n_tri & 1)) return false;
    long long first[3];
    for (int k = 0; k < 3; ++k)
        first[k] = ind_dtype == ALP_I32 ? (long long)((const int *)ind)[k] : ((const long long *)ind)[k];
    const long long gw = first[1] - first[0];
    if (first[0] != 0 || gw < 2 || first[2] != gw + 1 || n_vert % gw != 0) return false;
    const long long gh = n_vert / gw;
    if (gh < 2 || n_tri != 2 * (gh - 1) * (gw - 1)) return false;
    *gh_out = gh;
    *gw_out = gw;
    return true;
}

// ------------------------------------------------------------------ argmin of a population and its confirmation band
// float32 point sets: when the smallest loss and its runner-up differ by less than CONFIRM_GAP (relative) the candidates
// inside that band are evaluated again in float64 arithmetic before the argmin is returned (north star: "argmin pose
// index bit-exact"; float32 losses carry ~1e-6..1e-5)
constexpr int CONFIRM_MAX = 16;
constexpr double CONFIRM_GAP = 5e-5;

// loss[i] = sums[i] / n_total (np.mean over all vertices); returns the first index of the smallest non-NaN loss
// (-1 if every loss is NaN) and its value
inline int64_t losses_and_argmin(const double *sums, int64_t P, double n_total, double *loss, double *best_v_out) {
    int64_t best = -1;
    double best_v = 0;
    for (int64_t i = 0; i < P; ++i) {
        const double l = sums[i] / n_total;
        loss[i] = l;
        if (l == l && (best < 0 || l < best_v)) {       // NaN never wins; first index on ties
            best = i;
            best_v = l;
        }
    }
    *best_v_out = best_v;
    return best;
}

// candidates whose loss lies within CONFIRM_GAP of the smallest one: returns how many there are; which[0..*K) holds
// (up to) the CONFIRM_MAX smallest of them in ascending index order
inline int64_t confirm_band(const double *loss, int64_t P, double best_v, int64_t which[CONFIRM_MAX], int *K_out) {
    const double band = best_v + CONFIRM_GAP * std::fabs(best_v);
    int K = 0;
    int64_t in_band = 0;
    for (int64_t i = 0; i < P; ++i)
        if (loss[i] <= band) {
            ++in_band;
            if (K < CONFIRM_MAX) {
                which[K++] = i;
            } else {           // keep the CONFIRM_MAX smallest: replace the largest kept one if this is smaller
                int worst = 0;
                for (int k = 1; k < CONFIRM_MAX; ++k)
                    if (loss[which[k]] > loss[which[worst]] || (loss[which[k]] == loss[which[worst]] && which[k] > which[worst])) worst = k;
                if (loss[i] < loss[which[worst]]) which[worst] = i;
            }
        }
    std::sort(which, which + K);
    *K_out = K;
    return in_band;
}

// the float64 sums of the K confirmed candidates (sums[K] = vertex count) replace their losses; returns the argmin
// among them (first index on ties; which[0] if every one is NaN)
inline int64_t merge_confirmed(double *loss, const int64_t *which, int K, const double *sums) {
    int64_t best = -1;
    double best_v = 0;
    for (int k = 0; k < K; ++k) {
        const double l = sums[k] / sums[K];
        loss[which[k]] = l;
        if (l == l && (best < 0 || l < best_v)) {
            best = which[k];
            best_v = l;
        }
    }
    return best < 0 ? which[0] : best;
}

}  // namespace host
}  // namespace alp
