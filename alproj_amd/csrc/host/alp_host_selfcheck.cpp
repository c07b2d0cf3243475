// Self-checking driver of the HIP-free host code (host/alp_host.h, host/alp_host.cpp).  NOT part of libalproj_hip.so:
// alproj_amd/_build.py: build_host() links it with alp_host.cpp into build/host_san/alp_host_{plain,asan,tsan} and
// tests/test_host_sanitized.py runs the three executables.  Every threaded helper is called with sizes that straddle
// its thread thresholds (1, 70 001, one short of / exactly / one past a slice, two slices and a ragged rest), with
// thread counts from 1 to 64, from several caller threads at once (independent handles may be used concurrently:
// include/alproj_hip.h), and compared with a serial restatement written here.  Exit code 0 and "host selfcheck ok" =
// every comparison held; the sanitizers add their own verdict on stderr.
#include <cinttypes>
#include <mutex>
#include <random>
#include <string>

#include "host/alp_host.h"

using namespace alp;

namespace {

std::atomic<int> g_failures{0};
std::mutex g_print;

#define CHECK(cond, ...)                                           \
    do {                                                           \
        if (!(cond)) {                                             \
            std::lock_guard<std::mutex> lk(g_print);               \
            fprintf(stderr, "CHECK FAILED %s:%d: %s | ", __FILE__, __LINE__, #cond); \
            fprintf(stderr, __VA_ARGS__);                          \
            fprintf(stderr, "\n");                                 \
            g_failures.fetch_add(1);                               \
        }                                                          \
    } while (0)

const int THREAD_COUNTS[] = {1, 2, 3, 7, 16, 64};

// ------------------------------------------------------------------ alp_host_hash64
void check_hash() {
    const int64_t SL = host::HASH_SLICE_BYTES;
    const int64_t sizes[] = {0, 1, 7, 8, 31, 32, 33, 70001, SL - 1, SL, SL + 1, 2 * SL + 12345};
    std::vector<unsigned char> buf((size_t)(2 * SL + 12345) + 3);
    std::mt19937_64 rng(1);
    for (auto &b : buf) b = (unsigned char)rng();
    for (int shift = 0; shift < 2; ++shift)                    // an aligned and an odd start address
        for (int64_t n : sizes) {
            uint64_t d1 = 0;
            CHECK(alp_host_hash64(buf.data() + shift, n, 1, &d1) == ALP_OK, "n=%" PRId64, n);
            for (int T : THREAD_COUNTS) {
                uint64_t d = 0;
                CHECK(alp_host_hash64(buf.data() + shift, n, T, &d) == ALP_OK, "n=%" PRId64, n);
                CHECK(d == d1, "digest depends on the thread count: n=%" PRId64 " T=%d", n, T);
            }
            uint64_t d0 = 0;
            CHECK(alp_host_hash64(buf.data() + shift, n, 0, &d0) == ALP_OK && d0 == d1, "default thread count, n=%" PRId64, n);
            if (n > 0) {                                        // one changed byte (first, last) always changes the digest
                for (int64_t at : {(int64_t)0, n - 1}) {
                    buf[(size_t)(shift + at)] ^= 0x40;
                    uint64_t d = 0;
                    alp_host_hash64(buf.data() + shift, n, 3, &d);
                    CHECK(d != d1, "changed byte %" PRId64 " of %" PRId64 " not seen", at, n);
                    buf[(size_t)(shift + at)] ^= 0x40;
                }
            }
        }
    uint64_t d = 0;
    CHECK(alp_host_hash64(nullptr, 8, 1, &d) == ALP_EINVAL, "NULL buffer accepted");
    CHECK(alp_host_hash64(buf.data(), -1, 1, &d) == ALP_EINVAL, "negative size accepted");
    CHECK(alp_host_hash64(buf.data(), 8, 1, nullptr) == ALP_EINVAL, "NULL digest accepted");
    CHECK(strstr(alp_last_error(), "alp_host_hash64") != nullptr, "error text: %s", alp_last_error());
}

// ------------------------------------------------------------------ alp_host_minmax
void check_minmax() {
    const int64_t per = host::MINMAX_VALUES_PER_THREAD;
    const int64_t sizes[] = {1, 7, 8, 9, 70001, per - 1, per, per + 1, 2 * per + 77777};
    std::vector<double> v((size_t)(2 * per + 77777));
    std::mt19937_64 rng(2);
    std::uniform_real_distribution<double> U(-1e6, 1e6);
    for (auto &x : v) x = U(rng);
    for (int64_t n : sizes) {
        double lo = INFINITY, hi = -INFINITY;
        for (int64_t i = 0; i < n; ++i) { lo = std::min(lo, v[(size_t)i]); hi = std::max(hi, v[(size_t)i]); }
        for (int T : {0, 1, 2, 3, 8, 64}) {
            double out[2] = {0, 0};
            CHECK(alp_host_minmax(v.data(), n, T, out) == ALP_OK, "n=%" PRId64, n);
            CHECK(out[0] == lo && out[1] == hi, "n=%" PRId64 " T=%d: %g %g vs %g %g", n, T, out[0], out[1], lo, hi);
        }
        // a NaN anywhere (first value, last value, the first value of the second thread's share) poisons both
        for (int64_t at : {(int64_t)0, n - 1, n / 2}) {
            const double keep = v[(size_t)at];
            v[(size_t)at] = NAN;
            for (int T : {1, 2, 8}) {
                double out[2] = {0, 0};
                alp_host_minmax(v.data(), n, T, out);
                CHECK(out[0] != out[0] && out[1] != out[1], "NaN at %" PRId64 " of %" PRId64 " lost (T=%d)", at, n, T);
            }
            v[(size_t)at] = keep;
        }
    }
    double out[2];
    CHECK(alp_host_minmax(v.data(), 0, 1, out) == ALP_EINVAL, "n = 0 accepted");
    CHECK(alp_host_minmax(nullptr, 4, 1, out) == ALP_EINVAL, "NULL accepted");
}

// ------------------------------------------------------------------ alp_host_prefault
void check_prefault() {
    const int64_t sizes[] = {0, 1, 4095, 4096, 4097, 70001, (5 << 20) + 3, 9 << 20};
    for (int64_t n : sizes)
        for (int shift : {0, 1, 4095})
            for (int T : {0, 1, 4, 64}) {
                std::vector<unsigned char> buf((size_t)(n + shift) + 1, 0xA5);
                CHECK(alp_host_prefault(buf.data() + shift, n, T) == ALP_OK, "n=%" PRId64, n);
                // populated pages keep their contents (MADV_POPULATE_WRITE faults them in, it does not clear them)
                bool same = true;
                for (unsigned char b : buf) same &= b == 0xA5;
                CHECK(same, "prefault changed the buffer: n=%" PRId64 " shift=%d T=%d", n, shift, T);
            }
    CHECK(alp_host_prefault(nullptr, 16, 1) == ALP_EINVAL, "NULL accepted");
    CHECK(alp_host_prefault(nullptr, 0, 1) == ALP_OK, "empty range refused");
}

// ------------------------------------------------------------------ fold_pose against the unfolded arithmetic
// optimize.py:35-38 (intrinsic_mat), :71-95 (extrinsic_mat), :144-149 (project), :104-106 (_distort's centring), written
// out matrix by matrix as the reference has it; the folded record must give the same normalised coordinates.
void check_fold_pose() {
    std::mt19937_64 rng(3);
    std::uniform_real_distribution<double> U(-1, 1);
    for (int trial = 0; trial < 200; ++trial) {
        double p[ALP_NPARAM] = {0};
        const double origin[3] = {732000 + 100 * U(rng), 4048000 + 100 * U(rng), 2000 + 10 * U(rng)};
        p[0] = origin[0] + 500 * U(rng); p[1] = origin[1] + 500 * U(rng); p[2] = origin[2] + 200 * U(rng);
        p[3] = 60 + 25 * U(rng); p[4] = 180 * U(rng); p[5] = 30 * U(rng); p[6] = 10 * U(rng);
        for (int i = 7; i < 21; ++i) p[i] = 0.1 * U(rng);
        p[21] = 5616; p[22] = 3744; p[23] = 2808 + 50 * U(rng); p[24] = 1872 + 50 * U(rng);
        double rec[POSE_WORDS];
        fold_pose(p, origin, rec);
        PoseRec<float> rf;
        fold_pose_t<float>(p, origin, &rf);
        for (int i = 0; i < POSE_WORDS; ++i) CHECK(rf.v[i] == (float)rec[i], "fold_pose_t word %d", i);
        CHECK(rec[18] == 1 + p[7] && rec[19] == 1 + p[8] && rec[20] == 2 * p[15] && rec[21] == 2 * p[16], "coefficients");
        const double pi = M_PI, w = p[21], h = p[22];
        const double fovx = p[3] * pi / 180, fovy = fovx * h / w;
        const double fx = w / (2 * std::tan(fovx / 2)), fy = h / (2 * std::tan(fovy / 2));
        const double a = p[4] * pi / 180, b = -(p[5] + 90) * pi / 180, c = -p[6] * pi / 180;
        const double Rz[3][3] = {{cos(a), -sin(a), 0}, {sin(a), cos(a), 0}, {0, 0, 1}};
        const double Rx[3][3] = {{1, 0, 0}, {0, cos(b), -sin(b)}, {0, sin(b), cos(b)}};
        const double Ry[3][3] = {{cos(c), 0, sin(c)}, {0, 1, 0}, {-sin(c), 0, cos(c)}};
        double M[3][3] = {{0}}, R[3][3] = {{0}};
        for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) for (int k = 0; k < 3; ++k) M[i][j] += Rx[i][k] * Ry[k][j];
        for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) for (int k = 0; k < 3; ++k) R[i][j] += M[i][k] * Rz[k][j];
        const double c0 = (double)(float)((w - 1) / 2), c1 = (double)(float)((h - 1) / 2);
        for (int k = 0; k < 8; ++k) {
            const double q[3] = {2000 * U(rng), 2000 * U(rng), 300 * U(rng)};       // local coordinates
            const double P[3] = {origin[0] + q[0] - p[0], origin[1] + q[1] - p[1], origin[2] + q[2] - p[2]};
            double cam[3];
            for (int i = 0; i < 3; ++i) cam[i] = R[i][0] * P[0] + R[i][1] * P[1] + R[i][2] * P[2];
            const double x = fx * cam[0] + p[23] * cam[2], y = fy * cam[1] + p[24] * cam[2], z = cam[2];
            const double x1 = ((w - x / z) - c0) / c0, y1 = (y / z - c1) / c1;
            const double zf = rec[8] * q[0] + rec[9] * q[1] + rec[10] * q[2] + rec[11];
            const double x1f = (rec[0] * q[0] + rec[1] * q[1] + rec[2] * q[2] + rec[3]) / zf;
            const double y1f = (rec[4] * q[0] + rec[5] * q[1] + rec[6] * q[2] + rec[7]) / zf;
            if (std::fabs(z) < 1) continue;                    // next to the camera plane the quotient amplifies rounding
            CHECK(std::fabs(x1f - x1) <= 1e-9 * std::max(1.0, std::fabs(x1)), "x1 %.17g vs %.17g", x1f, x1);
            CHECK(std::fabs(y1f - y1) <= 1e-9 * std::max(1.0, std::fabs(y1)), "y1 %.17g vs %.17g", y1f, y1);
            // the lens-free folding of the same pose with its lens coefficients set to zero: optimize.py:112-118 is then
            // u = c0 x1 + c0, v = c1 y1 (1 + a1) / (1 + a2) + c1, and the folded rows give the residual against any (uo, vo)
            double pl[ALP_NPARAM];
            memcpy(pl, p, sizeof(pl));
            for (int i = 9; i <= 20; ++i) pl[i] = 0;
            CHECK(pose_is_lens_free(pl) && !pose_is_lens_free(p), "pose_is_lens_free");
            double lf[POSE_WORDS];
            fold_pose_lens_free(pl, origin, lf);
            const double uo = 5616 * U(rng), vo = 3744 * U(rng);
            const double u_ref = c0 * x1 + c0, v_ref = c1 * (y1 * (1 + pl[7]) / (1 + pl[8])) + c1;
            const double zl = lf[8] * q[0] + lf[9] * q[1] + lf[10] * q[2] + lf[11];
            const double du = (uo - lf[26]) + (lf[0] * q[0] + lf[1] * q[1] + lf[2] * q[2] + lf[3]) / zl;
            const double dv = (vo - lf[27]) + (lf[4] * q[0] + lf[5] * q[1] + lf[6] * q[2] + lf[7]) / zl;
            CHECK(std::fabs(du - (uo - u_ref)) <= 1e-8 * std::max(1.0, std::fabs(u_ref)), "lens-free du %.17g vs %.17g", du, uo - u_ref);
            CHECK(std::fabs(dv - (vo - v_ref)) <= 1e-8 * std::max(1.0, std::fabs(v_ref)), "lens-free dv %.17g vs %.17g", dv, vo - v_ref);
            CHECK(zl == zf && lf[26] == c0 && lf[27] == c1 && lf[12] == 0 && lf[31] == 0, "lens-free record layout");
        }
    }
}

// ------------------------------------------------------------------ conversion workers of the converting fetch
template <typename S, typename D>
void check_convert_t() {
    const int64_t lim = host::CONVERT_SERIAL_BELOW;
    const int64_t sizes[] = {0, 1, 15, 16, 17, lim - 1, lim, lim + 1, 70001, (1 << 20) + 3};
    std::mt19937_64 rng(4);
    std::uniform_real_distribution<double> U(-1e7, 1e7);
    for (int64_t n : sizes) {
        std::vector<S> src((size_t)n);                          // exactly n: a sanitizer sees one value too many on either side
        for (auto &x : src) x = (S)U(rng);
        for (int T : THREAD_COUNTS) {
            std::vector<D> dst((size_t)n, (D)-1);
            host::convert_threads(src.data(), dst.data(), n, T);
            bool same = true;
            for (int64_t i = 0; i < n; ++i) same &= dst[(size_t)i] == (D)src[(size_t)i];
            CHECK(same, "convert n=%" PRId64 " T=%d", n, T);
        }
    }
}

void check_convert() {
    check_convert_t<float, double>();
    check_convert_t<double, float>();
}

// ------------------------------------------------------------------ regular-grid recognition
template <typename I>
std::vector<I> grid_indices(long long gh, long long gw) {       // surface.py:194-201
    std::vector<I> ind;
    ind.reserve((size_t)((gh - 1) * (gw - 1) * 6));
    for (long long r = 0; r + 1 < gh; ++r)
        for (long long c = 0; c + 1 < gw; ++c) {
            const long long a = r * gw + c;
            for (long long v : {a, a + gw, a + gw + 1, a, a + gw + 1, a + 1}) ind.push_back((I)v);
        }
    return ind;
}

template <typename I>
void check_grid_t(int dtype) {
    const long long shapes[][2] = {{2, 2}, {3, 5}, {17, 4}, {2, 300}, {300, 251}};
    for (auto &sh : shapes) {
        const long long gh = sh[0], gw = sh[1];
        std::vector<I> ind = grid_indices<I>(gh, gw);
        const int64_t n_tri = (int64_t)ind.size() / 3, n_vert = gh * gw;
        long long cgh = 0, cgw = 0;
        CHECK(host::grid_candidate(ind.data(), dtype, n_tri, n_vert, &cgh, &cgw) && cgh == gh && cgw == gw, "candidate %lld x %lld", gh, gw);
        CHECK(!host::grid_candidate(ind.data(), dtype, n_tri - 1, n_vert, &cgh, &cgw), "odd triangle count accepted");
        CHECK(!host::grid_candidate(ind.data(), dtype, n_tri, n_vert + 1, &cgh, &cgw) || (n_vert + 1) % gw == 0, "wrong vertex count accepted");
        for (int T : {1, 2, 8, 64}) {
            {
                host::HostGridCheck chk;
                chk.start(ind.data(), dtype, gh, gw, T);
                CHECK(chk.is_grid(), "grid %lld x %lld not recognised (T=%d)", gh, gw, T);
            }
            for (size_t at : {(size_t)0, ind.size() / 2, ind.size() - 1}) {     // one wrong entry anywhere
                const I keep = ind[at];
                ind[at] = (I)(keep + 1);
                {
                    host::HostGridCheck chk;
                    chk.start(ind.data(), dtype, gh, gw, T);
                    CHECK(!chk.is_grid(), "wrong entry %zu of a %lld x %lld grid not seen (T=%d)", at, gh, gw, T);
                }
                ind[at] = keep;
            }
            {                                                   // leaving the scope without asking: the destructor stops and joins
                host::HostGridCheck chk;
                chk.start(ind.data(), dtype, gh, gw, T);
            }
            {                                                   // never started
                host::HostGridCheck chk;
                CHECK(!chk.is_grid(), "a check that never ran says grid");
            }
        }
    }
}

void check_grid() {
    check_grid_t<int>(ALP_I32);
    check_grid_t<long long>(ALP_I64);
}

// ------------------------------------------------------------------ argmin and confirmation band
void check_selection() {
    std::mt19937_64 rng(5);
    std::uniform_real_distribution<double> U(0, 1);
    for (int trial = 0; trial < 400; ++trial) {
        const int64_t P = 1 + (int64_t)(rng() % 300);
        const double n_total = 1000;
        std::vector<double> sums((size_t)P), loss((size_t)P);
        const int mode = trial % 5;
        for (int64_t i = 0; i < P; ++i) {
            double l = 10 + U(rng);                                              // mode 0: well separated
            if (mode == 1) l = 10 * (1 + 1e-5 * U(rng));                         // everything inside the band
            if (mode == 2 && rng() % 3 == 0) l = NAN;                            // NaNs in between
            if (mode == 3) l = (double)(1 + rng() % 3);                          // many exact ties
            if (mode == 4) l = NAN;                                              // all NaN
            sums[(size_t)i] = l * n_total;
        }
        double best_v = 0;
        const int64_t best = host::losses_and_argmin(sums.data(), P, n_total, loss.data(), &best_v);
        int64_t want = -1;
        for (int64_t i = 0; i < P; ++i)
            if (loss[(size_t)i] == loss[(size_t)i] && (want < 0 || loss[(size_t)i] < loss[(size_t)want])) want = i;
        CHECK(best == want, "argmin %" PRId64 " vs %" PRId64, best, want);
        if (best < 0) continue;
        CHECK(best_v == loss[(size_t)best], "best value");
        int64_t which[host::CONFIRM_MAX];
        int K = 0;
        const int64_t in_band = host::confirm_band(loss.data(), P, best_v, which, &K);
        std::vector<int64_t> band;
        for (int64_t i = 0; i < P; ++i)
            if (loss[(size_t)i] <= best_v + host::CONFIRM_GAP * std::fabs(best_v)) band.push_back(i);
        CHECK(in_band == (int64_t)band.size(), "band size");
        CHECK(K == (int)std::min<size_t>(band.size(), host::CONFIRM_MAX), "K");
        for (int k = 1; k < K; ++k) CHECK(which[k - 1] < which[k], "which[] not ascending");
        // the kept ones are the K smallest of the band: nothing left out is smaller than something kept
        double kept_max = -INFINITY;
        for (int k = 0; k < K; ++k) kept_max = std::max(kept_max, loss[(size_t)which[k]]);
        for (int64_t i : band)
            if (!std::binary_search(which, which + K, i)) CHECK(loss[(size_t)i] >= kept_max, "a smaller candidate was left out of the band");
        CHECK(std::binary_search(which, which + K, best), "the float32 argmin is not in its own band");
        // confirmation: float64 sums reorder the band
        std::vector<double> csums((size_t)K + 1);
        for (int k = 0; k < K; ++k) csums[(size_t)k] = (mode == 2 && k % 2 ? NAN : U(rng)) * n_total;
        csums[(size_t)K] = n_total;
        const int64_t cbest = host::merge_confirmed(loss.data(), which, K, csums.data());
        int64_t cwant = -1;
        double cwant_v = 0;
        for (int k = 0; k < K; ++k) {
            const double l = csums[(size_t)k] / n_total, stored = loss[(size_t)which[k]];
            CHECK(stored == l || (l != l && stored != stored), "confirmed loss not stored");
            if (l == l && (cwant < 0 || l < cwant_v)) { cwant = which[k]; cwant_v = l; }
        }
        CHECK(cbest == (cwant < 0 ? which[0] : cwant), "confirmed argmin");
    }
}

// ------------------------------------------------------------------ the error message is per thread
void check_errors() {
    std::vector<std::thread> th;
    for (int t = 0; t < 8; ++t)
        th.emplace_back([t] {
            for (int k = 0; k < 2000; ++k) {
                const int code = fail(ALP_EINVAL, "thread %d call %d", t, k);
                char want[64];
                snprintf(want, sizeof(want), "thread %d call %d", t, k);
                CHECK(code == ALP_EINVAL && !strcmp(alp_last_error(), want), "message of another thread: %s", alp_last_error());
            }
        });
    for (auto &x : th) x.join();
}

}  // namespace

// Deliberate defects, one per sanitizer: tests/test_host_sanitized.py expects the matching build to REPORT them -- a
// sanitizer that says nothing about these says nothing by staying silent on the real code either.
int canary(const char *which) {
    if (!strcmp(which, "overflow")) {                           // AddressSanitizer: one float past a conversion's destination
        std::vector<float> src(70001, 1.0f);
        std::vector<double> dst(70000);
        host::convert_threads(src.data(), dst.data(), 70001, 4);
        return dst[69999] == 1.0 ? 0 : 1;
    }
    if (!strcmp(which, "race")) {                               // ThreadSanitizer: two workers given overlapping shares
        std::vector<float> src(1 << 17, 1.0f);
        std::vector<double> dst(1 << 17);
        std::atomic<int> gate{0};                               // both workers write at the same time, as the real ones do
        auto arrive = [&] { gate.fetch_add(1); while (gate.load() < 2) {} };
        std::thread a([&] { arrive(); host::convert_slice(src.data(), dst.data(), (int64_t)1 << 17); });
        std::thread b([&] { arrive(); host::convert_slice(src.data(), dst.data() + 16, ((int64_t)1 << 17) - 16); });
        a.join();
        b.join();
        return 0;
    }
    if (!strcmp(which, "shift")) {                              // UndefinedBehaviorSanitizer: a 64-bit rotate by 64
        volatile int r = 64;
        volatile uint64_t x = 1;
        return (int)((x << r) & 1);
    }
    return 2;
}

int main(int argc, char **argv) {
    if (argc > 2 && !strcmp(argv[1], "--canary")) return canary(argv[2]);
    struct Group { const char *name; void (*fn)(); };
    const Group groups[] = {{"hash", check_hash},     {"minmax", check_minmax},   {"prefault", check_prefault}, {"fold_pose", check_fold_pose},
                            {"convert", check_convert}, {"grid", check_grid},     {"selection", check_selection}, {"errors", check_errors}};
    const bool concurrent = !(argc > 1 && !strcmp(argv[1], "--serial"));
    // every group on its own caller thread at once: the library promises that independent calls may overlap
    std::vector<std::thread> th;
    for (const Group &g : groups) {
        if (concurrent) th.emplace_back(g.fn);
        else g.fn();
    }
    for (auto &x : th) x.join();
    // ... and the same helper from two caller threads at once
    std::thread a(check_convert), b(check_convert);
    check_selection();
    a.join();
    b.join();
    if (g_failures.load()) {
        fprintf(stderr, "host selfcheck: %d comparison(s) failed\n", g_failures.load());
        return 1;
    }
    printf("host selfcheck ok: %zu groups%s\n", sizeof(groups) / sizeof(groups[0]), concurrent ? ", run concurrently" : "");
    return 0;
}
