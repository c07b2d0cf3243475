// libalproj_hip.so -- library context: device selection, stream, event timers, RCCL
// communicator, error reporting, and the float64 host-side folding of one camera pose.
#include "alp_internal.h"

#include <rccl/rccl.h>
#include <sys/mman.h>

#include <cmath>
#include <cstdlib>
#include <thread>
#include <vector>

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

Context &ctx() {
    static Context c;
    return c;
}

int require_init() {
    if (!ctx().ready) return fail(ALP_ENOTINIT, "alp_init has not been called (or failed)");
    return ALP_OK;
}

int scratch_reserve(size_t bytes, void **out) {
    Context &c = ctx();
    if (bytes > c.scratch_cap) {
        if (c.scratch) {
            ALP_HIP(hipStreamSynchronize(c.stream));
            hipFree(c.scratch);
        }
        c.scratch = nullptr;
        c.scratch_cap = 0;
        const size_t cap = (bytes + ((size_t)1 << 20) - 1) & ~(((size_t)1 << 20) - 1);
        ALP_HIP(hipMalloc(&c.scratch, cap));
        c.scratch_cap = cap;
    }
    *out = c.scratch;
    return ALP_OK;
}

// ------------------------------------------------------------------ kernel-section timer
namespace {
constexpr int KT_MAX = 256;
struct KTimer {
    bool on = false;
    int n = 0;               // sections recorded since the last reset
    bool open = false;
    hipEvent_t a[KT_MAX] = {}, b[KT_MAX] = {};
} g_kt;
}  // namespace

void ktime_begin() {
    if (!g_kt.on || g_kt.n >= KT_MAX || g_kt.open) return;
    if (!g_kt.a[g_kt.n] && (hipEventCreate(&g_kt.a[g_kt.n]) != hipSuccess || hipEventCreate(&g_kt.b[g_kt.n]) != hipSuccess)) return;
    g_kt.open = hipEventRecord(g_kt.a[g_kt.n], ctx().stream) == hipSuccess;
}

void ktime_end() {
    if (!g_kt.open) return;
    g_kt.open = false;
    if (hipEventRecord(g_kt.b[g_kt.n], ctx().stream) == hipSuccess) ++g_kt.n;
}

#define ALP_NCCL(expr)                                                                    \
    do {                                                                                  \
        ncclResult_t r__ = (expr);                                                        \
        if (r__ != ncclSuccess)                                                           \
            return ::alp::fail(ALP_ERCCL, "%s failed: %s | RCCL says: %s (%s:%d)", #expr, \
                               ncclGetErrorString(r__), ncclGetLastError(nullptr),        \
                               __FILE__, __LINE__);                                       \
    } while (0)

int comm_allreduce_sum_f64(double *dev_buf, int64_t count) {
    Context &c = ctx();
    if (!c.comm) return ALP_OK;
    ALP_NCCL(ncclAllReduce(dev_buf, dev_buf, (size_t)count, ncclDouble, ncclSum,
                           (ncclComm_t)c.comm, c.stream));
    return ALP_OK;
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

}  // namespace alp

using namespace alp;

extern "C" {

int alp_abi_version(void) { return ALP_ABI_VERSION; }

const char *alp_last_error(void) { return g_err; }

int alp_device_count(int *count) {
    ALP_REQUIRE(count, "count is NULL");
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) {
        *count = 0;
        return fail(ALP_ENODEVICE, "hipGetDeviceCount: %s", hipGetErrorString(e));
    }
    *count = n;
    return ALP_OK;
}

int alp_init(int device) {
    Context &c = ctx();
    if (c.ready && c.device == device) return ALP_OK;
    if (c.ready) return fail(ALP_EINVAL, "already initialised on device %d", c.device);
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0)
        return fail(ALP_ENODEVICE, "no HIP device available (%s); libalproj_hip has no CPU fallback",
                    e == hipSuccess ? "device count is 0" : hipGetErrorString(e));
    if (device < 0 || device >= n)
        return fail(ALP_ENODEVICE, "device %d out of range (have %d)", device, n);
    ALP_HIP(hipSetDevice(device));
    hipDeviceProp_t prop;
    ALP_HIP(hipGetDeviceProperties(&prop, device));
    c.cu_count = prop.multiProcessorCount;
    ALP_HIP(hipStreamCreateWithFlags(&c.stream, hipStreamNonBlocking));
    for (auto &ev : c.events) ALP_HIP(hipEventCreate(&ev));
    c.device = device;
    c.ready = true;
    return ALP_OK;
}

int alp_shutdown(void) {
    Context &c = ctx();
    if (!c.ready) return ALP_OK;
    alp_comm_destroy();
    hipStreamSynchronize(c.stream);
    points_release_staging();
    if (c.scratch) hipFree(c.scratch);
    c.scratch = nullptr;
    c.scratch_cap = 0;
    for (auto &ev : c.events) {
        if (ev) hipEventDestroy(ev);
        ev = nullptr;
    }
    for (int i = 0; i < KT_MAX; ++i) {
        if (g_kt.a[i]) hipEventDestroy(g_kt.a[i]);
        if (g_kt.b[i]) hipEventDestroy(g_kt.b[i]);
        g_kt.a[i] = g_kt.b[i] = nullptr;
    }
    g_kt = KTimer();
    hipStreamDestroy(c.stream);
    c.stream = nullptr;
    c.ready = false;
    c.device = -1;
    return ALP_OK;
}

int alp_device_info(char *name, int len, int *cu_count, int64_t *hbm_bytes) {
    if (int rc = require_init()) return rc;
    hipDeviceProp_t prop;
    ALP_HIP(hipGetDeviceProperties(&prop, ctx().device));
    if (name && len > 0) {
        strncpy(name, prop.gcnArchName, (size_t)len - 1);
        name[len - 1] = 0;
    }
    if (cu_count) *cu_count = prop.multiProcessorCount;
    if (hbm_bytes) *hbm_bytes = (int64_t)prop.totalGlobalMem;
    return ALP_OK;
}

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

int alp_host_hash64(const void *buf, int64_t bytes, int threads, uint64_t *digest) {
    ALP_REQUIRE(digest && bytes >= 0 && (bytes == 0 || buf), "bad argument");
    const unsigned char *p = (const unsigned char *)buf;
    int T = threads > 0 ? threads : (int)std::thread::hardware_concurrency();
    if (T < 1) T = 1;
    if (T > 64) T = 64;
    const int64_t SL = (int64_t)8 << 20;                        // fixed slices: the digest does not depend on the thread count
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
    *digest = avalanche(h);
    return ALP_OK;
}

int alp_host_minmax(const double *values, int64_t n, int threads, double out[2]) {
    ALP_REQUIRE(values && out && n >= 1, "bad argument");
    int T = threads > 0 ? threads : (int)std::min(8u, std::max(1u, std::thread::hardware_concurrency()));
    if (T > 64) T = 64;
    const int64_t per = (int64_t)1 << 20;                        // a thread is worth starting for a million values
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
    return ALP_OK;
}

#ifndef MADV_POPULATE_WRITE
#define MADV_POPULATE_WRITE 23          // Linux 5.14
#endif
int alp_host_prefault(void *buf, int64_t bytes, int threads) {
    ALP_REQUIRE(bytes >= 0 && (bytes == 0 || buf), "bad argument");
    const uintptr_t PG = 4096, HP = (uintptr_t)2 << 20;
    const uintptr_t a = ((uintptr_t)buf + PG - 1) & ~(PG - 1), b = ((uintptr_t)buf + (uintptr_t)bytes) & ~(PG - 1);
    if (b <= a) return ALP_OK;
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
    return ALP_OK;
}

int alp_device_pci_bus_id(char *id, int len) {
    if (int rc = require_init()) return rc;
    ALP_REQUIRE(id && len >= 16, "id is NULL or shorter than 16 bytes");
    ALP_HIP(hipDeviceGetPCIBusId(id, len, ctx().device));
    return ALP_OK;
}

int alp_kernel_timing(int enable) {
    if (int rc = require_init()) return rc;
    ALP_HIP(hipStreamSynchronize(ctx().stream));
    g_kt.on = enable != 0;
    g_kt.n = 0;
    g_kt.open = false;
    return ALP_OK;
}

int alp_kernel_time_ms(float *ms, int *sections) {
    if (int rc = require_init()) return rc;
    ALP_REQUIRE(ms, "ms is NULL");
    ALP_HIP(hipStreamSynchronize(ctx().stream));
    float sum = 0;
    for (int i = 0; i < g_kt.n; ++i) {
        float t = 0;
        ALP_HIP(hipEventElapsedTime(&t, g_kt.a[i], g_kt.b[i]));
        sum += t;
    }
    *ms = sum;
    if (sections) *sections = g_kt.n;
    g_kt.n = 0;
    return ALP_OK;
}

// Development switches compiled into this library, as a comma-separated list ("" = a release build): the
// translation units of the render keep timing / census / stage-skipping builds behind ALP_DEV_* macros, several
// of which produce wrong images by design; tests assert that the shipped library reports none.
const char *alp_build_flags(void) { return raster_dev_flags(); }

int alp_synchronize(void) {
    if (int rc = require_init()) return rc;
    ALP_HIP(hipStreamSynchronize(ctx().stream));
    return ALP_OK;
}

int alp_event_record(int slot) {
    if (int rc = require_init()) return rc;
    ALP_REQUIRE(slot >= 0 && slot < 64, "slot out of range");
    ALP_HIP(hipEventRecord(ctx().events[slot], ctx().stream));
    return ALP_OK;
}

int alp_event_elapsed_ms(int a, int b, float *ms) {
    if (int rc = require_init()) return rc;
    ALP_REQUIRE(a >= 0 && a < 64 && b >= 0 && b < 64 && ms, "bad slot or NULL");
    ALP_HIP(hipEventSynchronize(ctx().events[b]));
    ALP_HIP(hipEventElapsedTime(ms, ctx().events[a], ctx().events[b]));
    return ALP_OK;
}

// ------------------------------------------------------------------ RCCL
int alp_comm_unique_id(char id[ALP_UNIQUE_ID_BYTES]) {
    ALP_REQUIRE(id, "id is NULL");
    static_assert(sizeof(ncclUniqueId) == ALP_UNIQUE_ID_BYTES, "ncclUniqueId size");
    ncclUniqueId uid;
    ALP_NCCL(ncclGetUniqueId(&uid));
    memcpy(id, &uid, sizeof(uid));
    return ALP_OK;
}

int alp_comm_init(const char id[ALP_UNIQUE_ID_BYTES], int rank, int world_size) {
    if (int rc = require_init()) return rc;
    ALP_REQUIRE(id, "id is NULL");
    ALP_REQUIRE(world_size >= 1 && rank >= 0 && rank < world_size, "bad rank/world_size");
    Context &c = ctx();
    if (c.comm) return fail(ALP_EINVAL, "communicator already exists");
    ncclUniqueId uid;
    memcpy(&uid, id, sizeof(uid));
    ncclComm_t comm;
    ALP_HIP(hipSetDevice(c.device));
    const ncclResult_t r = ncclCommInitRank(&comm, world_size, uid, rank);
    if (r != ncclSuccess) {
        // The first contact of N ranks happens on a machine nobody may be watching: say everything RCCL knows and everything
        // about this rank that decides whether its peers can reach it (device, bus, the variables RCCL / ROCr read).
        char bus[32] = "?";
        hipDeviceGetPCIBusId(bus, (int)sizeof(bus), c.device);
        auto env = [](const char *k) { const char *v = getenv(k); return v ? v : "(unset)"; };
        const char *last = ncclGetLastError(nullptr);
        return fail(ALP_ERCCL,
                    "ncclCommInitRank failed: %s | RCCL says: %s | rank %d of %d on HIP device %d (pci %s) | "
                    "NCCL_DEBUG=%s HSA_ENABLE_IPC_MODE_LEGACY=%s HIP_VISIBLE_DEVICES=%s ROCR_VISIBLE_DEVICES=%s NCCL_SOCKET_IFNAME=%s | "
                    "hints: NCCL_DEBUG=INFO prints RCCL's own log; every rank needs a DIFFERENT device and the SAME 128-byte id; "
                    "hosts whose driver only supports dmabuf IPC need HSA_ENABLE_IPC_MODE_LEGACY=0 (hipIpcGetMemHandle: invalid argument otherwise)",
                    ncclGetErrorString(r), (last && *last) ? last : "(no further text)", rank, world_size, c.device, bus,
                    env("NCCL_DEBUG"), env("HSA_ENABLE_IPC_MODE_LEGACY"), env("HIP_VISIBLE_DEVICES"), env("ROCR_VISIBLE_DEVICES"),
                    env("NCCL_SOCKET_IFNAME"));
    }
    c.comm = (void *)comm;
    c.rank = rank;
    c.world = world_size;
    return ALP_OK;
}

int alp_comm_destroy(void) {
    Context &c = ctx();
    if (c.comm) {
        hipStreamSynchronize(c.stream);
        ncclCommDestroy((ncclComm_t)c.comm);
        c.comm = nullptr;
    }
    c.rank = 0;
    c.world = 1;
    return ALP_OK;
}

int alp_comm_bcast(void *buf, int64_t bytes, int root) {
    if (int rc = require_init()) return rc;
    ALP_REQUIRE(bytes >= 0 && (bytes == 0 || buf), "bad buffer");
    Context &c = ctx();
    if (!c.comm || bytes == 0) return ALP_OK;
    ALP_REQUIRE(root >= 0 && root < c.world, "root out of range");
    void *dev = nullptr;
    if (int rc = scratch_reserve((size_t)bytes, &dev)) return rc;
    if (c.rank == root) ALP_HIP(hipMemcpyAsync(dev, buf, (size_t)bytes, hipMemcpyHostToDevice, c.stream));
    ALP_NCCL(ncclBroadcast(dev, dev, (size_t)bytes, ncclChar, root, (ncclComm_t)c.comm, c.stream));
    ALP_HIP(hipMemcpyAsync(buf, dev, (size_t)bytes, hipMemcpyDeviceToHost, c.stream));
    ALP_HIP(hipStreamSynchronize(c.stream));
    return ALP_OK;
}

// Every rank contributes `bytes` bytes (its own count); counts[world] receives all of them in rank order.
int alp_comm_allgather_counts(int64_t bytes, int64_t *counts) {
    if (int rc = require_init()) return rc;
    ALP_REQUIRE(bytes >= 0 && counts, "bad argument");
    Context &c = ctx();
    if (!c.comm) {
        counts[0] = bytes;
        return ALP_OK;
    }
    int64_t *dev = nullptr;
    if (int rc = scratch_reserve((size_t)(c.world + 1) * sizeof(int64_t), (void **)&dev)) return rc;
    ALP_HIP(hipMemcpyAsync(dev + c.world, &bytes, sizeof(int64_t), hipMemcpyHostToDevice, c.stream));
    ALP_NCCL(ncclAllGather(dev + c.world, dev, 1, ncclInt64, (ncclComm_t)c.comm, c.stream));
    ALP_HIP(hipMemcpyAsync(counts, dev, (size_t)c.world * sizeof(int64_t), hipMemcpyDeviceToHost, c.stream));
    ALP_HIP(hipStreamSynchronize(c.stream));
    return ALP_OK;
}

// recv receives rank 0's send buffer, then rank 1's, ...: counts[r] bytes each (as alp_comm_allgather_counts returned
// them; counts[own rank] must be the size of `send`).  One ncclBroadcast per rank inside one group, staged through
// the device scratch.
int alp_comm_allgatherv(const void *send, void *recv, const int64_t *counts) {
    if (int rc = require_init()) return rc;
    ALP_REQUIRE(recv && counts, "NULL argument");
    Context &c = ctx();
    int64_t total = 0;
    for (int r = 0; r < c.world; ++r) {
        ALP_REQUIRE(counts[r] >= 0, "negative count");
        total += counts[r];
    }
    ALP_REQUIRE(counts[c.rank] == 0 || send, "send is NULL");
    if (!c.comm) {
        if (counts[0]) memcpy(recv, send, (size_t)counts[0]);
        return ALP_OK;
    }
    if (total == 0) return ALP_OK;
    char *dev = nullptr;
    if (int rc = scratch_reserve((size_t)total, (void **)&dev)) return rc;
    int64_t off = 0, mine = 0;
    for (int r = 0; r < c.rank; ++r) mine += counts[r];
    if (counts[c.rank]) ALP_HIP(hipMemcpyAsync(dev + mine, send, (size_t)counts[c.rank], hipMemcpyHostToDevice, c.stream));
    ALP_NCCL(ncclGroupStart());
    for (int r = 0; r < c.world; ++r) {
        if (counts[r]) {
            ncclResult_t e = ncclBroadcast(dev + off, dev + off, (size_t)counts[r], ncclChar, r, (ncclComm_t)c.comm, c.stream);
            if (e != ncclSuccess) {
                ncclGroupEnd();
                return fail(ALP_ERCCL, "ncclBroadcast failed: %s", ncclGetErrorString(e));
            }
        }
        off += counts[r];
    }
    ALP_NCCL(ncclGroupEnd());
    ALP_HIP(hipMemcpyAsync(recv, dev, (size_t)total, hipMemcpyDeviceToHost, c.stream));
    ALP_HIP(hipStreamSynchronize(c.stream));
    return ALP_OK;
}

int alp_comm_info(int *rank, int *world_size) {
    if (rank) *rank = ctx().rank;
    if (world_size) *world_size = ctx().world;
    return ALP_OK;
}

}  // extern "C"
