// libalproj_hip.so -- library context: device selection, stream, event timers, RCCL communicator.
// (Error reporting, the float64 folding of one camera pose and the alp_host_* helpers: host/alp_host.cpp.)
#include "alp_internal.h"

#include <rccl/rccl.h>
#include <cmath>
#include <cstdlib>
#include <vector>

namespace alp {

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

}  // namespace alp

using namespace alp;

extern "C" {

int alp_abi_version(void) { return ALP_ABI_VERSION; }

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
