"""ctypes binding of libalproj_hip.so (the C ABI declared in include/alproj_hip.h).

There is no CPU fallback: if the shared library is missing, or no MI355X-class HIP device
can be initialised, the first call that needs the device raises ``AlprojHipError``.
"""
import ctypes
import os
import threading
import weakref

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("ALPROJ_HIP_LIB", os.path.join(_HERE, "libalproj_hip.so"))   # override: dev ablation builds

ALP_F32, ALP_F64, ALP_I32, ALP_I64, ALP_U8, ALP_U16 = 0, 1, 2, 3, 4, 5
LOSS_MEAN_DIST, LOSS_HUBER = 0, 1
NPARAM = 25
UNIQUE_ID_BYTES = 128

PARAM_KEYS = ("x", "y", "z", "fov", "pan", "tilt", "roll", "a1", "a2",
              "k1", "k2", "k3", "k4", "k5", "k6", "p1", "p2",
              "s1", "s2", "s3", "s4", "w", "h", "cx", "cy")
DIST_KEYS = PARAM_KEYS[7:21]


# ---- recycled result memory
# A large result (the 237 MB raster of a georectified photograph, a 63 MB simulated image, the columns of a reverse_proj
# table) lands in a fresh numpy array; the device -> host copy into pages nobody has touched yet runs at 8-13 GB/s -- the
# kernel hands them out one fault at a time -- against 56 GB/s into pages that exist (tools/d2h_rate.hip).  So the memory of
# a result the caller has dropped (the array AND every view of it) is kept, up to a cap, and the next result of the same
# size is written into it: a series of photographs through one camera model pays for its pages once.  The arrays behave
# like np.empty's except for ``flags.owndata`` -- and, like np.empty's, they hold whatever bytes were there before (a previous
# result's): the library overwrites the whole array.  Kept buffers are resident memory (pre-faulted): clear_result_pool()
# gives them back, set_result_pool(0) turns the pool off.
_POOL_MIN = 8 << 20
try:
    _pool_cap = min(4 << 30, os.sysconf("SC_PHYS_PAGES") * os.sysconf("SC_PAGE_SIZE") // 16)     # at most a sixteenth of the host's memory
except (ValueError, OSError, AttributeError):
    _pool_cap = 1 << 30
_POOL_PER_SIZE = 2                  # a result and the one before it; more of one size is a leak, not a cache
_pool = {}                           # nbytes -> [backing, ...] (newest last)
_pool_age = []                       # (nbytes, id(backing)) in the order they came back: eviction takes the oldest first
_pool_bytes = 0
_pool_lock = threading.RLock()       # re-entrant: a garbage collection inside the locked region may run another result's finalizer
POOL_STATS = {"hits": 0, "misses": 0, "evicted": 0}


def _pool_drop_oldest(spare_size=None):
    """forget the buffer that has waited longest (of another size than ``spare_size`` if there is one); False when none is left"""
    global _pool_bytes
    pick = next((i for i, (nb, _) in enumerate(_pool_age) if nb != spare_size), 0 if _pool_age else None)
    if pick is None:
        return False
    nb, ident = _pool_age.pop(pick)
    bufs = _pool.get(nb, [])
    for j, b in enumerate(bufs):
        if id(b) == ident:
            del bufs[j]
            break
    if not bufs:
        _pool.pop(nb, None)
    _pool_bytes -= nb
    POOL_STATS["evicted"] += 1
    return True


def clear_result_pool():
    """Give back every kept buffer now (they are pre-faulted, so each counts in full against the process's resident set:
    up to the cap -- 4 GiB by default -- for as long as it is kept).  Results still held by the caller are not touched."""
    global _pool_bytes
    with _pool_lock:
        _pool.clear()
        del _pool_age[:]
        _pool_bytes = 0


def set_result_pool(cap_bytes):
    """Keep at most ``cap_bytes`` of dropped result memory for reuse (default 4 GiB or a sixteenth of the host's memory,
    whichever is less; 0: none).  What is kept beyond a new, smaller cap is released, oldest first.  Kept memory is
    resident memory: see clear_result_pool()."""
    global _pool_cap
    with _pool_lock:
        _pool_cap = int(cap_bytes)
        while _pool_bytes > _pool_cap and _pool_drop_oldest():
            pass


def _pool_release(backing):
    """A result's memory comes back.  It is the newest entry; to make room under the cap the oldest buffers go first --
    those of OTHER sizes before its own (a workload that changed its raster size must not keep the old size's buffers
    resident for ever and recycle nothing) -- and no size keeps more than _POOL_PER_SIZE."""
    global _pool_bytes
    nb = backing.nbytes
    with _pool_lock:
        if nb > _pool_cap:
            return
        same = _pool.get(nb, [])
        while len(same) >= _POOL_PER_SIZE:
            ident = id(same.pop(0))
            _pool_age[:] = [e for e in _pool_age if e != (nb, ident)]
            _pool_bytes -= nb
            POOL_STATS["evicted"] += 1
        while _pool_bytes + nb > _pool_cap and _pool_drop_oldest(spare_size=nb):
            pass
        while _pool_bytes + nb > _pool_cap and _pool_drop_oldest():
            pass
        _pool.setdefault(nb, []).append(backing)
        _pool_age.append((nb, id(backing)))
        _pool_bytes += nb


def _prefault(a):
    """a fresh array's pages, made by four threads in one go instead of one fault at a time inside the copy (alp_host_prefault)"""
    try:
        load().alp_host_prefault(a.ctypes.data_as(_c_void_p), a.nbytes, 0)
    except (OSError, AttributeError, AlprojHipError):       # no library: the copy that would have filled the array cannot happen either
        pass


def result_empty(shape, dtype):
    """np.empty(shape, dtype) for a result the device writes in full; large ones come from recycled memory."""
    global _pool_bytes
    dtype = np.dtype(dtype)
    shape = (int(shape),) if np.isscalar(shape) else tuple(int(v) for v in shape)
    nbytes = int(np.prod(shape, dtype=np.int64)) * dtype.itemsize
    if nbytes < _POOL_MIN:
        return np.empty(shape, dtype=dtype)
    if nbytes > _pool_cap:
        out = np.empty(shape, dtype=dtype)
        _prefault(out)
        return out
    with _pool_lock:
        free = _pool.get(nbytes)
        backing = free.pop() if free else None
        if backing is not None:
            _pool_age[:] = [e for e in _pool_age if e != (nbytes, id(backing))]
            if not free:
                _pool.pop(nbytes, None)
            _pool_bytes -= nbytes
            POOL_STATS["hits"] += 1
        else:
            POOL_STATS["misses"] += 1
    if backing is None:
        backing = np.empty(nbytes, dtype=np.uint8)
        _prefault(backing)
    # the array and all its views hold `window`; when the last of them is gone the memory goes back to the pool
    window = (ctypes.c_ubyte * nbytes).from_address(backing.ctypes.data)
    weakref.finalize(window, _pool_release, backing).atexit = False
    return np.frombuffer(window, dtype=dtype).reshape(shape)


class AlprojHipError(RuntimeError):
    def __init__(self, code, message):
        super().__init__(f"libalproj_hip error {code}: {message}")
        self.code = code


_c_void_p = ctypes.c_void_p
_c_int = ctypes.c_int
_c_i64 = ctypes.c_int64
_c_double = ctypes.c_double
_c_dp = ctypes.POINTER(ctypes.c_double)
_c_fp = ctypes.POINTER(ctypes.c_float)

# name -> argtypes; every function returns int unless listed in _RESTYPE
_SIGNATURES = {
    "alp_abi_version": [],
    "alp_last_error": [],
    "alp_init": [_c_int],
    "alp_shutdown": [],
    "alp_device_count": [ctypes.POINTER(_c_int)],
    "alp_device_info": [ctypes.c_char_p, _c_int, ctypes.POINTER(_c_int), ctypes.POINTER(_c_i64)],
    "alp_device_pci_bus_id": [ctypes.c_char_p, _c_int],
    "alp_host_hash64": [_c_void_p, _c_i64, _c_int, ctypes.POINTER(ctypes.c_uint64)],
    "alp_host_minmax": [_c_dp, _c_i64, _c_int, _c_dp],
    "alp_host_prefault": [_c_void_p, _c_i64, _c_int],
    "alp_synchronize": [],
    "alp_event_record": [_c_int],
    "alp_event_elapsed_ms": [_c_int, _c_int, _c_fp],
    "alp_kernel_timing": [_c_int],
    "alp_kernel_time_ms": [_c_fp, ctypes.POINTER(_c_int)],
    "alp_build_flags": [],
    "alp_comm_unique_id": [ctypes.c_char_p],
    "alp_comm_init": [ctypes.c_char_p, _c_int, _c_int],
    "alp_comm_destroy": [],
    "alp_comm_info": [ctypes.POINTER(_c_int), ctypes.POINTER(_c_int)],
    "alp_comm_bcast": [_c_void_p, _c_i64, _c_int],
    "alp_comm_allgather_counts": [_c_i64, ctypes.POINTER(_c_i64)],
    "alp_comm_allgatherv": [_c_void_p, _c_void_p, ctypes.POINTER(_c_i64)],
    "alp_points_create": [_c_void_p, _c_int, _c_i64, _c_dp, _c_int, ctypes.POINTER(_c_void_p)],
    "alp_points_create_columns": [_c_void_p, _c_void_p, _c_void_p, _c_int, _c_i64, _c_dp, _c_int, ctypes.POINTER(_c_void_p)],
    "alp_points_destroy": [_c_void_p],
    "alp_points_count": [_c_void_p, ctypes.POINTER(_c_i64)],
    "alp_points_set_observed": [_c_void_p, _c_void_p, _c_int],
    "alp_points_set_observed_columns": [_c_void_p, _c_void_p, _c_void_p, _c_int],
    "alp_project": [_c_void_p, _c_dp],
    "alp_projected_fetch": [_c_void_p, _c_void_p, _c_void_p, _c_int],
    "alp_projected_fetch_strided": [_c_void_p, _c_i64, _c_i64, _c_i64, _c_dp, _c_dp],
    "alp_residuals": [_c_void_p, _c_dp, _c_dp],
    "alp_residuals_batch": [_c_void_p, _c_dp, _c_i64, _c_dp],
    "alp_eval_population": [_c_void_p, _c_dp, _c_i64, _c_int, _c_double, _c_dp, ctypes.POINTER(_c_i64)],
    "alp_eval_population_enqueue": [_c_void_p, _c_dp, _c_i64, _c_int, _c_double],
    "alp_eval_population_wait": [_c_void_p, _c_dp, ctypes.POINTER(_c_i64)],
    "alp_eval_population_timing": [_c_void_p, _c_fp, _c_fp],
    "alp_eval_population_info": [_c_void_p, ctypes.POINTER(_c_i64)],
    "alp_loss_uv": [_c_dp, _c_dp, _c_i64, _c_int, _c_double, _c_dp],
    "alp_loss_uv_columns": [_c_dp, _c_dp, _c_dp, _c_dp, _c_i64, _c_int, _c_double, _c_dp],
    "alp_cma_sample": [_c_dp, _c_double, _c_dp, _c_dp, _c_dp, _c_int, _c_i64, _c_int, ctypes.c_uint64, ctypes.c_uint64, _c_dp,
                       ctypes.POINTER(ctypes.c_int32)],
    "alp_mesh_create": [_c_void_p, _c_int, _c_void_p, _c_int, _c_i64, _c_void_p, _c_int, _c_i64, _c_i64, _c_i64,
                        ctypes.POINTER(_c_void_p)],
    "alp_mesh_destroy": [_c_void_p],
    "alp_mesh_info": [_c_void_p, ctypes.POINTER(_c_i64)],
    "alp_render": [_c_void_p, _c_dp, _c_dp, _c_double, _c_fp],
    "alp_render_enqueue": [_c_void_p, _c_dp, _c_dp, _c_double],
    "alp_render_fetch": [_c_void_p, _c_fp],
    "alp_render_fetch_visibility": [_c_void_p, ctypes.POINTER(ctypes.c_uint64)],
    "alp_render_fetch_u8": [_c_void_p, ctypes.c_float, _c_int, ctypes.POINTER(ctypes.c_uint8)],
    "alp_mesh_set_value": [_c_void_p, _c_void_p, _c_int],
    "alp_mesh_frame_counts": [_c_void_p, ctypes.POINTER(_c_i64)],
    "alp_mesh_frame_ms": [_c_void_p, _c_fp],
    "alp_mesh_trim": [_c_void_p],
    "alp_mesh_set_value_source": [_c_void_p, _c_int],
    "alp_mesh_set_valid": [_c_void_p, ctypes.POINTER(ctypes.c_uint8)],
    "alp_mesh_from_rasters": [_c_void_p, _c_int, _c_i64, _c_i64, _c_dp, ctypes.c_double, _c_void_p, _c_int,
                              ctypes.c_double, ctypes.POINTER(ctypes.c_uint8), _c_dp, ctypes.POINTER(_c_void_p)],
    "alp_mesh_fetch": [_c_void_p, _c_fp, _c_fp, ctypes.POINTER(ctypes.c_uint8)],
    "alp_render_load": [_c_void_p, _c_fp, _c_i64, _c_i64],
    "alp_render_valid_count": [_c_void_p, ctypes.POINTER(_c_i64)],
    "alp_render_fetch_valid": [_c_void_p, _c_dp, ctypes.POINTER(ctypes.c_uint32), _c_dp],
    "alp_render_fetch_valid_planes": [_c_void_p, _c_dp, ctypes.POINTER(ctypes.c_uint32), _c_dp, _c_dp, _c_dp],
    "alp_render_gather": [_c_void_p, ctypes.POINTER(ctypes.c_int32), ctypes.POINTER(ctypes.c_int32), _c_i64, _c_dp,
                          _c_dp],
    "alp_render_fetch_valid_table": [_c_void_p, _c_dp, _c_void_p, _c_int, _c_i64, ctypes.POINTER(ctypes.c_int64),
                                     ctypes.POINTER(ctypes.c_int16), ctypes.POINTER(ctypes.c_int16), _c_dp],
    "alp_rasterize_columns": [_c_dp, _c_dp, ctypes.POINTER(ctypes.c_void_p), _c_i64, _c_i64, _c_double, _c_double, _c_double,
                              _c_i64, _c_i64, _c_int, _c_int, _c_int, ctypes.POINTER(ctypes.c_uint8)],
    "alp_distance_mask": [_c_dp, _c_i64, _c_dp, _c_double, _c_double, ctypes.POINTER(ctypes.c_uint8)],
    "alp_rasterize_points": [_c_dp, _c_dp, _c_dp, _c_i64, _c_i64, _c_double, _c_double, _c_double, _c_i64, _c_i64,
                             _c_int, _c_int, _c_int, ctypes.POINTER(ctypes.c_uint8)],
    "alp_rasterize_points_f32": [_c_dp, _c_dp, _c_dp, _c_i64, _c_i64, _c_double, _c_double, _c_double, _c_i64, _c_i64,
                                 _c_int, _c_int, _c_fp],
    "alp_render_rasterize_plan": [_c_void_p, _c_dp, ctypes.POINTER(_c_i64), _c_dp],
    "alp_render_rasterize": [_c_void_p, _c_void_p, _c_int, _c_i64, ctypes.POINTER(ctypes.c_int32), _c_i64, _c_double, _c_double,
                             _c_double, _c_i64, _c_i64, _c_int, _c_int, _c_int, ctypes.POINTER(ctypes.c_uint8)],
    "alp_distort_image": [_c_fp, _c_i64, _c_i64, _c_i64, _c_dp, _c_fp],
    "alp_distort_map": [_c_i64, _c_i64, _c_dp, _c_fp, _c_fp],
}
_RESTYPE = {"alp_last_error": ctypes.c_char_p, "alp_build_flags": ctypes.c_char_p}

_lock = threading.Lock()
_lib = None
_device = None


def load():
    """dlopen the library and declare prototypes (does not touch the GPU)."""
    global _lib
    with _lock:
        if _lib is None:
            if not os.path.exists(LIB_PATH):
                raise AlprojHipError(
                    -100, f"{LIB_PATH} not found: build it with "
                          "`python -m alproj_amd._build` (hipcc, gfx950); there is no CPU fallback")
            lib = ctypes.CDLL(LIB_PATH)
            for name, args in _SIGNATURES.items():
                fn = getattr(lib, name)
                fn.argtypes = args
                fn.restype = _RESTYPE.get(name, _c_int)
            _lib = lib
    return _lib


def check(rc):
    if rc != 0:
        raise AlprojHipError(rc, (load().alp_last_error() or b"").decode(errors="replace"))


def init(device=None):
    """Initialise the library on a device (default: LOCAL_RANK or 0).  Idempotent."""
    global _device
    lib = load()
    if device is None:
        device = _device if _device is not None else int(os.environ.get("LOCAL_RANK", "0"))
    check(lib.alp_init(int(device)))
    _device = int(device)
    return lib


def lib():
    """The initialised library (initialises device 0 / LOCAL_RANK on first use)."""
    return init()


def device_count():
    """HIP devices this process can see (alp_device_count; 0 without a driver or a GPU)"""
    n = _c_int()
    try:
        check(load().alp_device_count(ctypes.byref(n)))
    except AlprojHipError:
        return 0
    return int(n.value)


def device_info():
    l = lib()
    name = ctypes.create_string_buffer(128)
    cu = _c_int()
    mem = _c_i64()
    check(l.alp_device_info(name, 128, ctypes.byref(cu), ctypes.byref(mem)))
    bus = ctypes.create_string_buffer(32)
    check(l.alp_device_pci_bus_id(bus, 32))
    return {"arch": name.value.decode(), "cu_count": cu.value, "hbm_bytes": mem.value, "pci_bus_id": bus.value.decode()}


def host_hash64(a, threads=0):
    """64-bit content digest of a C-contiguous numpy array (alp_host_hash64; host threads, no device needed)"""
    a = np.asarray(a)
    if not a.flags["C_CONTIGUOUS"]:
        raise ValueError("array must be C-contiguous")
    d = ctypes.c_uint64()
    check(load().alp_host_hash64(a.ctypes.data_as(_c_void_p), a.nbytes, int(threads), ctypes.byref(d)))
    return int(d.value)


def host_minmax(a, threads=0):
    """(a.min(), a.max()) of a C-contiguous float64 array in one threaded pass (alp_host_minmax; NaN, NaN if any value is
    NaN, like numpy; no device needed).  An empty array raises numpy's own error."""
    a = np.ascontiguousarray(a, dtype=np.float64)
    if a.size == 0:
        return a.min(), a.max()
    out = (ctypes.c_double * 2)()
    check(load().alp_host_minmax(a.ctypes.data_as(_c_dp), a.size, int(threads), out))
    return np.float64(out[0]), np.float64(out[1])


def params_vector(params):
    """dict -> float64[25] in ABI order (raises KeyError on a missing key like the reference)."""
    # None is meaningful for cx / cy only; float(None) raises for every other key, as the reference's arithmetic does
    out = np.array([0.0 if (params[k] is None and k in ("cx", "cy")) else float(params[k]) for k in PARAM_KEYS], dtype=np.float64)
    # the reference's intrinsic_mat substitutes w/2, h/2 for cx/cy = None (optimize.py:27-30)
    if params["cx"] is None:
        out[23] = out[21] / 2
    if params["cy"] is None:
        out[24] = out[22] / 2
    return out


def as_dp(a):
    return a.ctypes.data_as(_c_dp)


def as_fp(a):
    return a.ctypes.data_as(_c_fp)


def dtype_code(a):
    if a.dtype == np.float64:
        return ALP_F64
    if a.dtype == np.float32:
        return ALP_F32
    raise TypeError(f"unsupported dtype {a.dtype}")


PRECISIONS = {"f32": ALP_F32, "f64": ALP_F64, ALP_F32: ALP_F32, ALP_F64: ALP_F64}


def _same_float_columns(cols, k):
    """k one-dimensional contiguous columns of one float type (float32 only if all of them are) and one length"""
    cols = [np.asarray(c) for c in cols]
    if len(cols) != k or any(c.ndim != 1 for c in cols) or len({len(c) for c in cols}) != 1:
        raise ValueError(f"need {k} one-dimensional columns of one length")
    dt = np.float32 if all(c.dtype == np.float32 for c in cols) else np.float64
    return [np.ascontiguousarray(c, dtype=dt) for c in cols]


class Points:
    """Device-resident point set (RAII wrapper of alp_points_t)."""

    def __init__(self, xyz, origin, precision="f32", _columns=None):
        """xyz: an (N, 3) array -- uploaded as it lies when it is row-major, and column by column when its COLUMNS are the
        contiguous runs (the transposed view pandas hands out for a block of float columns): no host-side interleaving or
        transposition either way.  ``Points.from_columns(x, y, z, ...)`` takes three 1-D columns."""
        l = lib()
        cols = _columns
        if cols is None:
            xyz = np.asarray(xyz)
            if xyz.ndim != 2 or xyz.shape[1] != 3:
                raise ValueError("xyz must have shape (N, 3)")
            if xyz.dtype not in (np.float32, np.float64):
                xyz = xyz.astype(np.float64)
            if not xyz.flags["C_CONTIGUOUS"] and xyz.shape[0] > 1 and all(xyz[:, k].flags["C_CONTIGUOUS"] for k in range(3)):
                cols = [xyz[:, k] for k in range(3)]
        self.precision = PRECISIONS[precision]
        self.origin = np.ascontiguousarray(origin, dtype=np.float64).reshape(3)
        h = _c_void_p()
        if cols is not None:
            cols = _same_float_columns(cols, 3)
            self.n = int(len(cols[0]))
            check(l.alp_points_create_columns(*[c.ctypes.data_as(_c_void_p) for c in cols], dtype_code(cols[0]), self.n,
                                              as_dp(self.origin), self.precision, ctypes.byref(h)))
        else:
            xyz = np.ascontiguousarray(xyz)
            self.n = int(xyz.shape[0])
            check(l.alp_points_create(xyz.ctypes.data_as(_c_void_p), dtype_code(xyz), self.n,
                                      as_dp(self.origin), self.precision, ctypes.byref(h)))
        self._h = h
        self._lib = l

    @classmethod
    def from_columns(cls, x, y, z, origin, precision="f32"):
        """The point set from its three columns as they lie (alp_points_create_columns)"""
        return cls(None, origin, precision, _columns=[x, y, z])

    def close(self):
        if getattr(self, "_h", None) is not None and self._h:
            self._lib.alp_points_destroy(self._h)
            self._h = None

    __del__ = close

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def set_observed(self, uv):
        """uv: (N, 2), row-major or with contiguous columns (uploaded column by column then)"""
        uv = np.asarray(uv)
        if uv.dtype not in (np.float32, np.float64):
            uv = uv.astype(np.float64)
        if uv.shape != (self.n, 2):
            raise ValueError(f"observed uv must have shape ({self.n}, 2)")
        if not uv.flags["C_CONTIGUOUS"] and self.n > 1 and uv[:, 0].flags["C_CONTIGUOUS"] and uv[:, 1].flags["C_CONTIGUOUS"]:
            return self.set_observed_columns(uv[:, 0], uv[:, 1])
        uv = np.ascontiguousarray(uv)
        check(self._lib.alp_points_set_observed(self._h, uv.ctypes.data_as(_c_void_p), dtype_code(uv)))

    def set_observed_columns(self, u, v):
        """the observed pixels from their two columns as they lie (alp_points_set_observed_columns)"""
        cols = _same_float_columns([u, v], 2)
        if len(cols[0]) != self.n:
            raise ValueError(f"observed u, v must have {self.n} elements")
        check(self._lib.alp_points_set_observed_columns(self._h, cols[0].ctypes.data_as(_c_void_p), cols[1].ctypes.data_as(_c_void_p),
                                                        dtype_code(cols[0])))

    def project(self, pvec):
        pvec = np.ascontiguousarray(pvec, dtype=np.float64)
        check(self._lib.alp_project(self._h, as_dp(pvec)))

    def fetch(self, dtype=np.float64):
        u = result_empty(self.n, dtype)
        v = result_empty(self.n, dtype)
        check(self._lib.alp_projected_fetch(self._h, u.ctypes.data_as(_c_void_p),
                                            v.ctypes.data_as(_c_void_p), dtype_code(u)))
        return u, v

    def fetch_strided(self, first, stride, count):
        u = result_empty(count, np.float64)
        v = result_empty(count, np.float64)
        check(self._lib.alp_projected_fetch_strided(self._h, first, stride, count, as_dp(u), as_dp(v)))
        return u, v

    def residuals(self, pvec):
        pvec = np.ascontiguousarray(pvec, dtype=np.float64)
        out = result_empty(2 * self.n, np.float64)
        check(self._lib.alp_residuals(self._h, as_dp(pvec), as_dp(out)))
        return out

    def residuals_batch(self, cand):
        """(B, 25) parameter vectors -> (B, 2N) residual vectors, one launch."""
        cand = np.ascontiguousarray(cand, dtype=np.float64)
        if cand.ndim != 2 or cand.shape[1] != NPARAM:
            raise ValueError("cand must have shape (B, 25)")
        out = result_empty((cand.shape[0], 2 * self.n), np.float64)
        check(self._lib.alp_residuals_batch(self._h, as_dp(cand), cand.shape[0], as_dp(out)))
        return out

    def eval_population(self, cand, loss_kind, f_scale=10.0, want_argmin=True):
        """-> (losses (P,), argmin).  ``want_argmin=False``: losses only -- the library then skips the float64 confirmation
        of near-tied candidates of a float32 set, and the argmin returned is numpy's over the losses as they are (first
        index on ties, NaN never wins)."""
        cand = np.ascontiguousarray(cand, dtype=np.float64)
        if cand.ndim != 2 or cand.shape[1] != NPARAM:
            raise ValueError("cand must have shape (P, 25)")
        P = cand.shape[0]
        losses = np.empty(P, dtype=np.float64)
        amin = _c_i64()
        check(self._lib.alp_eval_population(self._h, as_dp(cand), P, int(loss_kind), float(f_scale),
                                            as_dp(losses), ctypes.byref(amin) if want_argmin else None))
        if not want_argmin:
            ok = ~np.isnan(losses)
            return losses, (int(np.flatnonzero(ok)[np.argmin(losses[ok])]) if ok.any() else 0)
        return losses, int(amin.value)

    def eval_population_enqueue(self, cand, loss_kind, f_scale=10.0):
        cand = np.ascontiguousarray(cand, dtype=np.float64)
        check(self._lib.alp_eval_population_enqueue(self._h, as_dp(cand), cand.shape[0],
                                                    int(loss_kind), float(f_scale)))
        return cand.shape[0]

    def eval_population_timing(self):
        """(kernel_ms, allreduce_ms) of the last completed population evaluation (HIP events)."""
        k, a = ctypes.c_float(), ctypes.c_float()
        check(self._lib.alp_eval_population_timing(self._h, ctypes.byref(k), ctypes.byref(a)))
        return float(k.value), float(a.value)

    POP_VARIANTS = ("general", "shared_pose", "lens_free")

    def eval_population_info(self):
        """(variant, stripes, tile columns) of the last population evaluation's launch: alp_eval_population_info"""
        info = (_c_i64 * 3)()
        check(lib().alp_eval_population_info(self._h, info))
        return self.POP_VARIANTS[int(info[0])], int(info[1]), int(info[2])

    def eval_population_wait(self, P):
        losses = np.empty(P, dtype=np.float64)
        amin = _c_i64()
        check(self._lib.alp_eval_population_wait(self._h, as_dp(losses), ctypes.byref(amin)))
        return losses, int(amin.value)


def _vertex_array(a):
    """float32 and float64 arrays go to the library as they are (float64 is cast on the device during the upload:
    the reference's ``astype("f4")``, project.py:213-214, without a host pass); anything else becomes float64 first."""
    a = np.asarray(a)
    if a.dtype not in (np.float32, np.float64):
        a = a.astype(np.float64)
    return np.ascontiguousarray(a)


class Mesh:
    """Device-resident triangle mesh (RAII wrapper of alp_mesh_t).

    ``ind`` may be an (T, 3) int32/int64 array, or None together with ``grid=(rows, cols)`` for
    the implicit regular grid of the reference's get_colored_surface."""

    def __init__(self, vert, value=None, ind=None, grid=None):
        l = lib()
        vert = _vertex_array(vert)
        if vert.ndim != 2 or vert.shape[1] != 3:
            raise ValueError("vert must have shape (N, 3)")
        if value is not None:
            value = _vertex_array(value)
            if value.shape != vert.shape:
                raise ValueError("value must have the shape of vert")
        if ind is None:
            if grid is None:
                raise ValueError("either ind or grid=(rows, cols) is required")
            gh, gw = int(grid[0]), int(grid[1])
            ind_p, code, n_tri = None, ALP_I32, 0
        else:
            ind = np.ascontiguousarray(ind)
            if ind.dtype not in (np.int32, np.int64):
                ind = ind.astype(np.int64)
            if ind.ndim != 2 or ind.shape[1] != 3:
                raise ValueError("ind must have shape (T, 3)")
            gh = gw = 0
            ind_p, code, n_tri = ind.ctypes.data_as(_c_void_p), (ALP_I64 if ind.dtype == np.int64 else ALP_I32), ind.shape[0]
        h = _c_void_p()
        check(l.alp_mesh_create(vert.ctypes.data_as(_c_void_p), dtype_code(vert),
                                None if value is None else value.ctypes.data_as(_c_void_p),
                                ALP_F32 if value is None else dtype_code(value), vert.shape[0], ind_p, code,
                                n_tri, gh, gw, ctypes.byref(h)))
        self._h, self._lib = h, l
        self.shape = None
        self.n_vert = vert.shape[0]
        self.has_value = value is not None

    def set_value(self, value):
        """Replace the stored per-vertex values (None drops them: value == vert)."""
        if value is None:
            check(self._lib.alp_mesh_set_value(self._h, None, ALP_F32))
        else:
            value = _vertex_array(value)
            if value.shape != (self.n_vert, 3):
                raise ValueError("value must have shape (n_vert, 3)")
            check(self._lib.alp_mesh_set_value(self._h, value.ctypes.data_as(_c_void_p), dtype_code(value)))
        self.has_value = value is not None

    def info(self):
        """dict(implicit, grid_h, grid_w, n_tri): how the library holds the mesh (alp_mesh_info)"""
        c = (_c_i64 * 4)()
        check(self._lib.alp_mesh_info(self._h, c))
        return dict(implicit=bool(c[0]), grid_h=int(c[1]), grid_w=int(c[2]), n_tri=int(c[3]))

    def frame_ms(self):
        """device ms of the launches of the last render_enqueue (waits for the frame)"""
        ms = ctypes.c_float()
        check(self._lib.alp_mesh_frame_ms(self._h, ctypes.byref(ms)))
        return float(ms.value)

    def trim(self):
        """release the rasterisation work areas kept between calls (alp_mesh_trim)"""
        check(self._lib.alp_mesh_trim(self._h))

    def frame_counts(self):
        """(full frames, frames served from the visibility cache by the resolve stage alone)"""
        c = (_c_i64 * 2)()
        check(self._lib.alp_mesh_frame_counts(self._h, c))
        return int(c[0]), int(c[1])

    def fetch_u8(self, scale=255.0, reverse_channels=True):
        """The last frame as (h, w, 3) uint8 = (image * scale).astype(uint8), channels reversed: sim_image's tail"""
        out = result_empty(self.shape, np.uint8)
        check(self._lib.alp_render_fetch_u8(self._h, float(scale), int(bool(reverse_channels)),
                                            out.ctypes.data_as(ctypes.POINTER(ctypes.c_uint8))))
        return out

    @classmethod
    def from_rasters(cls, dsm, transform, z_max, aerial, color_div, nodata=None):
        """Implicit-grid mesh built on the device from a filled DSM (rows, cols), the affine
        coefficients (a, b, c, d, e, f), three aerial bands (3, rows, cols; uint8 / uint16 /
        float32) and the DSM nodata mask (alp_mesh_from_rasters).  Returns (mesh, offsets)."""
        l = lib()
        dsm = np.ascontiguousarray(dsm)
        if dsm.dtype not in (np.float32, np.float64):
            dsm = dsm.astype(np.float64)
        if dsm.ndim != 2:
            raise ValueError("dsm must have shape (rows, cols)")
        aerial = np.ascontiguousarray(aerial)
        if aerial.dtype not in (np.uint8, np.uint16, np.float32):
            aerial = aerial.astype(np.float32)
        if aerial.shape != (3,) + dsm.shape:
            raise ValueError("aerial must have shape (3, rows, cols)")
        codes = {np.dtype(np.float32): ALP_F32, np.dtype(np.float64): ALP_F64, np.dtype(np.uint8): ALP_U8,
                 np.dtype(np.uint16): ALP_U16}
        t = np.ascontiguousarray(transform, dtype=np.float64)
        if t.shape != (6,):
            raise ValueError("transform must have the six affine coefficients a, b, c, d, e, f")
        nod = None
        if nodata is not None:
            nod = np.ascontiguousarray(nodata, dtype=np.uint8)
            if nod.shape != dsm.shape:
                raise ValueError("nodata must have the shape of dsm")
        off = np.empty(3, dtype=np.float64)
        h = _c_void_p()
        check(l.alp_mesh_from_rasters(dsm.ctypes.data_as(_c_void_p), codes[dsm.dtype], dsm.shape[0], dsm.shape[1],
                                      as_dp(t), float(z_max), aerial.ctypes.data_as(_c_void_p), codes[aerial.dtype],
                                      float(color_div),
                                      None if nod is None else nod.ctypes.data_as(ctypes.POINTER(ctypes.c_uint8)),
                                      as_dp(off), ctypes.byref(h)))
        self = cls.__new__(cls)
        self._h, self._lib, self.shape, self.n_vert = h, l, None, dsm.size
        self.has_value = True
        return self, off

    def set_valid(self, valid):
        """Per-vertex mask (falsy = nodata: its triangles are not drawn); None removes it."""
        if valid is None:
            check(self._lib.alp_mesh_set_valid(self._h, None))
            return
        valid = np.ascontiguousarray(np.asarray(valid).ravel() != 0, dtype=np.uint8)
        if valid.shape[0] != self.n_vert:
            raise ValueError("valid must have one entry per vertex")
        check(self._lib.alp_mesh_set_valid(self._h, valid.ctypes.data_as(ctypes.POINTER(ctypes.c_uint8))))

    def fetch_arrays(self):
        """(vert, value, valid) of the resident mesh as numpy arrays (inspection / tests)."""
        vert = result_empty((self.n_vert, 3), np.float32)
        value = result_empty((self.n_vert, 3), np.float32)
        valid = result_empty(self.n_vert, np.uint8)
        check(self._lib.alp_mesh_fetch(self._h, as_fp(vert), as_fp(value),
                                       valid.ctypes.data_as(ctypes.POINTER(ctypes.c_uint8))))
        return vert, value, valid.astype(bool)

    def close(self):
        if getattr(self, "_h", None) is not None and self._h:
            self._lib.alp_mesh_destroy(self._h)
            self._h = None

    __del__ = close

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def render_enqueue(self, pvec, offsets=None, min_distance=None, coords=False):
        """``coords=True`` renders the vertices themselves (reverse_proj) instead of the stored
        per-vertex values."""
        pvec = np.ascontiguousarray(pvec, dtype=np.float64)
        off = None if offsets is None else np.ascontiguousarray(offsets, dtype=np.float64)
        check(self._lib.alp_mesh_set_value_source(self._h, 1 if coords else 0))
        self.generation = getattr(self, "generation", 0) + 1      # lets holders of an older frame notice
        check(self._lib.alp_render_enqueue(self._h, as_dp(pvec), None if off is None else as_dp(off),
                                           0.0 if min_distance is None else float(min_distance)))
        self.shape = (int(pvec[22]), int(pvec[21]), 3)

    def fetch(self):
        out = result_empty(self.shape, np.float32)
        check(self._lib.alp_render_fetch(self._h, as_fp(out)))
        return out

    def fetch_visibility(self):
        out = result_empty(self.shape[:2], np.uint64)
        check(self._lib.alp_render_fetch_visibility(self._h, out.ctypes.data_as(ctypes.POINTER(ctypes.c_uint64))))
        return out

    def render(self, pvec, offsets=None, min_distance=None):
        self.render_enqueue(pvec, offsets, min_distance)
        return self.fetch()

    def load_image(self, image):
        """Install an (h, w, 3) float32 coordinate image produced elsewhere as the current frame
        (alp_render_load): ``fetch_valid`` / ``gather`` then run on it."""
        image = np.ascontiguousarray(image, dtype=np.float32)
        if image.ndim != 3 or image.shape[2] != 3:
            raise ValueError("image must have shape (h, w, 3)")
        check(self._lib.alp_render_load(self._h, as_fp(image), image.shape[0], image.shape[1]))
        self.generation = getattr(self, "generation", 0) + 1
        self.shape = image.shape

    def fetch_valid(self, offsets=None):
        """After a render of the vertices themselves: (idx, xyz) of the pixels that see the
        surface (first channel > 0), row-major; xyz = channels (0, 2, 1) + offsets, float64."""
        n = _c_i64()
        check(self._lib.alp_render_valid_count(self._h, ctypes.byref(n)))
        idx = result_empty(n.value, np.uint32)
        xyz = result_empty((n.value, 3), np.float64)
        off = None if offsets is None else np.ascontiguousarray(offsets, dtype=np.float64)
        check(self._lib.alp_render_fetch_valid(self._h, None if off is None else as_dp(off),
                                               idx.ctypes.data_as(ctypes.POINTER(ctypes.c_uint32)), as_dp(xyz)))
        return idx, xyz

    def rasterize_plan(self, offsets=None):
        """After a render of the vertices themselves: (number of pixels that see the surface, (x_min, y_min, x_max,
        y_max) of their coordinates); keeps the compacted points on the device for ``rasterize``."""
        n = _c_i64()
        b = np.empty(4, dtype=np.float64)
        off = None if offsets is None else np.ascontiguousarray(offsets, dtype=np.float64)
        check(self._lib.alp_render_rasterize_plan(self._h, None if off is None else as_dp(off), ctypes.byref(n), as_dp(b)))
        return int(n.value), tuple(float(v) for v in b)

    def rasterize(self, array, band_channel, x_min, y_max, resolution, width, height, agg, sweeps, nodata):
        """(bands, height, width) uint8 raster of the planned points with band values from ``array`` (h, w, C)."""
        array = np.ascontiguousarray(array)
        codes = {np.dtype(np.uint8): ALP_U8, np.dtype(np.uint16): ALP_U16, np.dtype(np.float32): ALP_F32,
                 np.dtype(np.float64): ALP_F64}
        if array.dtype not in codes:
            array = array.astype(np.float64)
        if array.ndim != 3 or array.shape[:2] != tuple(self.shape[:2]):
            raise ValueError("array must have shape (h, w, channels) of the rendered frame")
        bc = np.ascontiguousarray(band_channel, dtype=np.int32)
        out = result_empty((len(bc), int(height), int(width)), np.uint8)
        check(self._lib.alp_render_rasterize(self._h, array.ctypes.data_as(_c_void_p), codes[array.dtype], array.shape[2],
                                             bc.ctypes.data_as(ctypes.POINTER(ctypes.c_int32)), len(bc), float(x_min),
                                             float(y_max), float(resolution), int(width), int(height), int(agg),
                                             int(sweeps), int(nodata), out.ctypes.data_as(ctypes.POINTER(ctypes.c_uint8))))
        return out

    def fetch_valid_block(self, offsets=None, extra_rows=0):
        """Like ``fetch_valid`` but as (idx (M,), block (3 + extra_rows, M) float64): rows 0..2 = x, y, z, each contiguous
        (the layout of a DataFrame's float64 block); the extra rows are left for the caller's channel columns."""
        n = _c_i64()
        check(self._lib.alp_render_valid_count(self._h, ctypes.byref(n)))
        M = int(n.value)
        idx = result_empty(M, np.uint32)
        block = result_empty((3 + int(extra_rows), M), np.float64)
        off = None if offsets is None else np.ascontiguousarray(offsets, dtype=np.float64)
        check(self._lib.alp_render_fetch_valid_planes(self._h, None if off is None else as_dp(off),
                                                      idx.ctypes.data_as(ctypes.POINTER(ctypes.c_uint32)),
                                                      as_dp(block[0]), as_dp(block[1]), as_dp(block[2])))
        return idx, block

    TABLE_DTYPES = {np.dtype(np.uint8): ALP_U8, np.dtype(np.uint16): ALP_U16, np.dtype(np.float32): ALP_F32,
                    np.dtype(np.float64): ALP_F64}

    def fetch_valid_table(self, array, offsets=None):
        """The columns of reverse_proj's table for the pixels that see the surface, formed on the device: (labels int64 (M,),
        u int16 (M,), v int16 (M,), block (3 + C, M) float64 = x, y, z and the C channels of ``array`` (h, w, C) at those
        pixels).  ``array.dtype`` must be one of TABLE_DTYPES."""
        array = np.ascontiguousarray(array)
        code = self.TABLE_DTYPES[array.dtype]
        C = int(array.shape[2])
        n = _c_i64()
        check(self._lib.alp_render_valid_count(self._h, ctypes.byref(n)))
        M = int(n.value)
        labels = result_empty(M, np.int64)
        u = result_empty(M, np.int16)
        v = result_empty(M, np.int16)
        block = result_empty((3 + C, M), np.float64)
        off = None if offsets is None else np.ascontiguousarray(offsets, dtype=np.float64)
        i16 = ctypes.POINTER(ctypes.c_int16)
        check(self._lib.alp_render_fetch_valid_table(self._h, None if off is None else as_dp(off), array.ctypes.data_as(_c_void_p),
                                                     code, C, labels.ctypes.data_as(ctypes.POINTER(ctypes.c_int64)),
                                                     u.ctypes.data_as(i16), v.ctypes.data_as(i16), as_dp(block)))
        return labels, u, v, block

    def gather(self, u, v, offsets=None):
        """After a render of the vertices themselves: (n, 3) float64 x, y, z seen by the pixels
        (u[i], v[i]); NaN outside the image or where no surface is seen."""
        u = np.ascontiguousarray(u, dtype=np.int32)
        v = np.ascontiguousarray(v, dtype=np.int32)
        if u.shape != v.shape or u.ndim != 1:
            raise ValueError("u and v must be 1-D arrays of the same length")
        xyz = result_empty((len(u), 3), np.float64)
        off = None if offsets is None else np.ascontiguousarray(offsets, dtype=np.float64)
        ip = ctypes.POINTER(ctypes.c_int32)
        check(self._lib.alp_render_gather(self._h, u.ctypes.data_as(ip), v.ctypes.data_as(ip), len(u),
                                          None if off is None else as_dp(off), as_dp(xyz)))
        return xyz


def distance_mask(xyz, camera, min_distance=None, max_distance=None):
    """Boolean mask of the rows of xyz (n, 3) float64 whose distance from ``camera`` (x, y, z) lies in
    [min_distance, max_distance] (None = no bound); rows with a NaN coordinate are False (gcp.py:711-724)."""
    xyz = np.ascontiguousarray(xyz, dtype=np.float64)
    if xyz.ndim != 2 or xyz.shape[1] != 3:
        raise ValueError("xyz must be (n, 3)")
    cam = np.ascontiguousarray(camera, dtype=np.float64)
    keep = np.zeros(len(xyz), dtype=np.uint8)
    check(lib().alp_distance_mask(as_dp(xyz), len(xyz), as_dp(cam),
                                  float("nan") if min_distance is None else float(min_distance),
                                  float("nan") if max_distance is None else float(max_distance),
                                  keep.ctypes.data_as(ctypes.POINTER(ctypes.c_uint8))))
    return keep.astype(bool)


AGG_CODES = {"mean": 0, "max": 1, "min": 2, "median": 3}


def rasterize_points_f32(x, y, values, resolution=1.0, interpolate=True, max_dist=1.0, agg_func="mean"):
    """The float32 raster (bands, height, width) of the reference's to_geotiff before its byte conversion (project.py:448-479;
    NaN = empty cell) for points x, y (n,) with band values (n, bands): alp_rasterize_points_f32.  -> (raster, bounds)"""
    x = np.ascontiguousarray(x, dtype=np.float64)
    y = np.ascontiguousarray(y, dtype=np.float64)
    values = np.ascontiguousarray(values, dtype=np.float64)
    if values.ndim != 2 or values.shape[0] != len(x) or len(y) != len(x):
        raise ValueError("x, y (n,) and values (n, bands) expected")
    (x_min, x_max), (y_min, y_max) = host_minmax(x), host_minmax(y)
    width = int(np.ceil((x_max - x_min) / resolution))
    height = int(np.ceil((y_max - y_min) / resolution))
    if width <= 0 or height <= 0:
        raise ValueError(f"Invalid raster dimensions: width={width}, height={height}")
    sweeps = int(np.ceil(max_dist / resolution)) if (interpolate and max_dist > 0) else 0
    out = result_empty((values.shape[1], height, width), np.float32)
    check(lib().alp_rasterize_points_f32(as_dp(x), as_dp(y), as_dp(values), len(x), values.shape[1], float(x_min), float(y_max),
                                         float(resolution), width, height, AGG_CODES[agg_func], sweeps, as_fp(out)))
    return out, (x_min, y_min, x_max, y_max, width, height)


def distort_image(img, coeffs):
    img = np.ascontiguousarray(img, dtype=np.float32)
    if img.ndim not in (2, 3):
        raise ValueError("img must be (h, w) or (h, w, c)")
    h, w = img.shape[:2]
    c = 1 if img.ndim == 2 else img.shape[2]
    cf = np.ascontiguousarray(coeffs, dtype=np.float64)
    if cf.shape != (14,):
        raise ValueError("distort_coeffs must have 14 entries")
    out = np.empty_like(img)
    check(lib().alp_distort_image(as_fp(img), h, w, c, as_dp(cf), as_fp(out)))
    return out


def cma_sample(mean, sigma, BD, bounds, P, n_max_resampling, seed, generation, return_tries=False):
    """(P, D) candidates of one CMA-ES generation drawn on the device (alp_cma_sample): x = mean + sigma BD z,
    re-drawn until inside ``bounds`` (D, 2) at most ``n_max_resampling`` times, then clipped."""
    mean = np.ascontiguousarray(mean, dtype=np.float64)
    D = mean.shape[0]
    BD = np.ascontiguousarray(BD, dtype=np.float64)
    if BD.shape != (D, D):
        raise ValueError("BD must be (D, D)")
    lo = hi = None
    if bounds is not None:
        b = np.asarray(bounds, dtype=np.float64)
        lo, hi = np.ascontiguousarray(b[:, 0]), np.ascontiguousarray(b[:, 1])
    x = result_empty((int(P), D), np.float64)
    tries = np.empty(int(P), dtype=np.int32) if return_tries else None
    check(lib().alp_cma_sample(as_dp(mean), float(sigma), as_dp(BD), None if lo is None else as_dp(lo),
                               None if hi is None else as_dp(hi), D, int(P), int(n_max_resampling),
                               ctypes.c_uint64(int(seed) & (2**64 - 1)), ctypes.c_uint64(int(generation)), as_dp(x),
                               None if tries is None else tries.ctypes.data_as(ctypes.POINTER(ctypes.c_int32))))
    return (x, tries) if return_tries else x


def distort_map(h, w, coeffs):
    """float32 (map_x, map_y) of ``distort`` for an (h, w) image (alp_distort_map)."""
    cf = np.ascontiguousarray(coeffs, dtype=np.float64)
    if cf.shape != (14,):
        raise ValueError("distort_coeffs must have 14 entries")
    mx = result_empty((int(h), int(w)), np.float32)
    my = result_empty((int(h), int(w)), np.float32)
    check(lib().alp_distort_map(int(h), int(w), as_dp(cf), as_fp(mx), as_fp(my)))
    return mx, my


def synchronize():
    check(lib().alp_synchronize())


def kernel_timing(enable):
    check(lib().alp_kernel_timing(int(bool(enable))))


def kernel_time_ms():
    """(sum of the kernel sections since the last call in ms, their number); see alp_kernel_timing"""
    ms, n = ctypes.c_float(), _c_int()
    check(lib().alp_kernel_time_ms(ctypes.byref(ms), ctypes.byref(n)))
    return float(ms.value), int(n.value)


def build_flags():
    return (load().alp_build_flags() or b"").decode()


def event_record(slot):
    check(lib().alp_event_record(slot))


def event_elapsed_ms(a, b):
    ms = ctypes.c_float()
    check(lib().alp_event_elapsed_ms(a, b, ctypes.byref(ms)))
    return float(ms.value)


def comm_unique_id():
    buf = ctypes.create_string_buffer(UNIQUE_ID_BYTES)
    check(lib().alp_comm_unique_id(buf))
    return buf.raw


def comm_init(uid, rank, world_size):
    if len(uid) != UNIQUE_ID_BYTES:
        raise ValueError("unique id must be 128 bytes")
    check(lib().alp_comm_init(uid, int(rank), int(world_size)))


def comm_info():
    r, w = _c_int(), _c_int()
    check(lib().alp_comm_info(ctypes.byref(r), ctypes.byref(w)))
    return r.value, w.value


def comm_bcast(array, root=0):
    """In-place broadcast of a C-contiguous numpy array from ``root`` (no-op without a communicator)."""
    if not array.flags["C_CONTIGUOUS"]:
        raise ValueError("array must be C-contiguous")
    check(lib().alp_comm_bcast(array.ctypes.data_as(_c_void_p), array.nbytes, int(root)))
    return array


def comm_allgather(array):
    """Concatenation, in rank order, of every rank's 1-D / 2-D C-contiguous array along axis 0 (rows of equal width, row
    counts may differ); a copy without a communicator (alp_comm_allgather_counts + alp_comm_allgatherv)."""
    array = np.ascontiguousarray(array)
    _, world = comm_info()
    counts = (_c_i64 * world)()
    check(lib().alp_comm_allgather_counts(array.nbytes, counts))
    row = array.strides[0] if array.ndim > 1 else array.itemsize
    total = sum(counts)
    out = result_empty((total // row,) + array.shape[1:], array.dtype)
    check(lib().alp_comm_allgatherv(array.ctypes.data_as(_c_void_p), out.ctypes.data_as(_c_void_p), counts))
    return out


def comm_destroy():
    check(lib().alp_comm_destroy())
