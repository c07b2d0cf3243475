"""Drop-in counterpart of the reference module ``alproj.optimize`` (src/alproj/optimize.py)
whose per-point arithmetic runs on an MI355X through libalproj_hip.so.

Same names, argument meaning and return types as the reference:

============================  =======================================  ======================
here                          reference                                device entry point
============================  =======================================  ======================
``project``                   optimize.py:122-155                      alp_project
``rmse`` / ``huber_loss``     optimize.py:157-178 / :181-212           alp_loss_uv
``compute_residuals``         optimize.py:215-237                      alp_residuals
``bounds_to_array``           optimize.py:249-276                      (host, D <= 21 scalars)
``CMAOptimizer.optimize``     optimize.py:359-439                      alp_eval_population
``LsqOptimizer.optimize``     optimize.py:467-539                      alp_residuals
``intrinsic_mat`` etc.        optimize.py:8-96                         (host, 3x3 / 4x4)
============================  =======================================  ======================

Extra keyword arguments (all optional, defaults keep the reference behaviour):
``precision`` (None | "f32" | "f64") selects the element type of the device-resident point set
-- None, the default, is the reference's own float64 arithmetic (optimize.py:329-357) for every
point set of up to ``F64_MAX_POINTS`` = 4 M points (any GCP-scale use of the reference: float64
costs nothing there) and float32 (20 B/vertex, ~1e-3 px) for DSM-sized sets above it;
``seed`` makes the CMA-ES trajectory reproducible.  There is no CPU fallback: without the
HIP library / a GPU every function that touches points raises ``AlprojHipError``.
"""
from math import cos, pi, sin, tan

import numpy as np
import pandas as pd
from scipy.optimize import least_squares
from tqdm import tqdm

from . import _lib
from .cma import CMA

__all__ = ["intrinsic_mat", "extrinsic_mat", "project", "rmse", "huber_loss",
           "compute_residuals", "DEFAULT_BOUND_WIDTHS", "bounds_to_array", "BaseOptimizer",
           "CMAOptimizer", "LsqOptimizer"]


# ------------------------------------------------------------------------------------------
# host scalars (per pose, not per point)
# ------------------------------------------------------------------------------------------
def intrinsic_mat(fov_x_deg, w, h, cx=None, cy=None):
    """OpenCV-style intrinsic matrix (reference optimize.py:8-44; fov_y = fov_x * h / w)."""
    if cx is None:
        cx = w / 2
    if cy is None:
        cy = h / 2
    half_x = fov_x_deg * pi / 180 / 2
    half_y = (fov_x_deg * pi / 180) * h / w / 2
    return np.array([[w / (2 * tan(half_x)), 0, cx],
                     [0, h / (2 * tan(half_y)), cy],
                     [0, 0, 1]], dtype=np.float64)


def extrinsic_mat(pan_deg, tilt_deg, roll_deg, t_x, t_y, t_z):
    """4x4 extrinsic matrix (reference optimize.py:46-96): Rx(-(tilt+90)) Ry(-roll) Rz(pan)."""
    a, b, c = pan_deg * pi / 180, -(tilt_deg + 90) * pi / 180, -roll_deg * pi / 180
    rz = np.array([[cos(a), -sin(a), 0], [sin(a), cos(a), 0], [0, 0, 1.0]])
    rx = np.array([[1.0, 0, 0], [0, cos(b), -sin(b)], [0, sin(b), cos(b)]])
    ry = np.array([[cos(c), 0, sin(c)], [0, 1.0, 0], [-sin(c), 0, cos(c)]])
    rot = rx @ ry @ rz
    m = np.eye(4)
    m[:3, :3] = rot
    m[:3, 3] = rot @ np.array([-t_x, -t_y, -t_z], dtype=np.float64)
    return m


def _camera_origin(params):
    return np.array([float(params["x"]), float(params["y"]), float(params["z"])])


def _columns(frame, names):
    """The named columns of a DataFrame as the device upload takes them, WITHOUT the copies of the reference's
    `obj_points[["x", "y", "z"]]` + `np.array(...)` (optimize.py:139-141; 96 ms for 10 M rows on one core): columns that lie
    contiguous in the frame's blocks are handed out as they are (a list of 1-D views); a frame that IS a row-major (N, k)
    float64 array hands out that array; only strided or non-float64 columns are copied, one by one."""
    cols = [frame[c].to_numpy() for c in names]                   # views of the frame's blocks: nothing is copied yet
    if all(c.dtype == np.float64 and c.flags["C_CONTIGUOUS"] for c in cols):
        return cols                                                # columns x rows blocks (or one block per column): as they lie
    if list(frame.columns) == list(names) and all(dt == np.float64 for dt in frame.dtypes):
        a = frame.to_numpy(copy=False)                             # one block holding a row-major (N, k) array: as it lies
        if a.flags["C_CONTIGUOUS"]:
            return a
    return [np.ascontiguousarray(c, dtype=np.float64) for c in cols]


def _first_two_columns(frame):
    """columns 0 and 1 of a DataFrame, whatever their labels, without copying where they lie contiguous"""
    if frame.shape[1] < 2:
        raise IndexError("index 1 is out of bounds for axis 1 with size %d" % frame.shape[1])      # numpy's, as the reference raises it
    cols = [frame.iloc[:, k].to_numpy() for k in (0, 1)]
    if all(c.dtype == np.float64 and c.flags["C_CONTIGUOUS"] for c in cols):
        return cols
    return [np.ascontiguousarray(c, dtype=np.float64) for c in cols]


def _xyz_array(obj_points):
    if isinstance(obj_points, pd.DataFrame):
        return _columns(obj_points, ["x", "y", "z"])
    return np.asarray(obj_points, dtype=np.float64)


def _uv_array(img_points):
    if isinstance(img_points, pd.DataFrame):
        return _columns(img_points, ["u", "v"])
    return np.asarray(img_points, dtype=np.float64)


def _rows(a):
    return len(a[0]) if isinstance(a, list) else len(a)


def _points(xyz, origin, precision):
    """device point set from what _xyz_array returned: an (N, 3) array or a list of three columns"""
    if isinstance(xyz, list):
        return _lib.Points.from_columns(xyz[0], xyz[1], xyz[2], origin, precision)
    return _lib.Points(xyz, origin, precision)


def _set_observed(pts, uv):
    if isinstance(uv, list):
        pts.set_observed_columns(uv[0], uv[1])
    else:
        pts.set_observed(uv)


def _as_rows(a):
    """(N, k) row-major float64 for the few callers that need the array itself"""
    return np.ascontiguousarray(np.column_stack(a) if isinstance(a, list) else a, dtype=np.float64)


# ------------------------------------------------------------------------------------------
# projection and losses
# ------------------------------------------------------------------------------------------
F64_MAX_POINTS = 4_000_000


def default_precision(n_points, precision=None):
    """The element type of a device-resident point set when the caller names none: the reference's float64
    (optimize.py:139-154, 329-357) up to F64_MAX_POINTS points, float32 above (DSM-sized sets, where the float64
    population kernel costs 3 x the float32 one and the set 2 x the HBM)."""
    if precision is not None:
        if precision not in ("f32", "f64"):
            raise ValueError("precision must be None, 'f32' or 'f64'")
        return precision
    return "f64" if int(n_points) <= F64_MAX_POINTS else "f32"


def project(obj_points, params, precision="f64"):
    """3D -> 2D perspective projection of ``obj_points`` (DataFrame with x, y, z) with the
    camera ``params``; returns a DataFrame with columns u, v (reference optimize.py:122-155).

    The points are uploaded relative to the camera position, projected by one HIP kernel and
    fetched back.  ``precision="f64"`` (default) reproduces the float64 reference to ~1e-12;
    ``"f32"`` streams 20 B/vertex and is accurate to ~1e-3 px.
    """
    xyz = _xyz_array(obj_points)
    with _points(xyz, _camera_origin(params), precision) as pts:
        pts.project(_lib.params_vector(params))
        u, v = pts.fetch(np.float64)
    return pd.DataFrame({"u": u, "v": v}, copy=False)      # the two result arrays ARE the columns (no 16 N-byte copy)


def _uv_pointers(a):
    """(pointer to u or to the interleaved pairs, pointer to v or None, rows, what must stay alive) for alp_loss_uv_columns"""
    if isinstance(a, list):
        cols = [np.ascontiguousarray(c, dtype=np.float64) for c in a]
        return _lib.as_dp(cols[0]), _lib.as_dp(cols[1]), len(cols[0]), cols
    a = np.asarray(a, dtype=np.float64)
    if a.ndim != 2 or a.shape[1] != 2:
        raise ValueError("pixel coordinates must have shape (N, 2)")
    if not a.flags["C_CONTIGUOUS"] and a.shape[0] > 1 and a[:, 0].flags["C_CONTIGUOUS"] and a[:, 1].flags["C_CONTIGUOUS"]:
        cols = [a[:, 0], a[:, 1]]
        return _lib.as_dp(cols[0]), _lib.as_dp(cols[1]), a.shape[0], cols
    a = np.ascontiguousarray(a)
    return _lib.as_dp(a), None, a.shape[0], a


def _loss_uv(img_points, projected, kind, f_scale):
    """rmse / huber_loss of two tables of pixel coordinates: each goes to the device as it lies -- row-major pairs or two
    columns (what project() returns) -- through alp_loss_uv_columns; no host-side interleaving"""
    # the reference takes `projected` BY POSITION (projected.to_numpy()[:, 0], [:, 1]: optimize.py:175-176, 203-206) and only
    # img_points by name: an unlabelled frame works, a frame ordered [v, u] is read as it lies
    prj = _first_two_columns(projected) if isinstance(projected, pd.DataFrame) else projected
    ou, ov, n_obs, keep_o = _uv_pointers(_uv_array(img_points))
    pu, pv, n_prj, keep_p = _uv_pointers(prj)
    if n_obs != n_prj:
        raise ValueError("img_points and projected must have the same shape")
    out = _lib.ctypes.c_double()
    _lib.check(_lib.lib().alp_loss_uv_columns(ou, ov, pu, pv, n_obs, kind, float(f_scale), _lib.ctypes.byref(out)))
    del keep_o, keep_p
    return float(out.value)


def rmse(img_points, projected):
    """Mean Euclidean reprojection distance in pixels -- what the reference calls RMSE
    (optimize.py:157-178)."""
    return _loss_uv(img_points, projected, _lib.LOSS_MEAN_DIST, 0.0)


def huber_loss(img_points, projected, f_scale=10.0):
    """Mean Huber loss of the reprojection distance (reference optimize.py:181-212)."""
    return _loss_uv(img_points, projected, _lib.LOSS_HUBER, f_scale)


def compute_residuals(obj_points, img_points, params):
    """Flattened residual vector (observed - projected), reference optimize.py:215-237."""
    xyz = _xyz_array(obj_points)
    with _points(xyz, _camera_origin(params), "f64") as pts:
        _set_observed(pts, _uv_array(img_points))
        return pts.residuals(_lib.params_vector(params))


# ------------------------------------------------------------------------------------------
# optimisers
# ------------------------------------------------------------------------------------------
DEFAULT_BOUND_WIDTHS = {
    "fov": 45, "pan": 45, "tilt": 45, "roll": 45,
    "x": 30, "y": 30, "z": 30,
    "a1": 0.2, "a2": 0.2,
    "k1": 0.2, "k2": 0.2, "k3": 0.2, "k4": 0.2, "k5": 0.2, "k6": 0.2,
    "p1": 0.2, "p2": 0.2,
    "s1": 0.2, "s2": 0.2, "s3": 0.2, "s4": 0.2,
}


def bounds_to_array(params_init, target_params, bound_widths=None):
    """(D, 2) array [init - width, init + width] (reference optimize.py:249-276); keys missing
    from both ``bound_widths`` and DEFAULT_BOUND_WIDTHS get width 0.2."""
    widths = bound_widths or {}
    centre = np.array([params_init[k] for k in target_params], dtype=np.float64)
    half = np.array([widths.get(k, DEFAULT_BOUND_WIDTHS.get(k, 0.2)) for k in target_params],
                    dtype=np.float64)
    return np.column_stack([centre - half, centre + half]).reshape(len(target_params), 2)


class BaseOptimizer:
    """Holds GCP object/image points and the initial parameters (reference optimize.py:279-319)."""

    def __init__(self, obj_points, img_points, params_init):
        self.obj_points = obj_points
        self.img_points = img_points
        self.params_init = params_init

    def set_target(self, target_params=["fov", "pan", "tilt", "roll", "a1", "a2", "k1", "k2", "k3",
                                        "k4", "k5", "k6", "p1", "p2", "s1", "s2", "s3", "s4"]):
        """Choose the parameters to optimise (x, y, z may be added)."""
        self.target_params = target_params
        self.target_params_init = np.array([self.params_init[t] for t in target_params])

    # -- device-resident copy of the GCPs, shared by both optimisers -------------------------
    def _device_points(self, precision=None):
        xyz = _xyz_array(self.obj_points)
        pts = _points(xyz, _camera_origin(self.params_init), default_precision(_rows(xyz), precision))
        _set_observed(pts, _uv_array(self.img_points))
        return pts

    def _candidate_matrix(self, values):
        """(P, D) target values -> (P, 25) ABI parameter vectors (non-target keys from
        params_init), the vectorised form of reference optimize.py:341-350."""
        values = np.atleast_2d(np.asarray(values, dtype=np.float64))
        base = _lib.params_vector(self.params_init)
        cand = np.tile(base, (values.shape[0], 1))
        cols = [_lib.PARAM_KEYS.index(t) for t in self.target_params]
        cand[:, cols] = values
        return cand

    def _result_params(self, best_values):
        params = self.params_init.copy()
        for t in self.target_params:
            params.pop(t)
        params.update(dict(zip(self.target_params, best_values)))
        return params


class CMAOptimizer(BaseOptimizer):
    """CMA-ES optimiser of the camera parameters (reference optimize.py:322-439).

    One generation = one ``alp_eval_population`` call: all ``population_size`` candidates are
    projected against every point and reduced to their losses by a single kernel (plus, with
    several GPUs, one RCCL all-reduce of the per-candidate sums).
    """

    def _loss_function(self, bounds, f_scale=None, precision=None):
        """Population loss closure: X (P, D) in [0,1] -> (losses (P,), argmin).  Vectorised
        counterpart of the reference's per-candidate ``_proj_error`` (optimize.py:347-356)."""
        lower, upper = bounds[:, 0], bounds[:, 1]
        pts = self._device_points(precision)
        kind = _lib.LOSS_MEAN_DIST if f_scale is None else _lib.LOSS_HUBER
        fs = 0.0 if f_scale is None else float(f_scale)

        def _proj_error(normalized_values, want_argmin=True):
            x = np.atleast_2d(np.asarray(normalized_values, dtype=np.float64))
            cand = self._candidate_matrix(x * (upper - lower) + lower)
            return pts.eval_population(cand, kind, fs, want_argmin)

        _proj_error.points = pts
        return _proj_error

    def optimize(self, sigma=0.2, bound_widths=None, generation=1000, population_size=10,
                 n_max_resampling=100, f_scale=None, precision=None, seed=None, progress=True):
        """Run CMA-ES; returns ``(params, error)`` like the reference: the best candidate of
        the LAST generation (optimize.py:427, quirk Q9) and its mean reprojection distance.
        ``precision=None``: float64 like the reference up to F64_MAX_POINTS points (per rank), float32 above."""
        bounds = bounds_to_array(self.params_init, self.target_params, bound_widths)
        lower, upper = bounds[:, 0], bounds[:, 1]
        normalized_init = (self.target_params_init - lower) / (upper - lower)
        d = len(self.target_params)
        normalized_bounds = np.column_stack([np.zeros(d), np.ones(d)])

        # Several ranks (one process per GPU, vertices sharded): every population evaluation
        # all-reduces the per-candidate sums, so every rank MUST evaluate the same candidates.
        # The reference seeds nothing (optimize.py:410-416); here rank 0's seed (its own entropy
        # when seed is None) and, every generation, rank 0's candidate matrix are broadcast.
        # Rank 0's element type travels with the seed: the float64 confirmation of near-tied float32
        # losses is a collective of its own, so shards on either side of F64_MAX_POINTS must not differ.
        precision = default_precision(len(self.obj_points), precision)
        _, world = _lib.comm_info()
        if world > 1:
            s = np.array([np.random.SeedSequence().entropy % (1 << 63) if seed is None else int(seed),
                          precision == "f64"], dtype=np.uint64)
            _lib.comm_bcast(s, root=0)
            seed, precision = int(s[0]), ("f64" if s[1] else "f32")
        loss_function = self._loss_function(bounds, f_scale, precision)
        pts = loss_function.points
        try:
            optimizer = CMA(mean=normalized_init.astype("float64"), sigma=float(sigma),
                            bounds=normalized_bounds, population_size=population_size,
                            n_max_resampling=n_max_resampling, seed=seed,
                            # the device sampler costs a launch + a copy (~0.07 ms): it pays from a few thousand
                            # deviates per generation (pop 256 / D 21: 0.10 ms against 7 ms of numpy at sigma = 1);
                            # at GCP scale (pop 50 / D 9) the numpy path is the faster one (0.17 vs 0.24 ms / generation)
                            sampler=_lib.cma_sample if (d <= 32 and population_size * d >= 2048) else None)
            it = range(generation)
            best_normalized = normalized_init
            for g in (tqdm(it) if progress else it):
                X = np.ascontiguousarray(optimizer.ask_population())
                if world > 1:
                    _lib.comm_bcast(X, root=0)
                # the result is the best candidate of the LAST generation (optimize.py:427, quirk Q9): only there is the
                # argmin itself needed -- and confirmed in float64 among near-tied candidates of a float32 point set
                losses, amin = loss_function(X, g == generation - 1)
                best_normalized = X[amin].copy()
                optimizer.tell_population(X, losses)
            best_values = best_normalized * (upper - lower) + lower
            params = self._result_params(best_values)
            # final error is always the mean distance (optimize.py:435-437), and float64 like the
            # reference's (and like LsqOptimizer's): a float32 point set of GCP size is evaluated once
            # more from a float64 copy; DSM-sized sets keep their float32 residency (a collective when
            # several ranks hold shards: every rank must reach this line)
            final = self._candidate_matrix(best_values)
            if pts.precision == _lib.ALP_F32 and pts.n <= self.F64_FINAL_MAX_POINTS:
                with self._device_points("f64") as p64:
                    err, _ = p64.eval_population(final, _lib.LOSS_MEAN_DIST, 0.0)
            else:
                err, _ = pts.eval_population(final, _lib.LOSS_MEAN_DIST, 0.0)
        finally:
            pts.close()
        return params, float(err[0])

    F64_FINAL_MAX_POINTS = F64_MAX_POINTS


class LsqOptimizer(BaseOptimizer):
    """scipy.optimize.least_squares driver (reference optimize.py:442-539); the residual
    vector of every trial point comes from ``alp_residuals`` on a float64 point set.

    Several ranks (points sharded, one process per GPU): the reference solves ONE problem over all points
    (optimize.py:510-528), so every rank all-gathers the residual vector -- and the rows of the batched Jacobian -- of
    all shards, in rank order (``alp_comm_allgatherv``): contiguous shards concatenate to the single-process vector,
    every rank runs the identical scipy solve on it and gets the identical optimum."""

    def _residual_function(self):
        pts = self._device_points("f64")
        _, world = _lib.comm_info()

        def _residuals(values):
            r = pts.residuals(self._candidate_matrix(values)[0])
            return _lib.comm_allgather(r) if world > 1 else r

        _residuals.points = pts
        return _residuals

    def _jacobian_function(self, pts, bounds=None):
        """2-point finite-difference Jacobian with scipy's own step rule (relative step
        sqrt(eps), sign-aware, flipped or shrunk at the bounds), but all D+1 residual vectors
        come from ONE ``alp_residuals_batch`` launch instead of D+1 sequential calls."""
        eps = np.finfo(np.float64).eps ** 0.5
        _, world = _lib.comm_info()
        lb = np.full(len(self.target_params), -np.inf) if bounds is None else np.asarray(bounds[0], dtype=np.float64)
        ub = np.full(len(self.target_params), np.inf) if bounds is None else np.asarray(bounds[1], dtype=np.float64)

        def _jac(values, *args, **kw):
            x0 = np.asarray(values, dtype=np.float64)
            h = eps * np.where(x0 >= 0, 1.0, -1.0) * np.maximum(1.0, np.abs(x0))
            lower_dist, upper_dist = x0 - lb, ub - x0
            x = x0 + h
            violated = (x < lb) | (x > ub)
            fitting = np.abs(h) <= np.maximum(lower_dist, upper_dist)
            h = np.where(violated & fitting, -h, h)
            h = np.where((upper_dist >= lower_dist) & ~fitting, upper_dist, h)
            h = np.where((upper_dist < lower_dist) & ~fitting, -lower_dist, h)
            d = len(x0)
            trial = np.tile(x0, (d + 1, 1))
            trial[np.arange(1, d + 1), np.arange(d)] += h
            dx = trial[np.arange(1, d + 1), np.arange(d)] - x0          # the representable step
            res = pts.residuals_batch(self._candidate_matrix(trial))
            jac = ((res[1:] - res[0]) / dx[:, None]).T               # (2 n_local, D): this rank's rows
            return _lib.comm_allgather(np.ascontiguousarray(jac)) if world > 1 else jac

        return _jac

    def optimize(self, method="trf", bound_widths=None, loss="linear", f_scale=1.0, **kwargs):
        if method == "lm" and bound_widths is not None:
            raise ValueError("method='lm' does not support bounds. Set bound_widths=None or use 'trf'/'dogbox'.")
        if method == "lm" and loss != "linear":
            raise ValueError("method='lm' does not support robust loss functions. Use loss='linear' or method='trf'/'dogbox'.")

        residual_func = self._residual_function()
        pts = residual_func.points
        try:
            if method == "lm":
                # MINPACK's own forward differences unless the caller asks for the batched ones
                if kwargs.get("jac") == "batched":
                    kwargs = dict(kwargs, jac=self._jacobian_function(pts))
                result = least_squares(residual_func, self.target_params_init, method=method, **kwargs)
            else:
                bounds = bounds_to_array(self.params_init, self.target_params, bound_widths)
                if kwargs.get("jac", "batched") == "batched":
                    kwargs = dict(kwargs, jac=self._jacobian_function(pts, (bounds[:, 0], bounds[:, 1])))
                result = least_squares(residual_func, self.target_params_init, method=method,
                                       bounds=(bounds[:, 0], bounds[:, 1]), loss=loss,
                                       f_scale=f_scale, **kwargs)
            # With a communicator every rank has solved the SAME gathered problem; the final error below is a collective
            # over the shards and must be evaluated for one solution on all of them: rank 0's is broadcast (bit-identical
            # to the others' on identical hosts; the broadcast makes it so on any).
            best = np.ascontiguousarray(result.x, dtype=np.float64)
            _, world = _lib.comm_info()
            if world > 1:
                _lib.comm_bcast(best, root=0)
            params = self._result_params(best)
            err, _ = pts.eval_population(self._candidate_matrix(best), _lib.LOSS_MEAN_DIST, 0.0)
        finally:
            pts.close()
        return params, float(err[0])
