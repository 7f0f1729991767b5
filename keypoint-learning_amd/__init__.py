"""keypoint-learning_amd -- MI355X-native engine for the scoring path of
pcl::keypoints::KeypointLearningDetector (feature -> random forest -> radius NMS).

The product is the C-ABI shared library `libkpl.so` (include/kpl.h; HIP kernels in csrc/).
This module is a thin ctypes binding of that ABI plus `KeypointLearningDetector`, a Python
mirror of the reference class (/root/reference/include/KeypointLearning.h:55-206) used by the
tests and the bench.  It never computes anything itself and has no CPU fallback: if the HIP
library is missing or no GPU is usable, it raises.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("KPL_LIB_PATH") or os.path.join(_HERE, "libkpl.so")

OK, ERR_INVALID_ARG, ERR_NO_FOREST, ERR_FOREST_PARSE, ERR_VAR_COUNT, ERR_GRID_TOO_LARGE, \
    ERR_CAPACITY, ERR_DEVICE, ERR_UNSUPPORTED, ERR_IO, ERR_NO_CLOUD, ERR_RETRY, ERR_INTERNAL = range(13)


WALK_AUTO, WALK_LANES, WALK_TWO_PASS = -1, 0, 1
NEIGHBORS_CANONICAL, NEIGHBORS_SORTED = 0, 1      # kpl_params.neighbor_order


class KplError(RuntimeError):
    def __init__(self, status, message):
        super().__init__("kpl status %d: %s" % (status, message))
        self.status = status


class Params(C.Structure):
    _fields_ = [("n_annulus", C.c_int), ("n_bins", C.c_int), ("radius_search", C.c_double),
                ("non_max_radius", C.c_double), ("prediction_th", C.c_double),
                ("non_maxima", C.c_int), ("non_maxima_draws_remove", C.c_int),
                ("non_maxima_draws_threshold", C.c_float), ("neighbor_order", C.c_int)]


class ForestSummary(C.Structure):
    _fields_ = [("ntrees", C.c_int), ("var_count", C.c_int), ("nnodes", C.c_int64),
                ("max_depth", C.c_int)]


class Timing(C.Structure):
    _fields_ = [("calls", C.c_int), ("index_ms", C.c_float), ("score_ms", C.c_float),
                ("nms_ms", C.c_float), ("feature_ms", C.c_float), ("forest_ms", C.c_float),
                ("walk", C.c_int), ("lanes_per_point", C.c_int), ("accept_words", C.c_int)]


class LaunchInfo(C.Structure):
    _fields_ = [("walk", C.c_int), ("lanes_per_point", C.c_int), ("accept_words", C.c_int),
                ("sorted_list_keys", C.c_int), ("sorted_all_large", C.c_int)]


class Stats(C.Structure):
    _fields_ = [(k, C.c_int64) for k in ("n_points", "n_scored", "n_thresholded", "sum_kf",
                                         "sum_kn", "sum_depth", "n_keypoints", "n_cells")]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


# every symbol include/kpl.h declares: (name, restype, argtypes)
_vp, _ip, _fp = C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_float)
SYMBOLS = {
    "kpl_version": (C.c_int, []),
    "kpl_source_hash": (C.c_char_p, []),
    "kpl_status_string": (C.c_char_p, [C.c_int]),
    "kpl_default_params": (None, [C.POINTER(Params)]),
    "kpl_create": (C.c_int, [C.POINTER(_vp), C.c_int]),
    "kpl_destroy": (None, [_vp]),
    "kpl_last_error": (C.c_char_p, [_vp]),
    "kpl_set_params": (C.c_int, [_vp, C.POINTER(Params)]),
    "kpl_get_params": (C.c_int, [_vp, C.POINTER(Params)]),
    "kpl_load_forest_file": (C.c_int, [_vp, C.c_char_p]),
    "kpl_load_forest_memory": (C.c_int, [_vp, _vp, C.c_size_t]),
    "kpl_load_forest_arrays": (C.c_int, [_vp, C.c_int, C.c_int, C.c_int, _ip, _ip, _fp, _ip, _ip,
                                         C.POINTER(C.c_double)]),
    "kpl_forest_info": (C.c_int, [_vp, _ip, _ip, C.POINTER(C.c_int64), _ip]),
    "kpl_forest_inspect": (C.c_int, [_vp, C.c_size_t, C.POINTER(ForestSummary), C.c_char_p, C.c_size_t]),
    "kpl_forest_export_arrays": (C.c_int, [_vp, C.c_size_t, C.c_int64, C.c_int, _vp, _vp, _vp, _vp, _vp,
                                           _vp, C.c_char_p, C.c_size_t]),
    "kpl_detect": (C.c_int, [_vp, _vp, C.c_size_t, _vp, C.c_size_t, C.c_int, _vp, _vp, C.c_int, _ip]),
    "kpl_detect_keypoints": (C.c_int, [_vp, _vp, C.c_size_t, _vp, C.c_size_t, C.c_int, _vp, _vp, C.c_int, _ip]),
    "kpl_host_staging": (C.c_int, [_vp, C.c_int, C.c_size_t, C.c_size_t, C.POINTER(_vp), C.POINTER(_vp)]),
    "kpl_detect_keypoints_staged": (C.c_int, [_vp, _vp, _vp, C.c_int, _ip]),
    "kpl_compute_features": (C.c_int, [_vp, _vp, C.c_size_t, _vp, C.c_size_t, C.c_int, _vp,
                                       C.c_int, _vp]),
    "kpl_bind_cloud_device": (C.c_int, [_vp, _vp, C.c_size_t, _vp, C.c_size_t, C.c_int]),
    "kpl_build_index_device": (C.c_int, [_vp, _vp]),
    "kpl_detect_device": (C.c_int, [_vp, _vp, _vp, C.c_int, _vp, _vp]),
    "kpl_compute_device": (C.c_int, [_vp, _vp, _vp, C.c_int, _vp, _vp]),
    "kpl_compute_features_device": (C.c_int, [_vp, _vp, C.c_int, _vp, _vp]),
    "kpl_compute_batch_device": (C.c_int, [_vp, C.c_int, _vp, _vp, _vp, _vp, _vp]),
    "kpl_compute_batch_keypoints_device": (C.c_int, [_vp, C.c_int, _vp, _vp, _vp, _vp, _vp]),
    "kpl_sync_status": (C.c_int, [_vp, _vp]),
    "kpl_reserve": (C.c_int, [_vp, C.c_int, C.c_size_t, C.c_size_t]),
    "kpl_get_last_launch": (C.c_int, [_vp, C.POINTER(LaunchInfo)]),
    "kpl_compute_features_batch_device": (C.c_int, [_vp, C.c_int, _vp, _vp, _vp, _vp]),
    "kpl_set_feature_walk": (C.c_int, [_vp, C.c_int, C.c_int]),
    "kpl_get_feature_walk": (C.c_int, [_vp, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_double)]),
    "kpl_enable_timing": (C.c_int, [_vp, C.c_int]),
    "kpl_get_timing": (C.c_int, [_vp, C.POINTER(Timing)]),
    "kpl_collect_stats": (C.c_int, [_vp, C.POINTER(Stats), _vp]),
    "kpl_cloud_resolution": (C.c_int, [_vp, _vp, C.c_size_t, C.c_int, C.POINTER(C.c_double)]),
    "kpl_set_grid_origin": (C.c_int, [_vp, _vp]),
    "kpl_estimate_normals": (C.c_int, [_vp, _vp, C.c_size_t, C.c_int, C.c_int, C.c_double, _vp, _vp, C.c_size_t,
                                       _vp, C.c_size_t]),
    "kpl_estimate_normals_device": (C.c_int, [_vp, C.c_int, C.c_double, _vp, _vp, C.c_size_t, _vp, C.c_size_t, _vp]),
    "kpl_estimate_normals_organized": (C.c_int, [_vp, _vp, C.c_size_t, C.c_int, C.c_int, C.c_float, _vp, _vp, C.c_size_t,
                                                 _vp, C.c_size_t]),
    "kpl_estimate_normals_organized_device": (C.c_int, [_vp, _vp, C.c_size_t, C.c_int, C.c_int, C.c_float, _vp, _vp,
                                                        C.c_size_t, _vp, C.c_size_t, _vp]),
}

_lib = None


def load_library():
    """dlopen libkpl.so and bind every symbol of include/kpl.h.  Raises if it is not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError("libkpl.so is not built: run `python keypoint-learning_amd/build.py` "
                          "(there is no CPU fallback)")
    # A process that also uses PyTorch must load torch's bundled HIP runtime BEFORE libkpl pulls in
    # /opt/rocm's: with the other order torch later reports "No HIP GPUs are available".  Callers of
    # this binding (tests, bench) hand torch-owned device buffers to libkpl, so torch goes first.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(lib, name)          # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def compute_batch_device(detectors, d_scores, d_kp_idx, kp_caps, d_kp_counts, stream=None):
    """kpl_compute_batch_device: one scoring launch for up to 8 independent views.  The lists hold one
    entry per detector: raw device addresses (ints; d_scores entries may be None) and capacities."""
    lib = load_library()
    k = len(detectors)
    for d in detectors:
        d._push()
    H = (_vp * k)(*[d._h for d in detectors])
    S = (_vp * k)(*[(_vp(p) if p else _vp()) for p in d_scores]) if d_scores else None
    K = (_vp * k)(*[_vp(p) for p in d_kp_idx])
    N = (_vp * k)(*[_vp(p) for p in d_kp_counts])
    Cp = (C.c_int * k)(*kp_caps)
    rc = lib.kpl_compute_batch_device(C.cast(H, _vp), k, C.cast(S, _vp) if S else None, C.cast(K, _vp),
                                      C.cast(Cp, _vp), C.cast(N, _vp), stream)
    if rc != OK:
        raise KplError(rc, lib.kpl_last_error(detectors[0]._h).decode())


def compute_features_batch_device(detectors, d_indices, ms, d_features, stream=None):
    """kpl_compute_features_batch_device: computePointsForTrainingFeatures of up to 8 bound views in one launch.  One entry
    per detector: device address of its int32 point indices, their number, device address of its m x F output."""
    lib = load_library()
    k = len(detectors)
    for d in detectors:
        d._push()
    H = (_vp * k)(*[d._h for d in detectors])
    I = (_vp * k)(*[(_vp(p) if p else _vp()) for p in d_indices])
    M = (C.c_int * k)(*[int(m) for m in ms])
    O = (_vp * k)(*[(_vp(p) if p else _vp()) for p in d_features])
    rc = lib.kpl_compute_features_batch_device(C.cast(H, _vp), k, C.cast(I, _vp), C.cast(M, _vp), C.cast(O, _vp), stream)
    if rc != OK:
        raise KplError(rc, lib.kpl_last_error(detectors[0]._h).decode())


def forest_inspect(data):
    """Host-only: parse YAML / YAML.gz bytes with libkpl's reader.  Returns a dict or raises."""
    lib = load_library()
    buf = C.create_string_buffer(bytes(data), len(data))
    out, err = ForestSummary(), C.create_string_buffer(512)
    rc = lib.kpl_forest_inspect(C.cast(buf, _vp), len(data), C.byref(out), err, 512)
    if rc != OK:
        raise KplError(rc, err.value.decode())
    return {"ntrees": out.ntrees, "var_count": out.var_count, "nnodes": out.nnodes,
            "max_depth": out.max_depth}


def forest_export_arrays(data):
    """Host-only: node arrays (root, var, thr, left, right, value) as libkpl's reader sees them."""
    lib = load_library()
    info = forest_inspect(data)
    nn, nt = info["nnodes"], info["ntrees"]
    buf = C.create_string_buffer(bytes(data), len(data))
    root, var, left, right = (np.empty(k, dtype=np.int32) for k in (nt, nn, nn, nn))
    thr, value = np.empty(nn, dtype=np.float32), np.empty(nn, dtype=np.float64)
    err = C.create_string_buffer(512)
    rc = lib.kpl_forest_export_arrays(C.cast(buf, _vp), len(data), nn, nt, root.ctypes.data,
                                      var.ctypes.data, thr.ctypes.data, left.ctypes.data,
                                      right.ctypes.data, value.ctypes.data, err, 512)
    if rc != OK:
        raise KplError(rc, err.value.decode())
    return {"root": root, "var": var, "thr": thr, "left": left, "right": right, "value": value,
            "var_count": info["var_count"]}


class KeypointLearningDetector:
    """Mirror of pcl::keypoints::KeypointLearningDetector over the C-ABI.

    Same constructor defaults and setter names as the reference class
    (/root/reference/include/KeypointLearning.h:81-155); clouds are numpy arrays [n,3] (or
    [n,k>=3] rows whose first three floats are xyz / normal, e.g. a PointXYZ array viewed as
    [n,4]).
    """

    def __init__(self, prediction_th=0.5, non_maxima=True, non_maxima_draws_remove=True,
                 non_max_radius=0.0, n_annulus=5, n_bins=10, device=0):
        self._lib = load_library()
        h = _vp()
        rc = self._lib.kpl_create(C.byref(h), device)
        if rc != OK:
            raise KplError(rc, "kpl_create failed (is a HIP device visible?): "
                           + self._lib.kpl_status_string(rc).decode())
        self._h = h
        self._p = Params()
        self._lib.kpl_default_params(C.byref(self._p))
        self._p.prediction_th = prediction_th
        self._p.non_maxima = int(non_maxima)
        self._p.non_maxima_draws_remove = int(non_maxima_draws_remove)
        self._p.non_max_radius = non_max_radius
        self._p.n_annulus = n_annulus
        self._p.n_bins = n_bins
        self._cloud = self._normals = None
        self._keep = []
        self.keypoints_indices = np.zeros(0, dtype=np.int32)

    def close(self):
        if getattr(self, "_h", None):
            self._lib.kpl_destroy(self._h)
            self._h = None

    __del__ = close

    # -- error plumbing ----------------------------------------------------------------------
    def _check(self, rc):
        if rc != OK:
            raise KplError(rc, self._lib.kpl_last_error(self._h).decode())

    def _push(self):
        self._check(self._lib.kpl_set_params(self._h, C.byref(self._p)))

    # -- the reference's setters -------------------------------------------------------------
    def setInputCloud(self, cloud):
        cloud = _f32(cloud)
        if self._normals is not None and self._cloud is not None and cloud is not self._cloud:
            self._normals = None            # impl/KeypointLearning.hpp:52-55
        self._cloud = cloud

    def setNormals(self, normals):
        self._normals = _f32(normals)

    def setNonMaxima(self, v):
        self._p.non_maxima = int(v)

    def setNonMaximaDrawsRemove(self, v):
        self._p.non_maxima_draws_remove = int(v)

    def setNonMaximaDrawsThreshold(self, v):
        self._p.non_maxima_draws_threshold = v

    def setPredictionThreshold(self, th):
        self._p.prediction_th = th

    def setNonMaxRadius(self, r):
        self._p.non_max_radius = r

    def setNAnnulus(self, n):
        self._p.n_annulus = n

    def setNBins(self, n):
        self._p.n_bins = n

    def setRadiusSearch(self, r):
        self._p.radius_search = r

    def setSortedSearch(self, on=True):
        """pcl::Keypoint::setSearchMethod with a pcl::search::KdTree(sorted = on): the feature loop meets the
        neighbors in ascending (squared distance, index) order instead of the engine's canonical order."""
        self._p.neighbor_order = NEIGHBORS_SORTED if on else NEIGHBORS_CANONICAL

    def setFeatureWalk(self, walk=WALK_AUTO, lanes_per_point=2):
        """kpl_set_feature_walk: how the feature kernels walk the canonical order (WALK_AUTO / WALK_LANES / WALK_TWO_PASS, 2 or
        4 lanes per point) -- a choice of speed only, the results are the same bits."""
        self._check(self._lib.kpl_set_feature_walk(self._h, int(walk), int(lanes_per_point)))

    def getFeatureWalk(self):
        """(walk, lanes per point, mean neighbors per point measured by the handle's earlier calls or -1) of the next launch."""
        self._push()
        w, l, k = C.c_int(), C.c_int(), C.c_double()
        self._check(self._lib.kpl_get_feature_walk(self._h, C.byref(w), C.byref(l), C.byref(k)))
        return w.value, l.value, k.value

    def loadForest(self, path):
        rc = self._lib.kpl_load_forest_file(self._h, os.fsencode(path))
        return rc == OK

    def loadForestMemory(self, data):
        buf = (C.c_char * len(data)).from_buffer_copy(data)
        self._check(self._lib.kpl_load_forest_memory(self._h, C.cast(buf, _vp), len(data)))

    def loadForestArrays(self, root, var, thr, left, right, value, var_count):
        root, var, left, right = (np.ascontiguousarray(a, dtype=np.int32) for a in (root, var, left, right))
        thr = _f32(thr)
        value = np.ascontiguousarray(value, dtype=np.float64)
        self._check(self._lib.kpl_load_forest_arrays(
            self._h, len(root), len(var), int(var_count),
            root.ctypes.data_as(_ip), var.ctypes.data_as(_ip), thr.ctypes.data_as(_fp),
            left.ctypes.data_as(_ip), right.ctypes.data_as(_ip),
            value.ctypes.data_as(C.POINTER(C.c_double))))

    def forestInfo(self):
        nt, vc, md, nn = C.c_int(), C.c_int(), C.c_int(), C.c_int64()
        self._check(self._lib.kpl_forest_info(self._h, C.byref(nt), C.byref(vc), C.byref(nn), C.byref(md)))
        return {"ntrees": nt.value, "var_count": vc.value, "nnodes": nn.value, "max_depth": md.value}

    def lastError(self):
        return self._lib.kpl_last_error(self._h).decode()

    # -- compute -----------------------------------------------------------------------------
    @staticmethod
    def _rows(a):
        a = _f32(a)
        if a.ndim != 2 or a.shape[1] < 3:
            raise ValueError("expected an [n, >=3] float32 array")
        return a, a.shape[1] * 4

    def compute(self, with_scores=True):
        """pcl::Keypoint::compute.  Returns (keypoints [k,4] = x,y,z,score, scores [n] or None: with_scores=False
        is the drop-in class's own path, kpl_detect_keypoints, which brings back the keypoints only);
        the indices are left in `keypoints_indices` like getKeypointsIndices()."""
        if self._cloud is None:
            raise KplError(ERR_NO_CLOUD, "no input cloud")
        if self._normals is None:
            raise KplError(ERR_UNSUPPORTED, "normals must be given (setNormals): normal estimation "
                           "inside the detector is outside the accelerated path")
        xyz, xs = self._rows(self._cloud)
        nrm, ns = self._rows(self._normals)
        n = xyz.shape[0]
        if nrm.shape[0] != n:
            raise KplError(ERR_INVALID_ARG, "the number of normals does not match the number of "
                           "input points")   # impl/KeypointLearning.hpp:149-153
        self._push()
        kp = np.empty(max(n, 1), dtype=np.int32)
        cnt = C.c_int()
        if with_scores:
            scores = np.empty(max(n, 1), dtype=np.float32)
            rc = self._lib.kpl_detect(self._h, xyz.ctypes.data, xs, nrm.ctypes.data, ns, n, scores.ctypes.data,
                                      kp.ctypes.data, n, C.byref(cnt))
            self._check(rc)
            self.keypoints_indices = kp[:cnt.value].copy()
            scores = scores[:n]
            out = np.concatenate([xyz[self.keypoints_indices, :3],
                                  scores[self.keypoints_indices, None]], axis=1)
            return out, scores
        # what detectKeypoints() leaves behind and nothing else: indices + the response of the keypoints
        kps = np.empty(max(n, 1), dtype=np.float32)
        rc = self._lib.kpl_detect_keypoints(self._h, xyz.ctypes.data, xs, nrm.ctypes.data, ns, n, kp.ctypes.data,
                                            kps.ctypes.data, n, C.byref(cnt))
        self._check(rc)
        self.keypoints_indices = kp[:cnt.value].copy()
        return np.concatenate([xyz[self.keypoints_indices, :3], kps[:cnt.value, None]], axis=1), None

    def hostStaging(self, n, xyz_stride=12, normals_stride=12):
        """kpl_host_staging: two pinned host buffers of the handle as writable numpy views ([n, stride / 4] float32);
        fill them (first three floats of a row = xyz resp. normal) and call computeStaged()."""
        px, pn = _vp(), _vp()
        self._check(self._lib.kpl_host_staging(self._h, int(n), xyz_stride, normals_stride, C.byref(px), C.byref(pn)))
        mk = lambda p, stride: np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_float)), shape=(max(n, 1), stride // 4))[:n]
        self._stage_cap = max(int(n), 1)
        return mk(px, xyz_stride), mk(pn, normals_stride)

    def computeStaged(self):
        """kpl_detect_keypoints_staged: compute() on the staged view.  Returns (indices, responses of the keypoints)."""
        self._push()
        cnt = C.c_int()
        kp = np.empty(self._kp_cap_hint(), dtype=np.int32)
        kps = np.empty(len(kp), dtype=np.float32)
        self._check(self._lib.kpl_detect_keypoints_staged(self._h, kp.ctypes.data, kps.ctypes.data, len(kp), C.byref(cnt)))
        self.keypoints_indices = kp[:cnt.value].copy()
        return self.keypoints_indices, kps[:cnt.value].copy()

    def _kp_cap_hint(self):
        return max(getattr(self, "_stage_cap", 1), 1)

    def cloudResolution(self, cloud):
        """kpl::computeCloudResolution of the reference: mean distance to the second nearest neighbor."""
        xyz, xs = self._rows(cloud)
        res = C.c_double()
        self._check(self._lib.kpl_cloud_resolution(self._h, xyz.ctypes.data, xs, xyz.shape[0], C.byref(res)))
        return res.value

    def setGridOrigin(self, origin):
        """kpl_set_grid_origin: the whole cloud's origin for a view that is a slab of it (None = automatic)."""
        if origin is None:
            self._check(self._lib.kpl_set_grid_origin(self._h, None))
        else:
            o = np.ascontiguousarray(origin, dtype=np.float32)
            self._check(self._lib.kpl_set_grid_origin(self._h, o.ctypes.data))

    def estimateNormals(self, cloud, k=10, radius=0.0, viewpoint=(0.0, 0.0, 0.0)):
        """pcl::NormalEstimation as TestDetector drives it (k-search 10) or, with k=0, as the detector's own
        fallback does (radius search).  Returns (normals[n,3], curvature[n])."""
        xyz, xs = self._rows(cloud)
        n = xyz.shape[0]
        vp = np.ascontiguousarray(viewpoint, dtype=np.float32)
        out = np.empty((max(n, 1), 4), dtype=np.float32)
        self._check(self._lib.kpl_estimate_normals(self._h, xyz.ctypes.data, xs, n, int(k), float(radius), vp.ctypes.data,
                                                   out.ctypes.data, 16, out.ctypes.data + 12, 16))
        return out[:n, :3].copy(), out[:n, 3].copy()

    def estimateNormalsOrganized(self, cloud, width, height, smoothing_size=5.0, viewpoint=(0.0, 0.0, 0.0)):
        """pcl::IntegralImageNormalEstimation (SIMPLE_3D_GRADIENT) as the detector's fallback drives it on an
        organized cloud (hpp:138-145).  Returns (normals[n,3], curvature[n]); NaN where PCL leaves NaN."""
        xyz, xs = self._rows(cloud)
        n = int(width) * int(height)
        assert xyz.shape[0] == n
        vp = np.ascontiguousarray(viewpoint, dtype=np.float32)
        out = np.empty((max(n, 1), 4), dtype=np.float32)
        self._check(self._lib.kpl_estimate_normals_organized(self._h, xyz.ctypes.data, xs, int(width), int(height),
                                                             float(smoothing_size), vp.ctypes.data, out.ctypes.data, 16,
                                                             out.ctypes.data + 12, 16))
        return out[:n, :3].copy(), out[:n, 3].copy()

    def estimateNormalsOrganizedDevice(self, d_xyz, xyz_stride, width, height, smoothing_size, viewpoint, d_normals,
                                       normals_stride, d_curvature=None, curvature_stride=0, stream=None):
        """kpl_estimate_normals_organized_device: device pointers in and out, asynchronous on `stream`."""
        vp = np.ascontiguousarray(viewpoint, dtype=np.float32)
        self._check(self._lib.kpl_estimate_normals_organized_device(self._h, d_xyz, xyz_stride, int(width), int(height),
                                                                    float(smoothing_size), vp.ctypes.data, d_normals,
                                                                    normals_stride, d_curvature, curvature_stride, stream))

    def estimateNormalsDevice(self, k, radius, viewpoint, d_normals, normals_stride, d_curvature=None,
                              curvature_stride=0, stream=None):
        vp = np.ascontiguousarray(viewpoint, dtype=np.float32)
        self._check(self._lib.kpl_estimate_normals_device(self._h, int(k), float(radius), vp.ctypes.data, d_normals,
                                                          normals_stride, d_curvature, curvature_stride, stream))

    def getKeypointsIndices(self):
        return self.keypoints_indices

    def computePointsForTrainingFeatures(self, indices):
        xyz, xs = self._rows(self._cloud)
        nrm, ns = self._rows(self._normals)
        idx = np.ascontiguousarray(indices, dtype=np.int32)
        self._push()
        F = self._p.n_annulus * self._p.n_bins
        out = np.empty((len(idx), F), dtype=np.float32)
        self._check(self._lib.kpl_compute_features(self._h, xyz.ctypes.data, xs, nrm.ctypes.data, ns,
                                                   xyz.shape[0], idx.ctypes.data, len(idx),
                                                   out.ctypes.data))
        return out

    # -- device-resident path (pointers are raw device addresses, e.g. torch .data_ptr()) -----
    def bindCloudDevice(self, d_xyz, xyz_stride, d_nrm, nrm_stride, n):
        self._check(self._lib.kpl_bind_cloud_device(self._h, d_xyz, xyz_stride, d_nrm, nrm_stride, n))

    def buildIndexDevice(self, stream=None):
        self._push()
        self._check(self._lib.kpl_build_index_device(self._h, stream))

    def detectDevice(self, d_scores, d_kp_idx, kp_cap, d_kp_count, stream=None):
        self._push()
        self._check(self._lib.kpl_detect_device(self._h, d_scores, d_kp_idx, kp_cap, d_kp_count, stream))

    def computeDevice(self, d_scores, d_kp_idx, kp_cap, d_kp_count, stream=None):
        self._push()
        self._check(self._lib.kpl_compute_device(self._h, d_scores, d_kp_idx, kp_cap, d_kp_count, stream))

    def syncStatus(self, stream=None):
        """Waits for `stream`; returns OK or ERR_RETRY (cell tables grown: enqueue again), raises otherwise."""
        rc = self._lib.kpl_sync_status(self._h, stream)
        if rc not in (OK, ERR_RETRY):
            self._check(rc)
        return rc

    def getLastLaunch(self):
        """what the last scoring launch took (no wait, nothing cleared): walk, lanes per point, accept words, sorted-mode list
        capacity and whether every point went straight to the collect / add kernels"""
        li = LaunchInfo()
        self._check(self._lib.kpl_get_last_launch(self._h, C.byref(li)))
        return {k: getattr(li, k) for k, _ in li._fields_}

    def reserve(self, n_points, xyz_stride=12, normals_stride=12):
        self._check(self._lib.kpl_reserve(self._h, int(n_points), xyz_stride, normals_stride))

    def enableTiming(self, on=True):
        self._check(self._lib.kpl_enable_timing(self._h, int(on)))

    def getTiming(self):
        t = Timing()
        self._check(self._lib.kpl_get_timing(self._h, C.byref(t)))
        return {"calls": t.calls, "index_ms": t.index_ms, "score_ms": t.score_ms, "nms_ms": t.nms_ms,
                "feature_ms": t.feature_ms, "forest_ms": t.forest_ms, "walk": t.walk, "lanes_per_point": t.lanes_per_point,
                "accept_words": t.accept_words}

    def collectStats(self, stream=None):
        self._push()
        st = Stats()
        self._check(self._lib.kpl_collect_stats(self._h, C.byref(st), stream))
        return st.as_dict()
