"""ctypes wrapper of the CPU oracle (oracle/kpl_oracle.c).

TEST INFRASTRUCTURE ONLY.  Import this from tests/, __graft_entry__.smoke() and the cpu_baseline
leg of bench.py -- never from the product package.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "libkpl_oracle.so")
_REF = os.path.join(_HERE, "_ref", "libkpl_ref_pairs.so")


def build(ref=True):
    """Compile the oracle (and, if /root/reference exists, the reference pair functions)."""
    subprocess.check_call(["make", "-s", "-C", _HERE, "all"])
    if ref:
        subprocess.check_call(["make", "-s", "-C", _HERE, "ref"])


class _Forest(C.Structure):
    _fields_ = [("ntrees", C.c_int), ("nnodes", C.c_int), ("var_count", C.c_int),
                ("root", C.POINTER(C.c_int)), ("var", C.POINTER(C.c_int)),
                ("thr", C.POINTER(C.c_float)), ("left", C.POINTER(C.c_int)),
                ("right", C.POINTER(C.c_int)), ("value", C.POINTER(C.c_double))]


# neighbor order of the feature loop (kpl_oracle.h): canonical grid order, or ascending (d2, index) = what a
# sorted pcl::search::KdTree handed to setSearchMethod returns
ORDER_CANONICAL, ORDER_SORTED = 0, 1

_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(_LIB):
        build(ref=False)
    L = C.CDLL(_LIB)
    ip, fp, dp = C.POINTER(C.c_int), C.POINTER(C.c_float), C.POINTER(C.c_double)
    i64p = C.POINTER(C.c_int64)
    L.kplo_find_annulus_pair.argtypes = [C.c_int, C.c_float, C.c_float, ip, ip, fp]
    L.kplo_find_annulus_pair.restype = None
    L.kplo_find_bin_pair.argtypes = [C.c_int, C.c_float, ip, ip, fp]
    L.kplo_find_bin_pair.restype = None
    L.kplo_grid_create.argtypes = [fp, C.c_int, C.c_double]
    L.kplo_grid_create.restype = C.c_void_p
    L.kplo_grid_free.argtypes = [C.c_void_p]
    L.kplo_grid_free.restype = None
    L.kplo_grid_info.argtypes = [C.c_void_p, ip, fp, fp, ip]
    L.kplo_grid_info.restype = None
    L.kplo_grid_sorted_indices.argtypes = [C.c_void_p, ip]
    L.kplo_grid_sorted_indices.restype = None
    L.kplo_radius_search.argtypes = [C.c_void_p, fp, C.c_int, C.c_double, ip, fp, C.c_int]
    L.kplo_radius_search.restype = C.c_int
    L.kplo_features.argtypes = [C.c_void_p, fp, fp, C.c_int, C.c_int, C.c_int, C.c_double,
                                ip, C.c_int, fp]
    L.kplo_features.restype = None
    L.kplo_radius_search_sorted.argtypes = [C.c_void_p, fp, C.c_int, C.c_double, ip, fp, C.c_int]
    L.kplo_radius_search_sorted.restype = C.c_int
    L.kplo_features_ordered.argtypes = [C.c_void_p, fp, fp, C.c_int, C.c_int, C.c_int, C.c_double, C.c_int,
                                        ip, C.c_int, fp]
    L.kplo_features_ordered.restype = None
    L.kplo_scores_ordered.argtypes = [C.c_void_p, fp, fp, C.c_int, C.c_int, C.c_int, C.c_double, C.c_int,
                                      C.POINTER(_Forest), fp, C.c_int]
    L.kplo_scores_ordered.restype = None
    L.kplo_detect_ordered.argtypes = [fp, fp, C.c_int, C.c_int, C.c_int, C.c_double, C.c_double,
                                      C.c_double, C.c_int, C.c_int, C.c_float, C.c_int, C.POINTER(_Forest),
                                      fp, ip, C.c_int]
    L.kplo_detect_ordered.restype = C.c_int
    L.kplo_forest_predict_sum.argtypes = [C.POINTER(_Forest), fp, ip]
    L.kplo_forest_predict_sum.restype = C.c_float
    L.kplo_scores.argtypes = [C.c_void_p, fp, fp, C.c_int, C.c_int, C.c_int, C.c_double,
                              C.POINTER(_Forest), fp, C.c_int]
    L.kplo_scores.restype = None
    L.kplo_nms.argtypes = [C.c_void_p, fp, fp, C.c_int, C.c_double, C.c_double, C.c_int,
                           C.c_float, ip, C.c_int]
    L.kplo_nms.restype = C.c_int
    L.kplo_detect.argtypes = [fp, fp, C.c_int, C.c_int, C.c_int, C.c_double, C.c_double,
                              C.c_double, C.c_int, C.c_int, C.c_float, C.POINTER(_Forest),
                              fp, ip, C.c_int]
    L.kplo_detect.restype = C.c_int
    L.kplo_alg_counters.argtypes = [C.c_void_p, fp, fp, C.c_int, C.c_int, C.c_int, C.c_double,
                                    C.c_double, C.c_double, C.POINTER(_Forest),
                                    i64p, i64p, i64p, i64p, i64p]
    L.kplo_alg_counters.restype = None
    L.kplo_cloud_resolution.argtypes = [fp, C.c_int]
    L.kplo_cloud_resolution.restype = C.c_double
    L.kplo_estimate_normals.argtypes = [fp, C.c_int, C.c_int, C.c_double, fp, fp, fp]
    L.kplo_estimate_normals.restype = None
    L.kplo_integral_image_normals.argtypes = [fp, C.c_int, C.c_int, C.c_float, fp, fp, fp]
    L.kplo_integral_image_normals.restype = None
    _lib = L
    return L


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _i32(a):
    return np.ascontiguousarray(a, dtype=np.int32)


def _p(a, t):
    return a.ctypes.data_as(C.POINTER(t))


def find_annulus_pair(n, distance, support):
    i, p, w = C.c_int(), C.c_int(), C.c_float()
    lib().kplo_find_annulus_pair(n, distance, support, C.byref(i), C.byref(p), C.byref(w))
    return i.value, p.value, np.float32(w.value)


def find_bin_pair(n, cosine):
    i, p, w = C.c_int(), C.c_int(), C.c_float()
    lib().kplo_find_bin_pair(n, cosine, C.byref(i), C.byref(p), C.byref(w))
    return i.value, p.value, np.float32(w.value)


class RefPairs:
    """The reference's own findAnnulusPair/findBinPair (oracle/_ref), when it has been built."""

    def __init__(self):
        self.lib = C.CDLL(_REF)
        ir, fr = C.POINTER(C.c_int), C.POINTER(C.c_float)
        self.fa = getattr(self.lib, "_Z15findAnnulusPairiffRiS_Rf")
        self.fa.argtypes = [C.c_int, C.c_float, C.c_float, ir, ir, fr]
        self.fa.restype = None
        self.fb = getattr(self.lib, "_Z11findBinPairifRiS_Rf")
        self.fb.argtypes = [C.c_int, C.c_float, ir, ir, fr]
        self.fb.restype = None

    @staticmethod
    def available():
        return os.path.exists(_REF)

    def annulus(self, n, distance, support):
        i, p, w = C.c_int(), C.c_int(), C.c_float()
        self.fa(n, distance, support, C.byref(i), C.byref(p), C.byref(w))
        return i.value, p.value, np.float32(w.value)

    def bin(self, n, cosine):
        i, p, w = C.c_int(), C.c_int(), C.c_float()
        self.fb(n, cosine, C.byref(i), C.byref(p), C.byref(w))
        return i.value, p.value, np.float32(w.value)


class Forest:
    """Holds the node arrays alive and exposes the C struct."""

    def __init__(self, root, var, thr, left, right, value, var_count):
        self.root = _i32(root)
        self.var = _i32(var)
        self.thr = _f32(thr)
        self.left = _i32(left)
        self.right = _i32(right)
        self.value = np.ascontiguousarray(value, dtype=np.float64)
        self.var_count = int(var_count)
        self.c = _Forest(len(self.root), len(self.var), self.var_count,
                         _p(self.root, C.c_int), _p(self.var, C.c_int), _p(self.thr, C.c_float),
                         _p(self.left, C.c_int), _p(self.right, C.c_int),
                         _p(self.value, C.c_double))

    @property
    def ntrees(self):
        return len(self.root)

    def predict_sum(self, x):
        x = _f32(x)
        d = C.c_int()
        s = lib().kplo_forest_predict_sum(C.byref(self.c), _p(x, C.c_float), C.byref(d))
        return np.float32(s), d.value


class Grid:
    def __init__(self, xyz, cell):
        self.xyz = _f32(xyz).reshape(-1, 3)
        self.n = self.xyz.shape[0]
        self.h = lib().kplo_grid_create(_p(self.xyz, C.c_float), self.n, float(cell))
        if not self.h:
            raise MemoryError("oracle grid too large")

    def __del__(self):
        if getattr(self, "h", None):
            lib().kplo_grid_free(self.h)
            self.h = None

    def info(self):
        dims = (C.c_int * 3)()
        mn = (C.c_float * 3)()
        h = C.c_float()
        nf = C.c_int()
        lib().kplo_grid_info(self.h, dims, mn, C.byref(h), C.byref(nf))
        return list(dims), [np.float32(v) for v in mn], np.float32(h.value), nf.value

    def sorted_indices(self):
        nf = self.info()[3]
        out = np.empty(max(nf, 1), dtype=np.int32)
        lib().kplo_grid_sorted_indices(self.h, _p(out, C.c_int))
        return out[:nf]

    def radius_search(self, i, radius, cap=None, sorted_results=False):
        cap = self.n if cap is None else cap
        idx = np.empty(max(cap, 1), dtype=np.int32)
        d2 = np.empty(max(cap, 1), dtype=np.float32)
        fn = lib().kplo_radius_search_sorted if sorted_results else lib().kplo_radius_search
        k = fn(self.h, _p(self.xyz, C.c_float), int(i), float(radius),
                                     _p(idx, C.c_int), _p(d2, C.c_float), cap)
        return idx[:min(k, cap)].copy(), d2[:min(k, cap)].copy(), k

    def features(self, nrm, A, B, r_feat, query, order=ORDER_CANONICAL):
        nrm = _f32(nrm).reshape(-1, 3)
        query = _i32(query)
        out = np.empty((len(query), A * B), dtype=np.float32)
        lib().kplo_features_ordered(self.h, _p(self.xyz, C.c_float), _p(nrm, C.c_float), self.n, A, B,
                                    float(r_feat), int(order), _p(query, C.c_int), len(query), _p(out, C.c_float))
        return out

    def scores(self, nrm, A, B, r_feat, forest, threads=1, order=ORDER_CANONICAL):
        nrm = _f32(nrm).reshape(-1, 3)
        out = np.empty(max(self.n, 1), dtype=np.float32)
        lib().kplo_scores_ordered(self.h, _p(self.xyz, C.c_float), _p(nrm, C.c_float), self.n, A, B,
                                  float(r_feat), int(order), C.byref(forest.c), _p(out, C.c_float), threads)
        return out[:self.n]

    def nms(self, scores, r_nms, threshold, draws_remove=False, draws_threshold=0.0, threads=1):
        scores = _f32(scores)
        kp = np.empty(max(self.n, 1), dtype=np.int32)
        k = lib().kplo_nms(self.h, _p(self.xyz, C.c_float), _p(scores, C.c_float), self.n,
                           float(r_nms), float(threshold), int(draws_remove),
                           float(draws_threshold), _p(kp, C.c_int), threads)
        return kp[:k].copy()

    def alg_counters(self, nrm, A, B, r_feat, r_nms, threshold, forest):
        nrm = _f32(nrm).reshape(-1, 3)
        v = [C.c_int64() for _ in range(5)]
        lib().kplo_alg_counters(self.h, _p(self.xyz, C.c_float), _p(nrm, C.c_float), self.n,
                                A, B, float(r_feat), float(r_nms), float(threshold),
                                C.byref(forest.c), *[C.byref(x) for x in v])
        keys = ("sum_kf", "sum_kn", "sum_depth", "n_scored", "n_thresholded")
        return dict(zip(keys, (x.value for x in v)))


def detect(xyz, nrm, A, B, r_feat, r_nms, threshold, forest, non_maxima=True,
           draws_remove=False, draws_threshold=0.0, threads=1, order=ORDER_CANONICAL):
    """Whole path.  Returns (scores[n] float32, keypoint indices int32 ascending)."""
    xyz = _f32(xyz).reshape(-1, 3)
    nrm = _f32(nrm).reshape(-1, 3)
    n = xyz.shape[0]
    scores = np.empty(max(n, 1), dtype=np.float32)
    kp = np.empty(max(n, 1), dtype=np.int32)
    k = lib().kplo_detect_ordered(_p(xyz, C.c_float), _p(nrm, C.c_float), n, A, B, float(r_feat),
                                  float(r_nms), float(threshold), int(non_maxima), int(draws_remove),
                                  float(draws_threshold), int(order), C.byref(forest.c), _p(scores, C.c_float),
                                  _p(kp, C.c_int), threads)
    if k < 0:
        raise MemoryError("oracle grid too large")
    return scores[:n], kp[:k].copy()


def cloud_resolution(xyz):
    xyz = _f32(xyz).reshape(-1, 3)
    return lib().kplo_cloud_resolution(_p(xyz, C.c_float), xyz.shape[0])


def estimate_normals(xyz, k=10, radius=0.0, viewpoint=(0.0, 0.0, 0.0)):
    """pcl::NormalEstimation restated: k > 0 = k-search, else radius search.  Returns (normals[n,3], curvature[n])."""
    xyz = _f32(xyz).reshape(-1, 3)
    n = xyz.shape[0]
    vp = _f32(np.asarray(viewpoint, dtype=np.float32))
    nrm = np.empty((max(n, 1), 3), dtype=np.float32)
    cv = np.empty(max(n, 1), dtype=np.float32)
    lib().kplo_estimate_normals(_p(xyz, C.c_float), n, int(k), float(radius), _p(vp, C.c_float),
                                _p(nrm, C.c_float), _p(cv, C.c_float))
    return nrm[:n], cv[:n]


def integral_image_normals(xyz, width, height, smoothing_size=5.0, viewpoint=(0.0, 0.0, 0.0)):
    """pcl::IntegralImageNormalEstimation (SIMPLE_3D_GRADIENT, smoothing size as set by the detector's fallback)
    restated for an organized cloud xyz[height * width, 3] (row-major).  Returns (normals[n,3], curvature[n])."""
    xyz = _f32(xyz).reshape(-1, 3)
    n = int(width) * int(height)
    assert xyz.shape[0] == n
    vp = _f32(np.asarray(viewpoint, dtype=np.float32))
    nrm = np.empty((max(n, 1), 3), dtype=np.float32)
    cv = np.empty(max(n, 1), dtype=np.float32)
    lib().kplo_integral_image_normals(_p(xyz, C.c_float), int(width), int(height), float(smoothing_size), _p(vp, C.c_float),
                                      _p(nrm, C.c_float), _p(cv, C.c_float))
    return nrm[:n], cv[:n]
