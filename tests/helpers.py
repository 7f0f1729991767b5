"""Shared seeded inputs for the tests (small enough for the oracle to finish in seconds)."""
import functools

import numpy as np

from oracle import kplo
from tools import synth


@functools.lru_cache(maxsize=None)
def cloud(nx=80, ny=60, seed=1, shuffle=True, nan_points=0, nan_normals=0, layers=1):
    xyz, nrm = synth.make_cloud(nx, ny, seed=seed, nan_points=nan_points, nan_normals=nan_normals,
                                overlap_layers=layers)
    if shuffle:
        xyz, nrm = synth.shuffle_cloud(xyz, nrm, seed + 1000)
    xyz.setflags(write=False)
    nrm.setflags(write=False)
    return xyz, nrm


@functools.lru_cache(maxsize=None)
def resolution(nx=80, ny=60, seed=1):
    xyz, _ = cloud(nx, ny, seed)
    return kplo.cloud_resolution(xyz)


@functools.lru_cache(maxsize=None)
def trained_forest(A=5, B=6, nx=80, ny=60, seed=1, ntrees=10, max_depth=10, rmul=6.0):
    """Extra-trees forest trained on the oracle's features of the (nx, ny, seed) cloud."""
    xyz, nrm = cloud(nx, ny, seed)
    r = rmul * resolution(nx, ny, seed)
    g = kplo.Grid(xyz, r)
    feat = g.features(nrm, A, B, r, np.arange(len(xyz)))
    ok = np.isfinite(feat).all(axis=1)
    lab = synth.saliency_labels(feat[ok], A, B)
    return synth.train_extra_trees(feat[ok], lab, ntrees=ntrees, max_depth=max_depth, seed=seed + 1)


def usable_cores():
    """threads for the oracle: the affinity mask / cgroup quota, not the machine's core count"""
    import os
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            n = min(n, max(1, int(q) // int(per)))
    except (OSError, ValueError):
        pass
    return max(1, min(n, 16))


def oracle_forest(fa):
    return kplo.Forest(fa.root, fa.var, fa.thr, fa.left, fa.right, fa.value, fa.var_count)


def load_arrays(det, fa):
    det.loadForestArrays(fa.root, fa.var, fa.thr, fa.left, fa.right, fa.value, fa.var_count)


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


def same_bits(a, b):
    """bitwise equality of float arrays, NaNs compared as NaN == NaN (any payload)."""
    a = np.ascontiguousarray(a, dtype=np.float32)
    b = np.ascontiguousarray(b, dtype=np.float32)
    if a.shape != b.shape:
        return False
    na, nb = np.isnan(a), np.isnan(b)
    return bool(np.array_equal(na, nb) and np.array_equal(bits(a)[~na], bits(b)[~nb]))
