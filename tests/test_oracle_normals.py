"""Oracle normal estimation (pcl::NormalEstimation restated, /root/reference/src/main_test_detector.cpp:162-169
and include/impl/KeypointLearning.hpp:125-148) against independent numpy/scipy computations."""
import numpy as np
import pytest
from scipy.spatial import cKDTree

from oracle import kplo
from tests import helpers


def angle(a, b):
    """unsigned angle between directions (atan2 form: arccos is ill-conditioned near 0)"""
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return np.arctan2(np.linalg.norm(np.cross(a, b), axis=-1), np.abs(np.sum(a * b, axis=-1)))


def test_plane_gives_the_exact_normal_and_zero_curvature():
    rng = np.random.default_rng(3)
    uv = rng.uniform(-5, 5, size=(500, 2)).astype(np.float32)
    xyz = np.stack([uv[:, 0], uv[:, 1], np.full(500, 2.0, np.float32)], axis=1)
    for kw in ({"k": 10}, {"k": 0, "radius": 1.5}):
        nrm, curv = kplo.estimate_normals(xyz, viewpoint=(0, 0, 10), **kw)
        ok = np.isfinite(nrm).all(1)
        assert ok.sum() > 450
        assert np.array_equal(nrm[ok], np.tile(np.float32([0, 0, 1]), (ok.sum(), 1)))
        assert np.all(curv[ok] == 0)
        flipped, _ = kplo.estimate_normals(xyz, viewpoint=(0, 0, -10), **kw)
        assert np.array_equal(flipped[ok], -nrm[ok])


def test_k_search_matches_eigh_on_a_surface():
    xyz, ref = helpers.cloud(60, 50, seed=5)
    nrm, curv = kplo.estimate_normals(xyz, k=10, viewpoint=(0, 0, 1e4))
    x64 = xyz.astype(np.float64)
    _, nn = cKDTree(x64).query(x64, k=10)
    checked = 0
    for i in range(0, len(xyz), 5):
        cov = np.cov(x64[nn[i]].T, bias=True)
        w, v = np.linalg.eigh(cov)
        if (w[1] - w[0]) < 1e-3 * w[2]:
            continue                                   # direction ill-defined
        assert angle(nrm[i], v[:, 0]) < 1e-5
        assert abs(curv[i] - w[0] / w.sum()) < 1e-5
        assert nrm[i] @ (np.float64([0, 0, 1e4]) - x64[i]) >= 0     # flipNormalTowardsViewpoint
        checked += 1
    assert checked > 300
    assert np.median(angle(nrm, ref)) < 0.8                         # the synthetic surface is rough at this scale


def test_radius_search_matches_eigh_and_counts_the_query_itself():
    xyz, _ = helpers.cloud(50, 40, seed=6)
    mr = kplo.cloud_resolution(xyz)
    r = float(np.float32(2.5 * mr))
    nrm, _ = kplo.estimate_normals(xyz, k=0, radius=r, viewpoint=(0, 0, 1e4))
    g = kplo.Grid(xyz, r)
    x64 = xyz.astype(np.float64)
    checked = 0
    for i in range(0, len(xyz), 7):
        idx, _, _ = g.radius_search(i, r)
        assert i in idx
        if len(idx) < 3:
            assert np.isnan(nrm[i]).all()
            continue
        w, v = np.linalg.eigh(np.cov(x64[idx].T, bias=True))
        if (w[1] - w[0]) < 1e-3 * w[2]:
            continue
        assert angle(nrm[i], v[:, 0]) < 1e-5
        checked += 1
    assert checked > 150


def test_degenerate_inputs():
    xyz = np.float32([[0, 0, 0], [1, 0, 0], [np.nan, 0, 0], [0, 1, 0], [5, 5, 5]])
    nrm, curv = kplo.estimate_normals(xyz, k=3, viewpoint=(0, 0, 1))
    assert np.isnan(nrm[2]).all() and np.isnan(curv[2])
    assert np.array_equal(nrm[0], np.float32([0, 0, 1]))           # its 3 nearest span the z = 0 plane
    two, _ = kplo.estimate_normals(xyz[:2], k=10)
    assert np.isnan(two).all()                                      # fewer than 3 neighbors
    none, _ = kplo.estimate_normals(np.zeros((0, 3), np.float32), k=10)
    assert none.shape == (0, 3)
    far, _ = kplo.estimate_normals(xyz, k=0, radius=0.5)
    assert np.isnan(far).all()                                      # nobody has 3 neighbors within 0.5


def test_k_search_ties_are_broken_by_index():
    # a regular lattice has many equal distances: the result must not depend on the point order
    g = np.stack(np.meshgrid(np.arange(6), np.arange(6), [0.0]), -1).reshape(-1, 3).astype(np.float32)
    g[:, 2] = 0.1 * g[:, 0]
    n1, _ = kplo.estimate_normals(g, k=5, viewpoint=(0, 0, 100))
    assert np.isfinite(n1).all()
    expect = np.float64([-0.1, 0, 1]) / np.sqrt(1.01)
    assert np.all(angle(n1, np.tile(expect, (len(g), 1))) < 1e-6)


def test_committed_normals_fixture():
    import os
    gold = os.path.join(os.path.dirname(__file__), "golden")
    z, c = np.load(os.path.join(gold, "normals_case.npz")), np.load(os.path.join(gold, "small_case.npz"))
    nk, ck = kplo.estimate_normals(c["xyz"], k=int(z["k"]), viewpoint=z["viewpoint"])
    nr, cr = kplo.estimate_normals(c["xyz"], k=0, radius=float(z["radius"]), viewpoint=z["viewpoint"])
    for a, b in ((nk, z["nrm_k"]), (ck, z["curv_k"]), (nr, z["nrm_r"]), (cr, z["curv_r"])):
        assert np.array_equal(a.view(np.uint32), b.view(np.uint32))
