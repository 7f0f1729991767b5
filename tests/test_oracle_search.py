"""Canonical grid + radius search of the oracle against brute force."""
import numpy as np
import pytest


def brute(xyz, i, r):
    p = xyz[i]
    d = xyz - p                                   # float32
    d2 = d[:, 0] * d[:, 0]
    d2 = d2 + d[:, 1] * d[:, 1]
    d2 = d2 + d[:, 2] * d[:, 2]
    r2 = np.float32(r * r)
    with np.errstate(invalid="ignore"):
        hit = d2 < r2
    return np.flatnonzero(hit), d2


@pytest.mark.parametrize("rmul", [1.0, 4.0, 6.0, 11.0])
def test_radius_search_set_and_order(oracle, cases, rmul):
    xyz, _ = cases.cloud(nan_points=25)
    mr = cases.resolution()
    r = float(np.float32(rmul * mr))
    g = oracle.Grid(xyz, float(np.float32(6 * mr)))        # grid cell is independent of the query radius
    dims, mn, h, nf = g.info()
    assert nf == int(np.isfinite(xyz).all(axis=1).sum())
    mn = np.array(mn, dtype=np.float32)
    cell = np.clip(np.floor((xyz - mn) / h), 0, np.array(dims) - 1)
    lin = (cell[:, 2] * dims[1] + cell[:, 1]) * dims[0] + cell[:, 0]
    rng = np.random.RandomState(1)
    for i in rng.choice(len(xyz), 60, replace=False):
        if not np.isfinite(xyz[i]).all():
            assert g.radius_search(i, r)[2] == 0
            continue
        idx, d2, k = g.radius_search(i, r)
        want, bd2 = brute(xyz, i, r)
        assert k == len(want) and set(idx.tolist()) == set(want.tolist())
        # canonical order: ascending (cell id, index); squared distances are the float32 ones
        key = np.lexsort((want, lin[want]))
        assert np.array_equal(idx, want[key])
        assert np.array_equal(d2, bd2[idx])


def test_sorted_order_is_cell_then_index(oracle, cases):
    xyz, _ = cases.cloud()
    g = oracle.Grid(xyz, 3.0)
    dims, mn, h, nf = g.info()
    order = g.sorted_indices()
    mn = np.array(mn, dtype=np.float32)
    cell = np.clip(np.floor((xyz - mn) / h), 0, np.array(dims) - 1)
    lin = (cell[:, 2] * dims[1] + cell[:, 1]) * dims[0] + cell[:, 0]
    assert np.array_equal(order, np.lexsort((np.arange(len(xyz)), lin)))


def test_cloud_resolution_matches_kdtree(oracle, cases):
    from scipy.spatial import cKDTree
    xyz, _ = cases.cloud()
    d, _ = cKDTree(xyz.astype(np.float64)).query(xyz.astype(np.float64), k=2)
    assert abs(oracle.cloud_resolution(xyz) - d[:, 1].mean()) < 1e-5
