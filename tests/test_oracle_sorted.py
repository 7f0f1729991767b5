"""Oracle, sorted-search mode (kplo.ORDER_SORTED): the neighbor order a caller gets from
pcl::search::KdTree(sorted = true) handed to the inherited setSearchMethod -- ascending (squared distance,
index), FLANN's DistanceIndex order.  Checked against brute force and an independent numpy restatement."""
import numpy as np
import pytest

from tests.test_oracle_features import f32, pair


def brute_sorted(xyz, i, r):
    """all j with d2 < (float)(r*r), d2 = ((dx*dx) + dy*dy) + dz*dz in float32, ascending (d2, j)"""
    d = (xyz[i][None, :] - xyz).astype(f32)
    d2 = (d[:, 0] * d[:, 0]).astype(f32)
    d2 = (d2 + (d[:, 1] * d[:, 1]).astype(f32)).astype(f32)
    d2 = (d2 + (d[:, 2] * d[:, 2]).astype(f32)).astype(f32)
    ok = np.flatnonzero(np.isfinite(d2) & (d2 < f32(float(r) * float(r))))
    order = np.lexsort((ok, d2[ok]))
    return ok[order].astype(np.int32), d2[ok][order]


def py_features_sorted(xyz, nrm, i, A, B, r):
    idx, d2 = brute_sorted(xyz, i, r)
    H = np.zeros((A, B), dtype=f32)
    support = f32(r)
    adim, bdim = support / f32(A), f32(2) / f32(B)
    for j, dd in list(zip(idx, d2))[1:]:            # hpp:336: element 0 is dropped
        nq, npv = nrm[j], nrm[i]
        if not np.isfinite(nq).all():
            continue
        dot = f32(npv[0] * nq[0]) + f32(f32(npv[1] * nq[1]) + f32(npv[2] * nq[2]))
        c = f32(min(max(f32(1) - dot, f32(0)), f32(2)))
        a, ap, aw = pair(A, np.sqrt(f32(dd)), adim)
        b, bp, bw = pair(B, c, bdim)
        H[a, b] += f32(f32(1) - bw) * f32(f32(1) - aw)
        H[a, bp] += bw * f32(f32(1) - aw)
        H[ap, b] += f32(f32(1) - bw) * aw
        H[ap, bp] += bw * aw
    for a in range(A):
        s = f32(0)
        for k in range(B):
            s = f32(s + f32(H[a, k] * H[a, k]))
        nr = np.sqrt(s)
        if nr > 0:
            H[a] = H[a] / nr
    return H.reshape(-1)


def lattice(nx=30, ny=24, dup=12):
    """a REGULAR lattice (many exactly equal distances) with a few exact duplicates of points appended:
    the tie-break by index decides the order, and for a duplicated query element 0 is not the query"""
    gx, gy = np.meshgrid(np.arange(nx, dtype=f32), np.arange(ny, dtype=f32), indexing="ij")
    z = (0.25 * np.sin(gx * 0.5) + 0.125 * np.cos(gy * 0.75)).astype(f32)
    z = (np.round(z * 8) / 8).astype(f32)                      # few distinct heights: more ties
    xyz = np.stack([gx.ravel(), gy.ravel(), z.ravel()], axis=1).astype(f32)
    rng = np.random.RandomState(5)
    src = rng.choice(len(xyz), dup, replace=False)
    xyz = np.concatenate([xyz, xyz[src]]).astype(f32)
    nrm = rng.normal(size=xyz.shape).astype(f32)
    nrm /= np.linalg.norm(nrm, axis=1, keepdims=True)
    perm = rng.permutation(len(xyz))
    return np.ascontiguousarray(xyz[perm]), np.ascontiguousarray(nrm[perm].astype(f32))


def test_sorted_search_equals_brute_force(oracle, cases):
    xyz, nrm = cases.cloud(nan_points=25)
    r = float(f32(5 * cases.resolution()))
    g = oracle.Grid(xyz, r)
    for i in (0, 17, 999, 2500, len(xyz) - 1):
        if not np.isfinite(xyz[i]).all():
            continue
        idx, d2, k = g.radius_search(i, r, sorted_results=True)
        bi, bd = brute_sorted(xyz, i, r)
        assert k == len(bi) and np.array_equal(idx, bi) and cases.same_bits(d2, bd)
        assert idx[0] == i and d2[0] == 0
        cidx, _, ck = g.radius_search(i, r)
        assert ck == k and np.array_equal(np.sort(cidx), np.sort(idx))       # the same set, another order


def test_ties_are_broken_by_index(oracle, cases):
    xyz, nrm = lattice()
    r = 3.3
    g = oracle.Grid(xyz, r)
    ties = firsts = 0
    for i in range(0, len(xyz), 7):
        idx, d2, k = g.radius_search(i, r, sorted_results=True)
        bi, bd = brute_sorted(xyz, i, r)
        assert np.array_equal(idx, bi) and cases.same_bits(d2, bd)
        same = d2[1:] == d2[:-1]
        assert np.all(idx[1:][same] > idx[:-1][same])
        ties += int(same.sum())
        firsts += int(idx[0] != i)
    assert ties > 1000
    # a duplicated point with the larger index meets its twin first (hpp:336 then drops the twin, not the query)
    dup = [i for i in range(len(xyz)) if g.radius_search(i, r, sorted_results=True)[0][0] != i]
    assert len(dup) == 12


@pytest.mark.parametrize("A,B", [(5, 6), (8, 10)])
def test_sorted_features_vs_python_restatement(oracle, cases, A, B):
    for xyz, nrm, r in ((*cases.cloud(nan_normals=30), float(f32(6 * cases.resolution()))), (*lattice(), 3.3)):
        g = oracle.Grid(xyz, r)
        q = np.array([0, 7, 333, len(xyz) - 1, 500], dtype=np.int32)
        q = q[np.isfinite(xyz[q]).all(axis=1)]
        got = g.features(nrm, A, B, r, q, order=oracle.ORDER_SORTED)
        for row, i in zip(got, q):
            assert cases.same_bits(row, py_features_sorted(xyz, nrm, int(i), A, B, r)), i
        # the order matters: the canonical rows differ somewhere, by rounding or by the dropped neighbor
        assert not cases.same_bits(got, g.features(nrm, A, B, r, q))


def test_sorted_detect_is_consistent(oracle, cases):
    A, B = 5, 6
    xyz, nrm = cases.cloud()
    mr = cases.resolution()
    r, rn = float(f32(6 * mr)), float(f32(4 * mr))
    fa = cases.trained_forest(A, B)
    of = cases.oracle_forest(fa)
    sc, kp = oracle.detect(xyz, nrm, A, B, r, rn, 0.5, of, order=oracle.ORDER_SORTED)
    g = oracle.Grid(xyz, r)
    assert cases.same_bits(sc, g.scores(nrm, A, B, r, of, order=oracle.ORDER_SORTED))
    assert np.array_equal(kp, g.nms(sc, rn, 0.5))
    assert cases.same_bits(sc, oracle.detect(xyz, nrm, A, B, r, rn, 0.5, of, order=oracle.ORDER_SORTED, threads=4)[0])


def test_sorted_fixture(oracle, cases):
    """tests/golden/sorted_case.npz: the oracle's sorted mode on the two committed clouds (regression anchor; the arrays a
    PCL + OpenCV run with a sorted search tree could regenerate bit for bit)."""
    import os
    from tools import forest_yaml
    gold = os.path.join(os.path.dirname(__file__), "golden")
    z, s = np.load(os.path.join(gold, "small_case.npz")), np.load(os.path.join(gold, "sorted_case.npz"))
    r, rn = float(z["r_feat"]), float(z["r_nms"])
    g = oracle.Grid(z["xyz"], r)
    for A, B in ((5, 6), (8, 10)):
        assert cases.same_bits(g.features(z["nrm"], A, B, r, z["query"], order=oracle.ORDER_SORTED), s["small_feat_%dx%d" % (A, B)])
    of = cases.oracle_forest(forest_yaml.load_forest(os.path.join(gold, "small_forest.yaml.gz")))
    sc, kp = oracle.detect(z["xyz"], z["nrm"], 5, 6, r, rn, float(f32(0.5)), of, order=oracle.ORDER_SORTED)
    assert cases.same_bits(sc, s["small_scores"]) and np.array_equal(kp, s["small_kp_thr050"])
    assert not cases.same_bits(s["small_scores"], z["scores"])
