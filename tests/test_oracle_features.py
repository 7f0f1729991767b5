"""Oracle feature rows: against an independent step-by-step numpy restatement (small sample),
properties, and the committed fixture."""
import os

import numpy as np
import pytest

f32 = np.float32
GOLD = os.path.join(os.path.dirname(__file__), "golden")


def pair(n, v, dim):
    k = int(np.floor(f32(v) / dim))
    if k == n:
        k -= 1
    center = f32(f32(k) * dim) + f32(dim / f32(2))
    w = f32(f32(v) - center) / dim
    p = k + 1 if w > 0 else k - 1
    if p == -1:
        p = 0
    if p == n:
        p = k
    return k, p, f32(abs(w))


def py_features(oracle_grid, xyz, nrm, i, A, B, r):
    """Appendix A of SURVEY.md, written with numpy float32 scalars (no FMA possible)."""
    idx, d2, _ = oracle_grid.radius_search(i, r)
    H = np.zeros((A, B), dtype=f32)
    support = f32(r)
    adim, bdim = support / f32(A), f32(2) / f32(B)
    for j, dd in list(zip(idx, d2))[1:]:
        nq = nrm[j]
        if not np.isfinite(nq).all():
            continue
        npv = nrm[i]
        dot = f32(npv[0] * nq[0]) + f32(f32(npv[1] * nq[1]) + f32(npv[2] * nq[2]))
        c = f32(1) - dot
        c = f32(min(max(c, f32(0)), f32(2)))
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


@pytest.mark.parametrize("A,B", [(5, 6), (8, 10), (1, 1)])
def test_features_vs_python_restatement(oracle, cases, A, B):
    xyz, nrm = cases.cloud(nan_normals=30)
    r = float(f32(6 * cases.resolution()))
    g = oracle.Grid(xyz, r)
    q = np.array([0, 7, 1234, 2500, len(xyz) - 1, 333, 4000], dtype=np.int32)
    q = q[np.isfinite(xyz[q]).all(axis=1)]
    got = g.features(nrm, A, B, r, q)
    for row, i in zip(got, q):
        assert cases.same_bits(row, py_features(g, xyz, nrm, int(i), A, B, r)), i


def test_feature_rows_are_unit_or_zero(oracle, cases):
    xyz, nrm = cases.cloud()
    r = float(f32(6 * cases.resolution()))
    A, B = 5, 6
    feat = oracle.Grid(xyz, r).features(nrm, A, B, r, np.arange(len(xyz), dtype=np.int32))
    norms = np.linalg.norm(feat.reshape(-1, A, B).astype(np.float64), axis=2)
    assert np.all((np.abs(norms - 1) < 1e-6) | (norms == 0))
    assert (feat >= 0).all() and (feat <= 1.0000001).all()


def test_non_finite_query_gives_nan_row(oracle, cases):
    xyz, nrm = cases.cloud(nan_points=40)
    bad = np.flatnonzero(~np.isfinite(xyz).all(axis=1))[:3].astype(np.int32)
    feat = oracle.Grid(xyz, 3.0).features(nrm, 5, 6, 3.0, bad)
    assert np.isnan(feat).all()


def test_committed_fixture(oracle, cases):
    z = np.load(os.path.join(GOLD, "small_case.npz"))
    xyz, nrm, q = z["xyz"], z["nrm"], z["query"]
    r = float(z["r_feat"])
    g = oracle.Grid(xyz, r)
    for A, B in ((5, 6), (5, 10), (8, 10)):
        assert cases.same_bits(g.features(nrm, A, B, r, q), z["feat_%dx%d" % (A, B)])
