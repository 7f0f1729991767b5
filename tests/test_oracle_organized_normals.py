"""CPU: the oracle's restatement of pcl::IntegralImageNormalEstimation (SIMPLE_3D_GRADIENT, smoothing size 5,
what /root/reference/include/impl/KeypointLearning.hpp:138-145 runs on an organized cloud without normals)
against a second, independently written restatement in Python and against what the method must give on
surfaces whose normals are known.  PCL itself is absent: "parity unpinned"."""
import math

import numpy as np
import pytest

from oracle import kplo
from tools import synth

F32 = np.float32


def py_integral_image_normals(xyz, W, H, smoothing=5.0, vp=(0.0, 0.0, 0.0)):
    """features/impl/integral_image_normal.hpp + integral_image2D.hpp of PCL 1.8.0, written from the published
    source with Python scalars (float64 = double; np.float32 where PCL computes in float)."""
    P = np.asarray(xyz, dtype=np.float32).reshape(H, W, 3)
    z = P[:, :, 2]
    out = np.full((H, W, 3), np.nan, dtype=np.float32)
    border = int(smoothing)
    if W <= 2 * border or H <= 2 * border:
        return out.reshape(-1, 3)
    change = np.full((H, W), 255, dtype=np.uint8)
    factor = F32(20.0) * F32(0.001)
    for r in range(H - 1):
        for c in range(W - 1):
            d, dr, dd = z[r, c], z[r, c + 1], z[r + 1, c]
            lim = factor * (abs(d) + F32(1.0)) * F32(2.0)
            if (not (math.isfinite(d) and math.isfinite(dr))) or abs(F32(d - dr)) > lim:
                change[r, c] = change[r, c + 1] = 0
            if (not (math.isfinite(d) and math.isfinite(dd))) or abs(F32(d - dd)) > lim:
                change[r, c] = change[r + 1, c] = 0
    flat = np.where(change.reshape(-1) == 0, F32(0), F32(W + H)).astype(np.float32)     # rows are contiguous
    one, diag = F32(1.0), F32(1.4)
    for r in range(1, H):
        for c in range(1, W):
            i = r * W + c
            m = min(min(flat[i - W - 1] + diag, flat[i - W] + one), min(flat[i - 1] + one, flat[i - W + 1] + diag))
            if m < flat[i]:
                flat[i] = m
    for r in range(H - 2, -1, -1):
        for c in range(W - 2, -1, -1):
            i = r * W + c
            m = min(min(flat[i + W - 1] + diag, flat[i + W] + one), min(flat[i + 1] + one, flat[i + W + 1] + diag))
            if m < flat[i]:
                flat[i] = m
    dist = flat.reshape(H, W)
    ii = [[(0.0, 0.0, 0.0)] * (W + 1) for _ in range(H + 1)]
    for r in range(H):
        row = [(0.0, 0.0, 0.0)] * (W + 1)
        prev = ii[r]
        for c in range(W):
            e = P[r, c]
            v = [prev[c + 1][a] + row[c][a] - prev[c][a] for a in range(3)]
            if math.isfinite(F32(F32(e[0] + e[1]) + e[2])):
                v = [v[a] + float(e[a]) for a in range(3)]
            row[c + 1] = tuple(v)
        ii[r + 1] = row

    def rect(sx, sy, w, h):
        return [ii[sy + h][sx + w][a] + ii[sy][sx][a] - ii[sy][sx + w][a] - ii[sy + h][sx][a] for a in range(3)]

    for r in range(border, H - border):
        for c in range(border, W - border):
            if not math.isfinite(z[r, c]):
                continue
            s = min(dist[r, c], F32(smoothing))
            if not s > 2.0:
                continue
            w = h = int(s)
            a1, a0 = rect(c + w // 2, r - h // 2, 1, h), rect(c - w // 2, r - h // 2, 1, h)
            gx = [a1[k] - a0[k] for k in range(3)]
            b1, b0 = rect(c - w // 2, r + h // 2, w, 1), rect(c - w // 2, r - h // 2, w, 1)
            gy = [b1[k] - b0[k] for k in range(3)]
            n = [gy[1] * gx[2] - gy[2] * gx[1], gy[2] * gx[0] - gy[0] * gx[2], gy[0] * gx[1] - gy[1] * gx[0]]
            ln = n[0] * n[0] + n[1] * n[1] + n[2] * n[2]
            if ln == 0.0:
                continue
            root = math.sqrt(ln)
            nf = [F32(n[k] / root) for k in range(3)]
            v = [F32(vp[k]) - P[r, c, k] for k in range(3)]
            if F32(F32(v[0] * nf[0] + v[1] * nf[1]) + v[2] * nf[2]) < 0:
                nf = [-x for x in nf]
            out[r, c] = nf
    return out.reshape(-1, 3)


def depth_image(W, H, seed, step=None, holes=0, bumps=0.05, period=9.0):
    """a pinhole view of a smooth surface: x, y from the pixel and the depth; an optional depth step and NaN holes"""
    rng = np.random.default_rng(seed)
    v, u = np.mgrid[0:H, 0:W].astype(np.float32)
    z = (1.5 + 0.002 * u + 0.003 * v + bumps * np.sin(u / period) * np.cos(v / (period - 2.0))).astype(np.float32)
    if step is not None:
        z[:, step:] += np.float32(0.4)
    xyz = np.stack([(u - W / 2) * z / 300.0, (v - H / 2) * z / 300.0, z], -1).astype(np.float32)
    for _ in range(holes):
        r, c = int(rng.integers(0, H)), int(rng.integers(0, W))
        xyz[r:r + int(rng.integers(1, 4)), c:c + int(rng.integers(1, 4))] = np.nan
    return xyz.reshape(-1, 3)


def same_bits(a, b):
    a, b = np.ascontiguousarray(a, np.float32), np.ascontiguousarray(b, np.float32)
    return a.shape == b.shape and np.array_equal(a.view(np.uint32)[~np.isnan(a)], b.view(np.uint32)[~np.isnan(b)]) and \
        np.array_equal(np.isnan(a), np.isnan(b))


@pytest.mark.parametrize("W,H,step,holes", [(31, 24, None, 0), (40, 23, 17, 0), (33, 29, 11, 6), (12, 14, None, 1)])
def test_oracle_equals_the_python_restatement(W, H, step, holes):
    xyz = depth_image(W, H, seed=W * H, step=step, holes=holes)
    nrm, curv = kplo.integral_image_normals(xyz, W, H, 5.0)
    ref = py_integral_image_normals(xyz, W, H, 5.0)
    assert same_bits(nrm, ref)
    assert np.isnan(curv).all()                       # SIMPLE_3D_GRADIENT computes no curvature
    assert np.isfinite(nrm[:, 0]).sum() > 0


def test_plane_border_step_and_viewpoint():
    W, H = 48, 36
    v, u = np.mgrid[0:H, 0:W].astype(np.float64)
    # the plane n . p = d seen by a pinhole camera: depth from the ray, so every point lies on it exactly (up to float)
    n_true = np.array([0.2, -0.3, -1.0])
    n_true /= np.linalg.norm(n_true)
    rays = np.stack([(u - W / 2) / 200.0, (v - H / 2) / 200.0, np.ones_like(u)], -1)
    t = -2.0 / (rays @ n_true)
    xyz = (rays * t[..., None]).astype(np.float32).reshape(-1, 3)
    nrm, _ = kplo.integral_image_normals(xyz, W, H, 5.0)
    img = nrm.reshape(H, W, 3)
    inner = img[5:H - 5, 5:W - 5].reshape(-1, 3)
    assert np.isfinite(inner).all()
    # towards the viewpoint (0, 0, 0): n . (0 - p) > 0, and parallel to the plane's normal
    assert (np.einsum("ij,ij->i", inner, -xyz.reshape(H, W, 3)[5:H - 5, 5:W - 5].reshape(-1, 3)) > 0).all()
    assert np.abs(np.abs(inner @ n_true) - 1.0).max() < 1e-5
    assert np.isnan(img[:5]).all() and np.isnan(img[H - 5:]).all() and np.isnan(img[:, :5]).all() and np.isnan(img[:, W - 5:]).all()
    # the other side of the sensor: all normals flip
    far, _ = kplo.integral_image_normals(xyz, W, H, 5.0, viewpoint=(0.0, 0.0, 10.0))
    assert same_bits(far.reshape(H, W, 3)[5:H - 5, 5:W - 5], -img[5:H - 5, 5:W - 5])
    # a depth step: no normal closer than 2 pixels (chamfer) to the discontinuity, normals again further away
    z = xyz.reshape(H, W, 3).copy()
    z[:, 24:] *= np.float32(1.3)
    stepped, _ = kplo.integral_image_normals(z.reshape(-1, 3), W, H, 5.0)
    s = stepped.reshape(H, W, 3)
    assert np.isnan(s[5:H - 5, 22:26]).all()
    assert np.isfinite(s[5:H - 5, 10]).all() and np.isfinite(s[5:H - 5, 36]).all()


def test_too_small_or_empty_images_are_all_nan():
    xyz = depth_image(10, 30, seed=1)
    nrm, curv = kplo.integral_image_normals(xyz, 10, 30, 5.0)
    assert np.isnan(nrm).all() and np.isnan(curv).all()
    nrm, _ = kplo.integral_image_normals(np.zeros((0, 3), np.float32), 0, 0, 5.0)
    assert nrm.shape == (0, 3)


def test_oracle_reproduces_the_committed_organized_golden():
    """tests/golden/organized_case.npz (tools/make_organized_golden.py): the array a PCL 1.8 run must reproduce; the oracle
    and -- tests/test_gpu_normals.py -- the device kernels are held to it bit for bit"""
    import os
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "organized_case.npz"))
    W, H = int(z["width"]), int(z["height"])
    for name in ("origin", "off"):
        nrm, curv = kplo.integral_image_normals(z["xyz"], W, H, float(z["smoothing"]), tuple(float(x) for x in z["viewpoint_" + name]))
        assert same_bits(nrm, z["normals_" + name]) and np.isnan(curv).all()
    fin = np.isfinite(z["normals_origin"]).all(axis=1)
    assert 0.5 < fin.mean() < 0.98                       # borders, holes and the depth step are left without a normal
    assert not np.array_equal(z["normals_origin"][fin], z["normals_off"][fin])       # the viewpoint flips some of them
