"""Seeded synthetic inputs for the scoring path: 2.5D clouds with normals, and forests.

Everything is driven by an explicit 64-bit integer generator (splitmix64, counter based) mapped
to floats by this file -- no numpy.random -- so a seed names the same integers everywhere.
(The transcendental functions of the height field may differ in the last ulp between numpy
builds; anything that has to be byte-stable is therefore committed as data under tests/golden/.)

Cloud recipe follows SURVEY.md 8(d) cfg2 and its realism note: jittered grid, smooth height
field + bumps + creases + depth steps, analytic unit normals, ~5 % patch-wise sign inversion.
"""
import math

import numpy as np

from .forest_yaml import ForestArrays

_GOLD = np.uint64(0x9E3779B97F4A7C15)
_M1 = np.uint64(0xBF58476D1CE4E5B9)
_M2 = np.uint64(0x94D049BB133111EB)


def _mix(z):
    z = np.asarray(z, dtype=np.uint64)
    with np.errstate(over="ignore"):
        z = (z ^ (z >> np.uint64(30))) * _M1
        z = (z ^ (z >> np.uint64(27))) * _M2
        z = z ^ (z >> np.uint64(31))
    return z


def u64(seed, stream, idx):
    """splitmix64 output for (seed, stream, idx); idx may be an array."""
    idx = np.asarray(idx, dtype=np.uint64)
    with np.errstate(over="ignore"):
        base = _mix(np.uint64(seed) * _GOLD + np.uint64(stream) * np.uint64(0xD1B54A32D192ED03))
        return _mix(base + (idx + np.uint64(1)) * _GOLD)


def uniform(seed, stream, idx):
    """float64 in [0, 1) with 24 random bits (exactly representable in float32)."""
    return (u64(seed, stream, idx) >> np.uint64(40)).astype(np.float64) * (1.0 / 16777216.0)


class Rng:
    """Sequential scalar view of the same generator (for tree building), block buffered."""

    def __init__(self, seed, stream=0):
        self.seed, self.stream, self.k = seed, stream, 0
        self._buf, self._base = None, 0

    def next(self):
        j = self.k - self._base
        if self._buf is None or j >= len(self._buf):
            self._base = self.k
            self._buf = uniform(self.seed, self.stream, np.arange(self.k, self.k + 4096)).tolist()
            j = 0
        self.k += 1
        return self._buf[j]

    def randint(self, hi):
        return min(int(self.next() * hi), hi - 1)


# ------------------------------------------------------------------------------------------
def make_cloud(nx=500, ny=400, seed=1, spacing=1.0, jitter=0.25, flip_fraction=0.05,
               nan_points=0, nan_normals=0, overlap_layers=1):
    """Returns (xyz float32 [n,3], normals float32 [n,3]).  n = nx*ny*overlap_layers.

    `overlap_layers` > 1 fuses several offset height fields (cfg5: locally denser cloud).
    """
    clouds, normals = [], []
    for layer in range(overlap_layers):
        s = seed + 7919 * layer
        n = nx * ny
        idx = np.arange(n)
        gx = (idx % nx).astype(np.float64)
        gy = (idx // nx).astype(np.float64)
        x = (gx + (uniform(s, 1, idx) - 0.5) * 2 * jitter + 0.37 * layer) * spacing
        y = (gy + (uniform(s, 2, idx) - 0.5) * 2 * jitter + 0.21 * layer) * spacing
        L = max(nx, ny) * spacing
        z = np.zeros(n)
        dzdx = np.zeros(n)
        dzdy = np.zeros(n)
        # smooth part: 6 sine products
        for k in range(6):
            a = (0.6 + 1.8 * uniform(s, 10, k)) * spacing * (1.0 + k * 0.1)
            f = (2 + 9 * uniform(s, 11, k)) * 2 * math.pi / L
            g = (2 + 9 * uniform(s, 12, k)) * 2 * math.pi / L
            ph, ps = 2 * math.pi * uniform(s, 13, k), 2 * math.pi * uniform(s, 14, k)
            sx, cx = np.sin(f * x + ph), np.cos(f * x + ph)
            sy, cy = np.sin(g * y + ps), np.cos(g * y + ps)
            z += a * sx * cy
            dzdx += a * f * cx * cy
            dzdy += -a * g * sx * sy
        # 12 gaussian bumps, sharp enough that 1-cos spans several bins
        for k in range(12):
            bx, by = uniform(s, 20, k) * nx * spacing, uniform(s, 21, k) * ny * spacing
            sig = (3.0 + 9.0 * uniform(s, 22, k)) * spacing
            amp = (4.0 + 10.0 * uniform(s, 23, k)) * spacing * (1 if uniform(s, 24, k) < 0.7 else -1)
            e = amp * np.exp(-((x - bx) ** 2 + (y - by) ** 2) / (2 * sig * sig))
            z += e
            dzdx += -e * (x - bx) / (sig * sig)
            dzdy += -e * (y - by) / (sig * sig)
        # creases: |sin| ridges along x and y
        fc = 2 * math.pi / (37.0 * spacing)
        ac = 2.5 * spacing
        sc = np.sin(fc * x + 0.3)
        z += ac * np.abs(sc)
        dzdx += ac * np.sign(sc) * fc * np.cos(fc * x + 0.3)
        fc2 = 2 * math.pi / (53.0 * spacing)
        sc2 = np.sin(fc2 * y + 1.1)
        z += 1.5 * spacing * np.abs(sc2)
        dzdy += 1.5 * spacing * np.sign(sc2) * fc2 * np.cos(fc2 * y + 1.1)
        # depth steps (2.5D discontinuities): no points on the riser, normals unaffected
        z += 9.0 * spacing * np.floor(x / (113.0 * spacing)) * (np.floor(y / (97.0 * spacing)) % 2)
        z += 20.0 * spacing * layer * 0.0
        nrm = np.stack([-dzdx, -dzdy, np.ones(n)], axis=1)
        nrm /= np.linalg.norm(nrm, axis=1, keepdims=True)
        # patch-wise sign inversion (locally inconsistent normal orientation, as on real scans)
        patch = (idx % nx) // 16 + ((idx // nx) // 16) * ((nx + 15) // 16)
        flip = uniform(s, 30, patch) < flip_fraction
        nrm[flip] *= -1.0
        clouds.append(np.stack([x, y, z], axis=1))
        normals.append(nrm)
    xyz = np.concatenate(clouds).astype(np.float32)
    nrm = np.concatenate(normals).astype(np.float32)
    n = xyz.shape[0]
    if nan_points:
        bad = (u64(seed, 40, np.arange(nan_points)) % np.uint64(n)).astype(np.int64)
        xyz[bad, (u64(seed, 41, np.arange(nan_points)) % np.uint64(3)).astype(np.int64)] = np.nan
    if nan_normals:
        bad = (u64(seed, 42, np.arange(nan_normals)) % np.uint64(n)).astype(np.int64)
        nrm[bad, (u64(seed, 43, np.arange(nan_normals)) % np.uint64(3)).astype(np.int64)] = np.nan
    return xyz, nrm


def shuffle_cloud(xyz, nrm, seed):
    """Deterministic permutation (real scans are not stored in raster order)."""
    key = u64(seed, 50, np.arange(xyz.shape[0]))
    perm = np.argsort(key, kind="stable")
    return xyz[perm], nrm[perm]


# ------------------------------------------------------------------------------------------
def saliency_labels(feat, A, B, keep_fraction=0.12):
    """Label 0 (= keypoint, /root/reference/src/main_train_detector.cpp:405-407) for the most
    'non-flat' feature rows, 1 otherwise.  Saliency = mass outside bin 0 summed over annuli."""
    f = feat.reshape(-1, A, B).astype(np.float64)
    sal = (1.0 - f[:, :, 0]).sum(axis=1)
    thr = np.quantile(sal, 1.0 - keep_fraction)
    return np.where(sal >= thr, 0, 1).astype(np.int32)


def train_extra_trees(feat, labels, ntrees=10, max_depth=12, min_samples=4, seed=2,
                      candidates=12, bootstrap_fraction=0.7):
    """Small deterministic extremely-randomised-trees trainer (numpy only).

    Splits are `x[var] <= thr -> left` with float32 thresholds, leaves carry the majority class
    label as their value -- the shape cv::ml::RTrees produces for a 2-class problem.
    """
    feat = np.asarray(feat, dtype=np.float32)
    labels = np.asarray(labels, dtype=np.int32)
    n, F = feat.shape
    root, var, thr, left, right, value, depth, cidx, qual = ([] for _ in range(9))

    def gini_gain(y, mask):
        nl = int(mask.sum())
        nr = y.size - nl
        if nl == 0 or nr == 0:
            return -1.0
        pl = y[mask].mean()
        pr = y[~mask].mean()
        p = y.mean()
        g = lambda q: 2 * q * (1 - q)
        return g(p) - (nl * g(pl) + nr * g(pr)) / y.size

    for t in range(ntrees):
        rng = Rng(seed, 100 + t)
        m = max(1, int(n * bootstrap_fraction))
        sample = np.unique((u64(seed, 200 + t, np.arange(m)) % np.uint64(n)).astype(np.int64))
        root.append(len(var))
        stack = [(sample, 0, -1, False)]
        while stack:
            ids, d, parent, is_right = stack.pop()
            nd = len(var)
            if parent >= 0:
                if is_right:
                    right[parent] = nd
                else:
                    left[parent] = nd
            y = labels[ids]
            maj = int(np.round(y.mean() + 1e-9)) if y.size else 1
            var.append(-1); thr.append(np.float32(0)); left.append(-1); right.append(-1)
            value.append(float(maj)); depth.append(d); cidx.append(maj); qual.append(0.0)
            if d >= max_depth or ids.size < min_samples or y.min() == y.max():
                continue
            best = (-1.0, None, None)
            for _ in range(candidates):
                v = rng.randint(F)
                col = feat[ids, v]
                lo, hi = float(col.min()), float(col.max())
                if not lo < hi:
                    continue
                th = np.float32(lo + (hi - lo) * rng.next())
                if not (th >= lo and th < hi):
                    continue
                gain = gini_gain(y, col <= th)
                if gain > best[0]:
                    best = (gain, v, th)
            if best[1] is None or best[0] <= 0.0:
                continue
            gain, v, th = best
            var[nd] = v
            thr[nd] = th
            qual[nd] = gain * ids.size
            mask = feat[ids, v] <= th
            # push right first so the left subtree is numbered first (pre-order, left first)
            stack.append((ids[~mask], d + 1, nd, True))
            stack.append((ids[mask], d + 1, nd, False))
    return ForestArrays(root, var, thr, left, right, value, F, depth, cidx, qual)


def random_forest(F, ntrees=100, max_depth=25, seed=3, target_nodes_per_tree=20000,
                  thr_lo=0.0, thr_hi=1.0, feat=None):
    """Untrained seeded trees for forest-bound stress (cfg5): grows each tree breadth-limited
    to about `target_nodes_per_tree` nodes, thresholds drawn from sample rows of `feat` when
    given (so both branches are actually taken), otherwise uniform in [thr_lo, thr_hi]."""
    root, var, thr, left, right, value, depth = ([] for _ in range(7))
    for t in range(ntrees):
        rng = Rng(seed, 300 + t)
        root.append(len(var))
        budget = target_nodes_per_tree
        stack = [(0, -1, False)]
        while stack:
            d, parent, is_right = stack.pop()
            nd = len(var)
            if parent >= 0:
                if is_right:
                    right[parent] = nd
                else:
                    left[parent] = nd
            p_split = 0.0 if d >= max_depth else (1.0 if d < 6 else 0.88)
            split = budget > 2 and rng.next() < p_split
            var.append(-1); thr.append(np.float32(0)); left.append(-1); right.append(-1)
            value.append(0.0); depth.append(d)
            if not split:
                value[nd] = float(rng.next() < 0.8)
                continue
            budget -= 2
            v = rng.randint(F)
            if feat is not None:
                th = np.float32(feat[rng.randint(feat.shape[0]), v])
            else:
                th = np.float32(thr_lo + (thr_hi - thr_lo) * rng.next())
            var[nd] = v
            thr[nd] = th
            stack.append((d + 1, nd, True))
            stack.append((d + 1, nd, False))
    return ForestArrays(root, var, thr, left, right, value, F, depth)
