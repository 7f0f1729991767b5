"""NMS of the oracle on hand-built score fields
(/root/reference/include/impl/KeypointLearning.hpp:197-256)."""
import os

import numpy as np

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def line(n=21, step=1.0):
    xyz = np.zeros((n, 3), dtype=np.float32)
    xyz[:, 0] = np.arange(n) * step
    return xyz


def test_strict_maximum(oracle):
    xyz = line()
    s = np.full(21, 0.5, dtype=np.float32)
    s[10] = 0.9
    kp = oracle.Grid(xyz, 2.5).nms(s, 2.5, 0.85)
    assert kp.tolist() == [10]
    # below threshold: nothing
    assert oracle.Grid(xyz, 2.5).nms(s, 2.5, 0.95).tolist() == []


def test_plateau_survives_without_draw_removal(oracle):
    xyz = line()
    s = np.full(21, 0.9, dtype=np.float32)
    kp = oracle.Grid(xyz, 2.5).nms(s, 2.5, 0.85, draws_remove=False)
    assert kp.tolist() == list(range(21))           # every member of an equal-score plateau


def test_strictly_greater_neighbor_suppresses(oracle):
    xyz = line()
    s = np.linspace(0.86, 0.99, 21).astype(np.float32)   # increasing: only the last is a maximum
    assert oracle.Grid(xyz, 1.5).nms(s, 1.5, 0.85).tolist() == [20]
    # radius is strict: a neighbor exactly at the radius does not count
    assert oracle.Grid(xyz, 1.0).nms(s, 1.0, 0.85).tolist() == list(range(21))


def test_threshold_is_a_double_compare_of_a_float(oracle):
    xyz = line(3)
    thr = float(np.float32(0.85))                  # TestDetector: float 0.85f promoted to double
    s = np.array([np.float32(0.85), np.nextafter(np.float32(0.85), np.float32(0)), 0.1], dtype=np.float32)
    assert oracle.Grid(xyz, 0.5).nms(s, 0.5, thr).tolist() == [0]
    assert oracle.Grid(xyz, 0.5).nms(s, 0.5, 0.85).tolist() == [0]     # double 0.85 < 0.85f
    # a double threshold just above 0.85f must reject 0.85f (it would pass if thr were rounded to float)
    assert oracle.Grid(xyz, 0.5).nms(s, 0.5, thr + 1e-9).tolist() == []


def test_nan_scores_never_keypoints_and_never_suppress(oracle):
    xyz = line(5)
    s = np.array([0.9, np.nan, 0.9, np.nan, 0.2], dtype=np.float32)
    assert oracle.Grid(xyz, 1.5).nms(s, 1.5, 0.5).tolist() == [0, 2]


def test_draw_removal_greedy(oracle):
    # plateau of 4 in a row, spacing 1, nms radius 1.5 (each sees its direct neighbors)
    xyz = line(4)
    s = np.full(4, 0.9, dtype=np.float32)
    g = oracle.Grid(xyz, 1.5)
    # draws threshold 1.2: idx0 survives and marks 1; idx1 skipped; idx2 survives (draw 3 within
    # 1.2; draw 1 too -> both appended to skip); idx3 skipped
    assert g.nms(s, 1.5, 0.5, draws_remove=True, draws_threshold=1.2).tolist() == [0, 2]
    # draws but none within the threshold => dropped (hpp:236-249)
    assert g.nms(s, 1.5, 0.5, draws_remove=True, draws_threshold=0.5).tolist() == []
    # a strict maximum without draws is emitted as usual
    s2 = np.array([0.6, 0.9, 0.6, 0.6], dtype=np.float32)
    assert g.nms(s2, 1.5, 0.5, draws_remove=True, draws_threshold=0.5).tolist() == [1]


def test_draw_removal_is_the_first_maximal_independent_set(oracle):
    """The reference's loop over the maxima with draws (hpp:231-250: skip list, ascending index) restated as the rule the
    device evaluates out of order (kernels.hip, draws pass): a listed maximum is dropped iff a listed maximum of LOWER index
    that is itself not dropped has it within the draws threshold, and one without any draw within the threshold does not
    survive either.  Random clouds with few score levels (many plateaus), both computed from scratch in numpy."""
    rng = np.random.default_rng(2026)
    for trial in range(12):
        n = int(rng.integers(40, 400))
        xyz = (rng.random((n, 3)) * np.array([12.0, 9.0, 1.0])).astype(np.float32)
        s = (rng.integers(0, 4, size=n) / 4.0).astype(np.float32)
        r_nms = float(np.float32(rng.uniform(1.0, 2.5)))
        dthr = float(np.float32(rng.uniform(0.3, 3.0)))
        thr = 0.3
        kp = oracle.Grid(xyz, r_nms).nms(s, r_nms, thr, draws_remove=True, draws_threshold=dthr)
        # the same from the definition
        d2 = np.zeros((n, n), np.float32)
        for k in range(3):                                   # FLANN's L2_Simple order: ((dx*dx) + dy*dy) + dz*dz
            dk = xyz[:, None, k] - xyz[None, :, k]
            d2 = d2 + dk * dk
        near = d2 < np.float32(r_nms) * np.float32(r_nms)
        np.fill_diagonal(near, False)
        cand = s.astype(np.float64) >= thr
        is_max = cand & ~np.array([np.any(near[i] & (s > s[i])) for i in range(n)])
        draw = near & (s[:, None] == s[None, :])
        listed = is_max & draw.any(axis=1)
        dx, dy, dz = (xyz[:, None, k] - xyz[None, :, k] for k in range(3))
        dist = np.sqrt(dx * dx + (dy * dy + dz * dz))        # hpp:239 (a - b).norm()
        within = draw & (dist < np.float32(dthr))
        dropped = np.zeros(n, bool)
        keep = []
        for i in range(n):                                   # ascending index = the reference's order
            if is_max[i] and not listed[i]:
                keep.append(i)                               # a strict maximum without draws
            if not listed[i]:
                continue
            lower = np.nonzero(within[i, :i] & listed[:i] & ~dropped[:i])[0]
            lower = [j for j in lower if within[j].any()]    # j itself survived (it has a draw within the threshold: i)
            dropped[i] = len(lower) > 0
            if not dropped[i] and within[i].any():
                keep.append(i)
            elif not dropped[i]:
                dropped[i] = True                            # no draw within the threshold: does not survive, marks nobody
        assert kp.tolist() == sorted(keep), (trial, n, r_nms, dthr)


def test_committed_fixture_end_to_end(oracle, cases):
    from tools import forest_yaml
    z = np.load(os.path.join(GOLD, "small_case.npz"))
    fa = forest_yaml.load_forest(os.path.join(GOLD, "small_forest.yaml.gz"))
    of = cases.oracle_forest(fa)
    for thr in (0.0, 0.5, 0.85):
        for dr in (0, 1):
            sc, kp = oracle.detect(z["xyz"], z["nrm"], 5, 6, float(z["r_feat"]), float(z["r_nms"]),
                                   float(np.float32(thr)), of, draws_remove=bool(dr),
                                   draws_threshold=float(z["draws_threshold"]))
            assert np.array_equal(kp, z["kp_thr%03d_dr%d" % (int(thr * 100), dr)])
            assert cases.same_bits(sc, z["scores"])


def test_cheff_fixture(oracle, cases):
    """config 1 anchor: a real scan of the reference's data set + the cfg forest."""
    from tools import forest_yaml
    z = np.load(os.path.join(GOLD, "cheff000.npz"))
    root = os.path.dirname(os.path.dirname(__file__))
    fa = forest_yaml.load_forest(os.path.join(root, "data", "forests", "synth200k_a5b6_t10.yaml.gz"))
    sc, kp = oracle.detect(z["xyz"], z["nrm"], 5, 6, float(z["r_feat"]), float(z["r_nms"]), float(z["thr"]),
                           cases.oracle_forest(fa), threads=4)
    assert np.array_equal(kp, z["kp"]) and cases.same_bits(sc, z["scores"])
