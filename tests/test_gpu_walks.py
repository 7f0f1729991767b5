"""How the feature kernels WALK a neighborhood is a choice of speed, never of result (kpl_set_feature_walk): every lane
fetching its own candidates or the wave staging the candidates of its points' cell in LDS (large neighborhoods), two or
four lanes per point.  Every combination must give the oracle's bits -- on the reference's default operating point
(cheff001, ~2 300 neighbors per point, the regime the two-pass walk exists for), on small neighborhoods (where the groups of
a wave lie in several cells), with non-finite points and normals, and on a view in several overlapping layers.  And the
automatic choice must come from the handle's own earlier calls."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FOREST = os.path.join(ROOT, "data", "forests", "cheff_a5b10_t10.yaml.gz")
WALKS = [("lanes", 2), ("lanes", 4), ("twopass", 2), ("twopass", 4)]


def _walk(kpl, name):
    return kpl.WALK_TWO_PASS if name == "twopass" else kpl.WALK_LANES


@pytest.fixture(scope="module")
def gold():
    return np.load(os.path.join(ROOT, "tests", "golden", "cheff001.npz"))


@pytest.mark.parametrize("walk,lanes", WALKS)
def test_every_walk_gives_the_oracles_bits_at_the_default_operating_point(kpl, cases, gold, walk, lanes):
    det = kpl.KeypointLearningDetector()
    det.setNonMaxima(True)
    det.setNonMaxRadius(float(gold["r_nms"]))
    det.setNonMaximaDrawsRemove(False)
    det.setPredictionThreshold(float(gold["thr"]))
    det.setRadiusSearch(float(gold["r_feat"]))
    assert det.loadForest(FOREST), det.lastError()
    det.setFeatureWalk(_walk(kpl, walk), lanes)
    assert det.getFeatureWalk()[:2] == (_walk(kpl, walk), lanes)
    det.setInputCloud(gold["xyz"])
    det.setNormals(gold["nrm"])
    _, scores = det.compute()
    assert cases.same_bits(scores, gold["scores_canonical"])
    assert np.array_equal(det.getKeypointsIndices(), gold["kp_canonical"])


@pytest.mark.parametrize("walk,lanes", WALKS)
@pytest.mark.parametrize("case", ["small_neighborhoods", "nonfinite", "layers", "huge_radius"])
def test_every_walk_on_awkward_views(kpl, oracle, cases, walk, lanes, case):
    """waves whose points lie in many cells (6 mr: ~36 points per cell), points / normals that are not finite, a view in
    several layers (boxes of three cell layers), and a radius that puts the whole view into a handful of cells"""
    A, B = 5, 6
    if case == "nonfinite":
        xyz, nrm = cases.cloud(80, 60, seed=3, nan_points=37, nan_normals=53)
    elif case == "layers":
        xyz, nrm = cases.cloud(70, 50, seed=5, layers=3)
    else:
        xyz, nrm = cases.cloud()
    mr = cases.resolution()
    rmul = 21.0 if case == "huge_radius" else 6.0
    r, rn, thr = float(np.float32(rmul * mr)), float(np.float32(4 * mr)), float(np.float32(0.6))
    fa = cases.trained_forest(A, B)
    det = kpl.KeypointLearningDetector()
    det.setNAnnulus(A); det.setNBins(B); det.setNonMaxima(True); det.setNonMaxRadius(rn)
    det.setNonMaximaDrawsRemove(False); det.setPredictionThreshold(thr); det.setRadiusSearch(r)
    cases.load_arrays(det, fa)
    det.setFeatureWalk(_walk(kpl, walk), lanes)
    det.setInputCloud(xyz)
    det.setNormals(nrm)
    _, scores = det.compute()
    o_scores, o_kp = oracle.detect(xyz, nrm, A, B, r, rn, thr, cases.oracle_forest(fa))
    assert cases.same_bits(scores, o_scores)
    assert np.array_equal(det.getKeypointsIndices(), o_kp)


def test_the_automatic_walk_follows_what_the_handle_measured(kpl, cases, gold):
    """first call: nothing known -> every lane for itself, two lanes per point; after it the handle knows ~2 300 neighbors
    per point on this radius -> the two-pass walk (four lanes per point: a 63 k-point view); a small radius on the same
    handle makes the hint stale -> back to the default, and its own measurement keeps it there."""
    det = kpl.KeypointLearningDetector()
    det.setNonMaxima(True)
    det.setNonMaxRadius(float(gold["r_nms"]))
    det.setNonMaximaDrawsRemove(False)
    det.setPredictionThreshold(float(gold["thr"]))
    det.setRadiusSearch(float(gold["r_feat"]))
    assert det.loadForest(FOREST), det.lastError()
    det.setInputCloud(gold["xyz"])
    det.setNormals(gold["nrm"])
    assert det.getFeatureWalk() == (kpl.WALK_LANES, 2, -1.0)
    _, s1 = det.compute()
    walk, lanes, kf = det.getFeatureWalk()
    assert walk == kpl.WALK_TWO_PASS and lanes == 4 and 1500 < kf < 3500, (walk, lanes, kf)
    _, s2 = det.compute()                                        # ... through the two-pass walk now
    assert cases.same_bits(s1, gold["scores_canonical"]) and cases.same_bits(s2, gold["scores_canonical"])
    assert np.array_equal(det.getKeypointsIndices(), gold["kp_canonical"])
    det.setRadiusSearch(float(gold["r_feat"]) / 5.0)             # ~1/25 of the neighbors: the hint no longer applies
    assert det.getFeatureWalk()[:2] == (kpl.WALK_LANES, 2)
    det.compute()
    walk, lanes, kf = det.getFeatureWalk()
    assert walk == kpl.WALK_LANES and lanes == 2 and 20 < kf < 400, (walk, lanes, kf)
