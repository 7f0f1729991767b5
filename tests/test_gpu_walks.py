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


def _cheff_detector(kpl, gold):
    det = kpl.KeypointLearningDetector()
    det.setNonMaxima(True)
    det.setNonMaxRadius(float(gold["r_nms"]))
    det.setNonMaximaDrawsRemove(False)
    det.setPredictionThreshold(float(gold["thr"]))
    det.setRadiusSearch(float(gold["r_feat"]))
    assert det.loadForest(FOREST), det.lastError()
    return det


def test_the_automatic_walk_follows_what_the_handle_measured(kpl, cases, gold):
    """device entry points, first call: nothing known -> every lane for itself, two lanes per point; after it the handle
    knows ~2 300 neighbors per point on this radius -> the two-pass walk (four lanes per point); a small radius on the same
    handle makes the hint stale -> back to the default, and its own measurement keeps it there."""
    import torch
    det = _cheff_detector(kpl, gold)
    dev = torch.device("cuda", 0)
    n = len(gold["xyz"])
    dx, dn = torch.from_numpy(gold["xyz"]).to(dev), torch.from_numpy(gold["nrm"]).to(dev)
    ds = torch.empty(n, dtype=torch.float32, device=dev)
    dk = torch.zeros(n + 1, dtype=torch.int32, device=dev)
    det.bindCloudDevice(dx.data_ptr(), 12, dn.data_ptr(), 12, n)
    assert det.getFeatureWalk() == (kpl.WALK_LANES, 2, -1.0)

    def run():
        det.computeDevice(ds.data_ptr(), dk[1:].data_ptr(), n, dk[0:1].data_ptr())
        assert det.syncStatus(None) == kpl.OK
        t = det.getTiming()
        return ds.cpu().numpy(), dk[1:1 + int(dk[0].item())].cpu().numpy(), (t["walk"], t["lanes_per_point"])
    s1, k1, took1 = run()
    assert took1 == (kpl.WALK_LANES, 2)
    walk, lanes, kf = det.getFeatureWalk()
    assert walk == kpl.WALK_TWO_PASS and lanes == 4 and 1500 < kf < 3500, (walk, lanes, kf)
    s2, k2, took2 = run()                                        # ... through the two-pass walk now
    assert took2 == (kpl.WALK_TWO_PASS, 4)
    assert cases.same_bits(s1, gold["scores_canonical"]) and cases.same_bits(s2, gold["scores_canonical"])
    assert np.array_equal(k1, gold["kp_canonical"]) and np.array_equal(k2, gold["kp_canonical"])
    det.setRadiusSearch(float(gold["r_feat"]) / 5.0)             # ~1/25 of the neighbors: the hint no longer applies
    assert det.getFeatureWalk()[:2] == (kpl.WALK_LANES, 2)
    _, _, took3 = run()
    assert took3 == (kpl.WALK_LANES, 2)
    walk, lanes, kf = det.getFeatureWalk()
    assert walk == kpl.WALK_LANES and lanes == 2 and 20 < kf < 400, (walk, lanes, kf)


@pytest.mark.parametrize("staging", [False, True])
def test_a_first_host_call_estimates_the_neighborhood_from_the_bounding_box(kpl, oracle, cases, gold, staging):
    """a drop-in TestDetector run makes ONE call on a fresh handle: the host entry points have the points in hand and
    estimate the neighbors per point from the view's bounding box -- cheff001 at the reference's default radius takes the
    two-pass walk on its very first call (and gives the golden bits); a 6-mesh-resolution view (~70 neighbors) does not."""
    def first_call(det, xyz, nrm):
        if staging:
            sx, sn = det.hostStaging(len(xyz))
            sx[:], sn[:] = xyz, nrm
            det.computeStaged()
            scores = None
        else:
            det.setInputCloud(xyz)
            det.setNormals(nrm)
            _, scores = det.compute()
        t = det.getTiming()
        return scores, (t["walk"], t["lanes_per_point"])
    det = _cheff_detector(kpl, gold)
    scores, took = first_call(det, gold["xyz"], gold["nrm"])
    assert took == (kpl.WALK_TWO_PASS, 4)
    if scores is not None:
        assert cases.same_bits(scores, gold["scores_canonical"])
    assert np.array_equal(det.getKeypointsIndices(), gold["kp_canonical"])
    A, B = 5, 6
    xyz, nrm = cases.cloud()
    mr = cases.resolution()
    small = kpl.KeypointLearningDetector()
    small.setNAnnulus(A); small.setNBins(B); small.setNonMaxima(True); small.setNonMaxRadius(float(np.float32(4 * mr)))
    small.setNonMaximaDrawsRemove(False); small.setPredictionThreshold(0.6); small.setRadiusSearch(float(np.float32(6 * mr)))
    cases.load_arrays(small, cases.trained_forest(A, B))
    _, took = first_call(small, xyz, nrm)
    assert took == (kpl.WALK_LANES, 2)


def test_short_accept_lists_follow_the_measured_neighborhood(kpl, oracle, cases):
    """the one-kernel walk collects up to 24 accept words per point between two drains until the handle has measured the
    neighborhood: ~70 neighbors per point -> 12 words from the second call on (more resident waves, search and drain of the
    waves out of step), ~125 -> 16, ~155 -> 20; every capacity gives the oracle's bits.  The launch takes the short lists only
    when it has more waves than are resident with 24 words (> 131 k points at two lanes per point): the first case is that big."""
    A, B = 5, 6
    fa = cases.trained_forest(A, B)
    of = cases.oracle_forest(fa)
    for (nx, ny), rmul, words, lo, hi in [((440, 330), 6.0, 12, 40, 80), ((80, 60), 6.0, 12, 40, 80), ((80, 60), 8.0, 16, 80, 140),
                                          ((440, 330), 9.0, 20, 140, 175), ((80, 60), 11.0, 24, 175, 400)]:
        xyz, nrm = cases.cloud(nx, ny)
        mr = cases.resolution(nx, ny)
        r, rn, thr = float(np.float32(rmul * mr)), float(np.float32(4 * mr)), float(np.float32(0.6))
        det = kpl.KeypointLearningDetector()
        det.setNAnnulus(A); det.setNBins(B); det.setNonMaxima(True); det.setNonMaxRadius(rn)
        det.setNonMaximaDrawsRemove(False); det.setPredictionThreshold(thr); det.setRadiusSearch(r)
        cases.load_arrays(det, fa)
        det.setInputCloud(xyz)
        det.setNormals(nrm)
        o_scores, o_kp = oracle.detect(xyz, nrm, A, B, r, rn, thr, of, threads=cases.usable_cores())
        for call in range(2):
            _, scores = det.compute()
            t = det.getTiming()
            assert (t["walk"], t["lanes_per_point"]) == (kpl.WALK_LANES, 2)
            assert t["accept_words"] == (24 if call == 0 else words), (rmul, call, t, det.getFeatureWalk())
            assert cases.same_bits(scores, o_scores)
            assert np.array_equal(det.getKeypointsIndices(), o_kp)
        assert lo < det.getFeatureWalk()[2] <= hi, det.getFeatureWalk()


def test_two_pass_word_list_grows_through_retry(kpl, oracle, cases):
    """a random volume, ~7 500 neighbors and ~30 000 candidates per point -- about 1 000 accept words per point where a
    fresh handle reserves ~500: the device entry point cannot grow the list itself, the first call reports KPL_ERR_RETRY
    through kpl_sync_status (count -1), the second one has room; non-finite points and normals mixed in; then the same view
    through the host entry point on a fresh handle, which retries by itself"""
    import torch
    from tools import synth
    rng = np.random.default_rng(5)
    xyz = rng.uniform(0, 1, size=(60000, 3)).astype(np.float32)
    nrm = rng.normal(size=(60000, 3)).astype(np.float32)
    nrm /= np.linalg.norm(nrm, axis=1, keepdims=True)
    xyz[::997] = np.nan
    nrm[5::811] = np.nan
    A, B, r = 3, 4, 0.36
    fa = synth.random_forest(A * B, ntrees=6, max_depth=8, seed=12, target_nodes_per_tree=120)
    o_scores, _ = oracle.detect(xyz, nrm, A, B, r, 0.0, 0.0, cases.oracle_forest(fa), non_maxima=False, threads=cases.usable_cores())

    def make():
        det = kpl.KeypointLearningDetector()
        det.setNAnnulus(A); det.setNBins(B); det.setNonMaxima(False); det.setNonMaxRadius(0.0)
        det.setPredictionThreshold(0.0); det.setRadiusSearch(r)
        cases.load_arrays(det, fa)
        det.setFeatureWalk(kpl.WALK_TWO_PASS, 4)
        return det
    det = make()
    dev = torch.device("cuda", 0)
    dx, dn = torch.from_numpy(xyz).to(dev), torch.from_numpy(nrm).to(dev)
    ds = torch.empty(len(xyz), dtype=torch.float32, device=dev)
    dk = torch.zeros(len(xyz) + 1, dtype=torch.int32, device=dev)
    det.bindCloudDevice(dx.data_ptr(), 12, dn.data_ptr(), 12, len(xyz))
    det.computeDevice(ds.data_ptr(), dk[1:].data_ptr(), len(xyz), dk[0:1].data_ptr())
    assert det.syncStatus(None) == kpl.ERR_RETRY and int(dk[0].item()) == -1
    assert "accept words" in det.lastError()
    det.computeDevice(ds.data_ptr(), dk[1:].data_ptr(), len(xyz), dk[0:1].data_ptr())
    assert det.syncStatus(None) == kpl.OK
    assert cases.same_bits(ds.cpu().numpy(), o_scores)
    det2 = make()
    det2.setInputCloud(xyz)
    det2.setNormals(nrm)
    _, scores = det2.compute()
    assert cases.same_bits(scores, o_scores)


def test_batch_mixes_the_walks(kpl, oracle, cases):
    """one kpl_compute_batch_device call over views that take different walks (forced): every kernel of the stage skips the
    views of the others; each view's scores and keypoints are the oracle's"""
    import torch
    A, B = 5, 6
    xyz, nrm = cases.cloud()
    mr = cases.resolution()
    fa = cases.trained_forest(A, B)
    of = cases.oracle_forest(fa)
    dev = torch.device("cuda", 0)
    modes = [(kpl.WALK_LANES, 2, 6.0), (kpl.WALK_TWO_PASS, 4, 12.0), (kpl.WALK_LANES, 4, 8.0), (kpl.WALK_TWO_PASS, 2, 6.0)]
    dets, bufs, want = [], [], []
    for walk, lanes, rmul in modes:
        r, rn, thr = float(np.float32(rmul * mr)), float(np.float32(4 * mr)), float(np.float32(0.6))
        det = kpl.KeypointLearningDetector()
        det.setNAnnulus(A); det.setNBins(B); det.setNonMaxima(True); det.setNonMaxRadius(rn)
        det.setNonMaximaDrawsRemove(False); det.setPredictionThreshold(thr); det.setRadiusSearch(r)
        cases.load_arrays(det, fa)
        det.setFeatureWalk(walk, lanes)
        dx, dn = torch.from_numpy(np.array(xyz)).to(dev), torch.from_numpy(np.array(nrm)).to(dev)
        ds = torch.empty(len(xyz), dtype=torch.float32, device=dev)
        dk = torch.zeros(len(xyz) + 1, dtype=torch.int32, device=dev)
        det.bindCloudDevice(dx.data_ptr(), 12, dn.data_ptr(), 12, len(xyz))
        dets.append(det); bufs.append((dx, dn, ds, dk))
        want.append(oracle.detect(xyz, nrm, A, B, r, rn, thr, of))
    for attempt in range(3):
        kpl.compute_batch_device(dets, [b[2].data_ptr() for b in bufs], [b[3][1:].data_ptr() for b in bufs],
                                 [len(xyz)] * len(dets), [b[3][0:1].data_ptr() for b in bufs], None)
        torch.cuda.synchronize()
        rcs = [d.syncStatus(None) for d in dets]
        if kpl.ERR_RETRY not in rcs:
            break
    assert all(rc == kpl.OK for rc in rcs), rcs
    for (dx, dn, ds, dk), (o_sc, o_kp) in zip(bufs, want):
        assert cases.same_bits(ds.cpu().numpy(), o_sc)
        assert np.array_equal(dk[1:1 + int(dk[0].item())].cpu().numpy(), o_kp)
