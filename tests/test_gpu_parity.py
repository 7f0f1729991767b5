"""GPU parity: libkpl (HIP, through the C-ABI) against the CPU oracle on the same seeded inputs.
Features bit-exact, scores bit-exact (tolerance allowed by the north star: 1e-5), keypoint
index lists identical."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def make_det(kpl, A, B, r_feat, r_nms, thr, fa=None, draws_remove=False):
    det = kpl.KeypointLearningDetector()
    det.setNAnnulus(A)
    det.setNBins(B)
    det.setNonMaxima(True)
    det.setNonMaxRadius(r_nms)
    det.setNonMaximaDrawsRemove(draws_remove)
    det.setPredictionThreshold(thr)
    det.setRadiusSearch(r_feat)
    if fa is not None:
        from tests.helpers import load_arrays
        load_arrays(det, fa)
    return det


@pytest.mark.parametrize("A,B", [(5, 6), (5, 10), (8, 10), (1, 1), (3, 2)])
def test_features_bit_exact(kpl, oracle, cases, A, B):
    xyz, nrm = cases.cloud()
    mr = cases.resolution()
    r = float(np.float32(6 * mr))
    det = make_det(kpl, A, B, r, 0.0, 0.5)
    det.setInputCloud(xyz)
    det.setNormals(nrm)
    q = np.arange(len(xyz), dtype=np.int32)
    got = det.computePointsForTrainingFeatures(q)
    want = oracle.Grid(xyz, r).features(nrm, A, B, r, q)
    assert cases.same_bits(got, want)
    # sparse, unordered, repeated query list (the training entry point)
    q2 = np.array([5, 0, len(xyz) - 1, 5, 17, 1234], dtype=np.int32)
    assert cases.same_bits(det.computePointsForTrainingFeatures(q2), want[q2])


@pytest.mark.parametrize("thr", [0.0, 0.5, 0.85, 1.0])
def test_detect_matches_oracle(kpl, oracle, cases, thr):
    A, B = 5, 6
    xyz, nrm = cases.cloud()
    mr = cases.resolution()
    r, rn = float(np.float32(6 * mr)), float(np.float32(4 * mr))
    fa = cases.trained_forest(A, B)
    det = make_det(kpl, A, B, r, rn, float(np.float32(thr)), fa)
    det.setInputCloud(xyz)
    det.setNormals(nrm)
    kp_xyzi, scores = det.compute()
    o_scores, o_kp = oracle.detect(xyz, nrm, A, B, r, rn, float(np.float32(thr)), cases.oracle_forest(fa))
    assert cases.same_bits(scores, o_scores)
    assert np.array_equal(det.getKeypointsIndices(), o_kp)
    assert len(o_kp) > 0
    assert cases.same_bits(kp_xyzi[:, 3], o_scores[o_kp])
    # run twice: deterministic
    det.compute()
    assert np.array_equal(det.getKeypointsIndices(), o_kp)


def test_non_finite_points_and_normals(kpl, oracle, cases):
    A, B = 5, 6
    xyz, nrm = cases.cloud(nan_points=40, nan_normals=60)
    mr = cases.resolution()
    r, rn = 6 * mr, 4 * mr
    fa = cases.trained_forest(A, B)
    det = make_det(kpl, A, B, r, rn, 0.5, fa)
    det.setInputCloud(xyz)
    det.setNormals(nrm)
    _, scores = det.compute()
    o_scores, o_kp = oracle.detect(xyz, nrm, A, B, r, rn, 0.5, cases.oracle_forest(fa))
    assert np.isnan(o_scores).sum() >= 60
    assert cases.same_bits(scores, o_scores)
    assert np.array_equal(det.getKeypointsIndices(), o_kp)


@pytest.mark.parametrize("rf,rn", [(4.0, 4.0), (10.0, 4.0), (3.0, 7.5), (6.0, 0.0)])
def test_radius_combinations(kpl, oracle, cases, rf, rn):
    A, B = 5, 6
    xyz, nrm = cases.cloud()
    mr = cases.resolution()
    fa = cases.trained_forest(A, B)
    det = make_det(kpl, A, B, rf * mr, rn * mr, 0.6, fa)
    det.setInputCloud(xyz)
    det.setNormals(nrm)
    _, scores = det.compute()
    o_scores, o_kp = oracle.detect(xyz, nrm, A, B, rf * mr, rn * mr, 0.6, cases.oracle_forest(fa))
    assert cases.same_bits(scores, o_scores)
    assert np.array_equal(det.getKeypointsIndices(), o_kp)


def test_no_nms_returns_every_scoreable_point(kpl, oracle, cases):
    A, B = 5, 6
    xyz, nrm = cases.cloud(nan_points=10, nan_normals=10)
    mr = cases.resolution()
    fa = cases.trained_forest(A, B)
    det = make_det(kpl, A, B, 6 * mr, 4 * mr, 0.85, fa)
    det.setNonMaxima(False)
    det.setInputCloud(xyz)
    det.setNormals(nrm)
    _, scores = det.compute()
    o_scores, o_kp = oracle.detect(xyz, nrm, A, B, 6 * mr, 4 * mr, 0.85, cases.oracle_forest(fa), non_maxima=False)
    assert np.array_equal(det.getKeypointsIndices(), o_kp)
    assert np.array_equal(o_kp, np.flatnonzero(~np.isnan(o_scores)))


def test_edge_sizes(kpl, oracle, cases):
    A, B = 5, 6
    fa = cases.trained_forest(A, B)
    of = cases.oracle_forest(fa)
    det = make_det(kpl, A, B, 2.0, 1.0, 0.0, fa)
    for n in (0, 1, 2, 63, 64, 65):
        xyz, nrm = cases.cloud()
        xyz, nrm = xyz[:n].copy(), nrm[:n].copy()
        det.setInputCloud(xyz.reshape(-1, 3))
        det.setNormals(nrm.reshape(-1, 3))
        _, scores = det.compute()
        o_scores, o_kp = oracle.detect(xyz, nrm, A, B, 2.0, 1.0, 0.0, of)
        assert cases.same_bits(scores, o_scores), n
        assert np.array_equal(det.getKeypointsIndices(), o_kp), n
    # all points identical (one cell, every distance 0)
    xyz = np.ones((100, 3), dtype=np.float32)
    nrm = np.tile(np.array([[0, 0, 1]], dtype=np.float32), (100, 1))
    det.setInputCloud(xyz)
    det.setNormals(nrm)
    _, scores = det.compute()
    o_scores, o_kp = oracle.detect(xyz, nrm, A, B, 2.0, 1.0, 0.0, of)
    assert cases.same_bits(scores, o_scores)
    assert np.array_equal(det.getKeypointsIndices(), o_kp)


def test_strided_pcl_layouts(kpl, oracle, cases):
    """PointXYZ (16 B) and Normal (32 B) arrays are passed as they are."""
    A, B = 5, 6
    xyz, nrm = cases.cloud()
    mr = cases.resolution()
    fa = cases.trained_forest(A, B)
    p16 = np.full((len(xyz), 4), 1.0, dtype=np.float32)
    p16[:, :3] = xyz
    n32 = np.full((len(xyz), 8), np.nan, dtype=np.float32)   # padding must never be read as data
    n32[:, :3] = nrm
    det = make_det(kpl, A, B, 6 * mr, 4 * mr, 0.85, fa)
    det.setInputCloud(p16)
    det.setNormals(n32)
    _, scores = det.compute()
    o_scores, o_kp = oracle.detect(xyz, nrm, A, B, 6 * mr, 4 * mr, 0.85, cases.oracle_forest(fa))
    assert cases.same_bits(scores, o_scores)
    assert np.array_equal(det.getKeypointsIndices(), o_kp)


def test_yaml_gz_forest_file_equals_arrays(kpl, oracle, cases, tmp_path):
    from tools import forest_yaml
    A, B = 5, 6
    xyz, nrm = cases.cloud()
    mr = cases.resolution()
    fa = cases.trained_forest(A, B)
    path = tmp_path / "forest.yaml.gz"
    forest_yaml.save_forest(fa, str(path))
    det = make_det(kpl, A, B, 6 * mr, 4 * mr, 0.85)
    assert det.loadForest(str(path))
    info = det.forestInfo()
    assert info["ntrees"] == fa.ntrees and info["var_count"] == A * B and info["nnodes"] == fa.nnodes
    det.setInputCloud(xyz)
    det.setNormals(nrm)
    _, scores = det.compute()
    o_scores, o_kp = oracle.detect(xyz, nrm, A, B, 6 * mr, 4 * mr, 0.85, cases.oracle_forest(fa))
    assert cases.same_bits(scores, o_scores)
    assert np.array_equal(det.getKeypointsIndices(), o_kp)


def test_errors(kpl, cases):
    A, B = 5, 6
    xyz, nrm = cases.cloud()
    fa = cases.trained_forest(A, B)
    det = make_det(kpl, A, B, 3.0, 2.0, 0.5)
    det.setInputCloud(xyz)
    det.setNormals(nrm)
    with pytest.raises(kpl.KplError) as e:
        det.compute()
    assert e.value.status == kpl.ERR_NO_FOREST
    cases.load_arrays(det, fa)
    det.setNBins(10)
    with pytest.raises(kpl.KplError) as e:
        det.compute()
    assert e.value.status == kpl.ERR_VAR_COUNT
    det.setNBins(B)
    det.setRadiusSearch(0.0)
    with pytest.raises(kpl.KplError) as e:
        det.compute()
    assert e.value.status == kpl.ERR_INVALID_ARG
    det.setRadiusSearch(1e-6)
    with pytest.raises(kpl.KplError) as e:
        det.compute()
    assert e.value.status == kpl.ERR_GRID_TOO_LARGE
    assert not det.loadForest("/nonexistent/forest.yaml.gz")


@pytest.mark.parametrize("thr,dthr_mul", [(0.0, 2.0), (0.5, 1.2), (0.85, 3.0), (0.5, 0.0)])
def test_draws_remove_greedy_pass(kpl, oracle, cases, thr, dthr_mul):
    """non_maxima_draws_remove = true (the class default): order-dependent greedy pass."""
    A, B = 5, 6
    xyz, nrm = cases.cloud()
    mr = cases.resolution()
    r, rn = float(np.float32(6 * mr)), float(np.float32(4 * mr))
    dthr = float(np.float32(dthr_mul * mr))
    fa = cases.trained_forest(A, B)
    det = make_det(kpl, A, B, r, rn, float(np.float32(thr)), fa, draws_remove=True)
    det.setNonMaximaDrawsThreshold(dthr)
    det.setInputCloud(xyz)
    det.setNormals(nrm)
    _, scores = det.compute()
    o_scores, o_kp = oracle.detect(xyz, nrm, A, B, r, rn, float(np.float32(thr)), cases.oracle_forest(fa),
                                   draws_remove=True, draws_threshold=dthr)
    assert cases.same_bits(scores, o_scores)
    assert np.array_equal(det.getKeypointsIndices(), o_kp)
    # switching the mode off again on the same handle gives the plain predicate result
    det.setNonMaximaDrawsRemove(False)
    det.compute()
    _, kp_plain = oracle.detect(xyz, nrm, A, B, r, rn, float(np.float32(thr)), cases.oracle_forest(fa))
    assert np.array_equal(det.getKeypointsIndices(), kp_plain)
    assert len(kp_plain) >= len(o_kp)


def test_draws_remove_plateau_line(kpl, oracle, cases):
    """hand-built plateau: 4 collinear points with equal scores (constant forest)."""
    import numpy as np
    xyz = np.zeros((4, 3), dtype=np.float32)
    xyz[:, 0] = np.arange(4)
    nrm = np.tile(np.array([[0, 0, 1]], dtype=np.float32), (4, 1))
    from tools.forest_yaml import ForestArrays
    fa = ForestArrays([0], [-1], [0.0], [-1], [-1], [0.0], 30)        # one leaf: every score = 1
    det = make_det(kpl, 5, 6, 1.5, 1.5, 0.5, fa, draws_remove=True)
    det.setInputCloud(xyz)
    det.setNormals(nrm)
    for dthr, want in ((1.2, [0, 2]), (0.5, []), (5.0, [0])):
        det.setNonMaximaDrawsThreshold(dthr)
        det.compute()
        _, o_kp = oracle.detect(xyz, nrm, 5, 6, 1.5, 1.5, 0.5, cases.oracle_forest(fa), draws_remove=True,
                                draws_threshold=dthr)
        assert det.getKeypointsIndices().tolist() == o_kp.tolist()


@pytest.mark.parametrize("order", ["scan", "shuffled"])
def test_draws_remove_on_a_constant_score_plane(kpl, oracle, cases, order):
    """Every point a maximum with draws (a constant forest: one plateau over the whole view): the greedy pass is then a
    chain through the list -- in scan order every entry waits for its left neighbor and for the row above, so the parallel
    rounds of the draws pass decide a front at a time and the sequential rest takes the bulk; in shuffled order the
    rounds take nearly all of it.  Same survivors as the reference's loop in both orders."""
    from tools import synth
    from tools.forest_yaml import ForestArrays
    xyz, nrm = synth.make_cloud(120, 90, seed=31)
    if order == "shuffled":
        xyz, nrm = synth.shuffle_cloud(xyz, nrm, 77)
    mr = oracle.cloud_resolution(xyz)
    A, B = 5, 6
    r, rn = float(np.float32(6 * mr)), float(np.float32(4 * mr))
    fa = ForestArrays([0], [-1], [0.0], [-1], [-1], [0.0], A * B)         # one leaf: every score = 1
    det = make_det(kpl, A, B, r, rn, 0.5, fa, draws_remove=True)
    det.setInputCloud(xyz)
    det.setNormals(nrm)
    for dthr_mul in (1.5, 2.5, 4.0):
        dthr = float(np.float32(dthr_mul * mr))
        det.setNonMaximaDrawsThreshold(dthr)
        det.compute()
        _, o_kp = oracle.detect(xyz, nrm, A, B, r, rn, 0.5, cases.oracle_forest(fa), draws_remove=True, draws_threshold=dthr)
        assert np.array_equal(det.getKeypointsIndices(), o_kp)
        assert 0 < len(o_kp) < len(xyz) // 2
    # a wide NMS radius and draw threshold: ~75 listed maxima of lower index within the threshold of every entry -- more
    # than the 32 an adjacency row holds, so the rounds and the pipelined rest sweep the neighborhood instead (states
    # through LDS by list position)
    rn7, dthr7 = float(np.float32(7 * mr)), float(np.float32(6.5 * mr))
    det.setNonMaxRadius(rn7)
    det.setNonMaximaDrawsThreshold(dthr7)
    det.compute()
    _, o_kp = oracle.detect(xyz, nrm, A, B, r, rn7, 0.5, cases.oracle_forest(fa), draws_remove=True, draws_threshold=dthr7)
    assert np.array_equal(det.getKeypointsIndices(), o_kp) and 0 < len(o_kp) < len(xyz) // 20


def test_cloud_resolution_bit_exact(kpl, oracle, cases):
    """kpl_cloud_resolution = computeCloudResolution (point_cloud_utilities.hpp:120-151)."""
    det = kpl.KeypointLearningDetector()
    for nan in (0, 30):
        xyz, _ = cases.cloud(nan_points=nan)
        assert det.cloudResolution(xyz) == oracle.cloud_resolution(xyz)
    p16 = np.zeros((len(xyz), 4), dtype=np.float32)
    p16[:, :3] = xyz
    assert det.cloudResolution(p16) == oracle.cloud_resolution(xyz)
    assert det.cloudResolution(xyz[:1]) == 0.0 and det.cloudResolution(xyz[:0].reshape(0, 3)) == 0.0
    two = np.array([[0, 0, 0], [3, 4, 0]], dtype=np.float32)
    assert det.cloudResolution(two) == 5.0


def test_cloud_resolution_sum_order_fallback(kpl, oracle, cases):
    """The device adds the distances in parallel only when that provably equals the reference's sequential
    double sum; a cloud whose distances span ~50 binades (a nearly coincident pair next to far-apart points)
    must take the sequential fallback and still match bit for bit."""
    det = kpl.KeypointLearningDetector()
    rng = np.random.default_rng(9)
    xyz = (rng.uniform(0, 1, size=(3000, 3)) * np.float32(3.0e7)).astype(np.float32)
    xyz[1] = xyz[0] + np.float32([0, 0, 2.0])         # ulp at 3e7 is 2: the closest representable pair
    tiny = rng.uniform(0, 1, size=(500, 3)).astype(np.float32) * np.float32(1e-9)
    both = np.concatenate([xyz, tiny]).astype(np.float32)
    expect = oracle.cloud_resolution(both)
    assert det.cloudResolution(both) == expect
    assert det.cloudResolution(both[::-1].copy()) == oracle.cloud_resolution(both[::-1].copy())


def test_cell_tables_grow_on_demand(kpl, oracle, cases):
    """A sparse view needs far more grid cells than the initial tables hold: the host entry point retries by itself,
    the asynchronous one reports KPL_ERR_RETRY once (count = -1) and succeeds on the next call."""
    import torch
    rng = np.random.default_rng(12)
    xyz = rng.uniform(0, 150, size=(3000, 3)).astype(np.float32)          # ~150^3 / 1.5^3 = 1 M cells
    xyz[:600] = rng.uniform(70, 80, size=(600, 3)).astype(np.float32)     # a dense clump so that features exist
    nrm = rng.normal(size=(3000, 3)).astype(np.float32)
    nrm /= np.linalg.norm(nrm, axis=1, keepdims=True)
    A, B, r, rn, thr = 5, 6, 1.5, 1.0, 0.0
    fa = cases.trained_forest(A, B)
    o_sc, o_kp = oracle.detect(xyz, nrm, A, B, r, rn, thr, cases.oracle_forest(fa))
    det = make_det(kpl, A, B, r, rn, thr, fa)
    det.setInputCloud(xyz); det.setNormals(nrm)
    _, sc = det.compute()                                                  # kpl_detect: internal retry
    assert cases.same_bits(sc, o_sc) and np.array_equal(det.getKeypointsIndices(), o_kp)
    det2 = make_det(kpl, A, B, r, rn, thr, fa)
    dev = torch.device("cuda", 0)
    dx, dn = torch.from_numpy(xyz).to(dev), torch.from_numpy(nrm).to(dev)
    ds = torch.empty(len(xyz), dtype=torch.float32, device=dev)
    dk = torch.zeros(len(xyz) + 1, dtype=torch.int32, device=dev)
    det2.bindCloudDevice(dx.data_ptr(), 12, dn.data_ptr(), 12, len(xyz))
    det2.computeDevice(ds.data_ptr(), dk[1:].data_ptr(), len(xyz), dk[0:1].data_ptr(), None)
    assert det2.syncStatus(None) == kpl.ERR_RETRY and int(dk[0].item()) == -1
    det2.computeDevice(ds.data_ptr(), dk[1:].data_ptr(), len(xyz), dk[0:1].data_ptr(), None)
    assert det2.syncStatus(None) == kpl.OK
    assert cases.same_bits(ds.cpu().numpy(), o_sc)
    assert np.array_equal(dk[1:1 + int(dk[0].item())].cpu().numpy(), o_kp)


def test_far_from_the_origin(kpl, oracle, cases):
    """coordinates around 1e5 with unit spacing (a geo-referenced scan): float cell arithmetic far from zero"""
    xyz, nrm = cases.cloud(70, 50, seed=21)
    xyz = (xyz + np.float32([123456.0, -98765.0, 4321.0])).astype(np.float32)
    A, B = 5, 6
    fa = cases.trained_forest(A, B)
    mr = oracle.cloud_resolution(xyz)
    r, rn = float(np.float32(6 * mr)), float(np.float32(4 * mr))
    det = make_det(kpl, A, B, r, rn, 0.5, fa)
    det.setInputCloud(xyz); det.setNormals(nrm)
    _, sc = det.compute()
    o_sc, o_kp = oracle.detect(xyz, nrm, A, B, r, rn, 0.5, cases.oracle_forest(fa))
    assert cases.same_bits(sc, o_sc) and np.array_equal(det.getKeypointsIndices(), o_kp)
    assert det.cloudResolution(xyz) == mr
    n1, c1 = det.estimateNormals(xyz, k=10)
    o1, oc1 = oracle.estimate_normals(xyz, k=10)
    assert cases.same_bits(n1, o1) and cases.same_bits(c1, oc1)


def test_largest_histogram(kpl, oracle, cases):
    """n_annulus * n_bins = 255 is the documented limit (one byte of the packed forest node names the variable;
    the wave's histogram is then 65 280 B of LDS); 256 is refused."""
    xyz, nrm = cases.cloud(50, 40, seed=4)
    mr = oracle.cloud_resolution(xyz)
    r = float(np.float32(6 * mr))
    det = kpl.KeypointLearningDetector()
    det.setNAnnulus(15); det.setNBins(17); det.setRadiusSearch(r); det.setNonMaxRadius(1.0)
    det.setInputCloud(xyz); det.setNormals(nrm)
    q = np.arange(0, len(xyz), 9, dtype=np.int32)
    feat = det.computePointsForTrainingFeatures(q)
    assert cases.same_bits(feat, oracle.Grid(xyz, r).features(nrm, 15, 17, r, q))
    fa = cases.trained_forest(15, 17, ntrees=3, max_depth=5)
    cases.load_arrays(det, fa)
    _, sc = det.compute()
    o_sc, o_kp = oracle.detect(xyz, nrm, 15, 17, r, 1.0, 0.5, cases.oracle_forest(fa))
    assert cases.same_bits(sc, o_sc) and np.array_equal(det.getKeypointsIndices(), o_kp)
    det.setNAnnulus(16); det.setNBins(16)
    with pytest.raises(kpl.KplError) as e:
        det.computePointsForTrainingFeatures(q)
    assert e.value.status == kpl.ERR_UNSUPPORTED


def test_randomised_soak():
    """tools/fuzz_parity.py for a few seconds (random clouds, shapes, forests, NMS modes, normals, resolution):
    the long runs (tens of thousands of cases) are recorded in BASELINE.md."""
    import subprocess
    import sys
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "tools", "fuzz_parity.py"), "8", "5"], capture_output=True,
                         text=True, timeout=300, cwd=root)
    assert out.returncode == 0 and "all bit-exact" in out.stdout, out.stdout + out.stderr


@pytest.mark.parametrize("ntrees,nodes_per_tree", [(23, 60), (37, 900)])
def test_fractional_leaf_values_are_summed_in_tree_order(kpl, oracle, cases, ntrees, nodes_per_tree):
    """A forest whose leaf values are not small integers (a regression forest): the double sum of
    cv::ml::RTrees::predict(PREDICT_SUM) depends on the order of the trees, so the device walks them
    in step (kernels.hip forest_sum) instead of letting every lane run ahead (forest_sum_any_order);
    the second shape is larger than the part of a forest that is staged in LDS."""
    import copy
    from tools import synth
    A, B = 5, 6
    xyz, nrm = cases.cloud()
    mr = cases.resolution()
    r, rn = float(np.float32(6 * mr)), float(np.float32(4 * mr))
    feat = oracle.Grid(xyz, r).features(nrm, A, B, r, np.arange(0, len(xyz), 5, dtype=np.int32))
    fa = copy.deepcopy(synth.random_forest(A * B, ntrees=ntrees, max_depth=14, seed=11,
                                           target_nodes_per_tree=nodes_per_tree, feat=feat))
    rng = np.random.default_rng(5)
    value = np.asarray(fa.value, dtype=np.float64).copy()
    leaves = np.asarray(fa.var) < 0
    value[leaves] = rng.uniform(-3.0, 3.0, size=int(leaves.sum())).astype(np.float32).astype(np.float64)
    fa.value = value
    thr = float(np.float32(0.4))
    det = make_det(kpl, A, B, r, rn, thr, fa)
    det.setInputCloud(xyz)
    det.setNormals(nrm)
    _, scores = det.compute()
    o_scores, o_kp = oracle.detect(xyz, nrm, A, B, r, rn, thr, cases.oracle_forest(fa))
    assert cases.same_bits(scores, o_scores)
    assert np.array_equal(det.getKeypointsIndices(), o_kp)
    assert len(np.unique(o_scores[np.isfinite(o_scores)])) > 100


@pytest.mark.parametrize("A,B,ntrees,nodes_per_tree,max_depth,walker", [
    (5, 6, 10, 5000, 25, "forest_sum_deep: top part in LDS, then whole blocks, trees in step"),
    (5, 6, 60, 1500, 20, "forest_sum_deep, more trees than ways"),
    (5, 6, 3, 40000, 28, "forest_sum_deep, chains of blocks"),
    (8, 10, 50, 100, 12, "forest_split_kernel, chained forest entirely in LDS"),
    (8, 10, 64, 1200, 22, "forest_split_kernel, chained forest beyond the LDS: deep nodes by exec-masked loads"),
    (8, 10, 41, 700, 20, "forest_split_kernel, chains of unequal length (41 trees over 16 chains)"),
    (8, 4, 47, 900, 24, "forest_split_kernel, F = 32 (the smallest chained forest), 47 trees"),
    (5, 6, 40, 150, 12, "forest_kernel, entirely in LDS, more trees than ways: out of step"),
])
def test_every_forest_walker(kpl, oracle, cases, A, B, ntrees, nodes_per_tree, max_depth, walker):
    """Class-label forests (integer leaf values, what the reference trains) of the shapes that take the
    different walks of kernels.hip: entirely in LDS, with 8-slot blocks below the top part or chained (forest.h),
    one lane or four lanes per point."""
    from tools import synth
    xyz, nrm = cases.cloud()
    mr = cases.resolution()
    r, rn = float(np.float32(6 * mr)), float(np.float32(4 * mr))
    feat = oracle.Grid(xyz, r).features(nrm, A, B, r, np.arange(0, len(xyz), 5, dtype=np.int32))
    fa = synth.random_forest(A * B, ntrees=ntrees, max_depth=max_depth, seed=17, target_nodes_per_tree=nodes_per_tree, feat=feat)
    thr = float(np.float32(0.4))
    det = make_det(kpl, A, B, r, rn, thr, fa)
    det.setInputCloud(xyz)
    det.setNormals(nrm)
    _, scores = det.compute()
    o_scores, o_kp = oracle.detect(xyz, nrm, A, B, r, rn, thr, cases.oracle_forest(fa))
    assert cases.same_bits(scores, o_scores), walker
    assert np.array_equal(det.getKeypointsIndices(), o_kp), walker
    assert len(np.unique(o_scores[np.isfinite(o_scores)])) >= 3       # the walks do reach different leaves


def test_pinned_host_staging_path(kpl, oracle, cases):
    """kpl_host_staging + kpl_detect_keypoints_staged: the view in pinned buffers of the handle (packed 12-byte
    records and PCL's 16 / 32-byte records), uploaded by DMA with the normals overlapped -- same keypoints and
    responses as the oracle; repeated calls, a larger and a smaller view on the same handle, the empty view."""
    A, B = 5, 6
    fa = cases.trained_forest(A, B)
    of = cases.oracle_forest(fa)
    det = None
    for (nx, ny, seed, xs, ns) in ((80, 60, 1, 12, 12), (120, 90, 3, 16, 32), (40, 30, 5, 12, 32), (80, 60, 1, 16, 12)):
        xyz, nrm = cases.cloud(nx, ny, seed=seed, nan_points=7, nan_normals=5)
        mr = oracle.cloud_resolution(xyz)
        r, rn, thr = float(np.float32(6 * mr)), float(np.float32(4 * mr)), float(np.float32(0.5))
        if det is None:
            det = make_det(kpl, A, B, r, rn, thr, fa)
        det.setRadiusSearch(r)
        det.setNonMaxRadius(rn)
        sx, sn = det.hostStaging(len(xyz), xs, ns)
        assert sx.shape == (len(xyz), xs // 4) and sn.shape == (len(xyz), ns // 4)
        sx[:] = 777.0
        sn[:] = -3.0                                     # padding floats must not matter
        sx[:, :3] = xyz
        sn[:, :3] = nrm
        o_sc, o_kp = oracle.detect(xyz, nrm, A, B, r, rn, thr, of)
        for rep in range(2):
            kp, kps = det.computeStaged()
            assert np.array_equal(kp, o_kp) and len(o_kp) > 0
            assert cases.same_bits(kps, o_sc[o_kp])
    det.hostStaging(0)
    kp, kps = det.computeStaged()
    assert len(kp) == 0


def test_chained_forest_in_a_batch_with_a_tree_order_forest(kpl, oracle, cases):
    """Two views in ONE batch: one with a chained forest (64 trees, F = 80: level-major throughout, leaf records
    linked -- forest.h), one whose forest has fractional leaf values (trees in order).  The batch cannot take the
    chained kernel, so the chained layout goes through forest_kernel's walk_deep (no blocks: every level node by node);
    alone, the chained view takes forest_split_kernel.  Same scores and keypoints either way."""
    import copy
    import torch
    from tools import synth
    A, B = 8, 10
    xyz, nrm = cases.cloud()
    mr = cases.resolution()
    r, rn, thr = float(np.float32(6 * mr)), float(np.float32(4 * mr)), float(np.float32(0.4))
    feat = oracle.Grid(xyz, r).features(nrm, A, B, r, np.arange(0, len(xyz), 5, dtype=np.int32))
    chained = synth.random_forest(A * B, ntrees=64, max_depth=22, seed=17, target_nodes_per_tree=1200, feat=feat)
    ordered = copy.deepcopy(synth.random_forest(A * B, ntrees=12, max_depth=16, seed=19, target_nodes_per_tree=1500, feat=feat))
    value = np.asarray(ordered.value, dtype=np.float64).copy()
    leaves = np.asarray(ordered.var) < 0
    value[leaves] = np.random.default_rng(7).uniform(-2.0, 2.0, size=int(leaves.sum())).astype(np.float32).astype(np.float64)
    ordered.value = value
    dev = torch.device("cuda", 0)
    dets, bufs, want = [], [], []
    for fa in (chained, ordered):
        det = make_det(kpl, A, B, r, rn, thr, fa)
        n = len(xyz)
        dx, dn = torch.from_numpy(xyz.copy()).to(dev), torch.from_numpy(nrm.copy()).to(dev)
        ds = torch.empty(n, dtype=torch.float32, device=dev)
        dk = torch.zeros(n + 1, dtype=torch.int32, device=dev)
        det.bindCloudDevice(dx.data_ptr(), 12, dn.data_ptr(), 12, n)
        dets.append(det)
        bufs.append((dx, dn, ds, dk))
        want.append(oracle.detect(xyz, nrm, A, B, r, rn, thr, cases.oracle_forest(fa)))
    st = torch.cuda.Stream()

    def check(which):
        st.synchronize()
        for k in which:
            ds, dk = bufs[k][2], bufs[k][3]
            assert dets[k].syncStatus(st.cuda_stream) == kpl.OK
            assert cases.same_bits(ds.cpu().numpy(), want[k][0])
            assert np.array_equal(dk[1:1 + int(dk[0].item())].cpu().numpy(), want[k][1])
            ds.fill_(-1.0)
            dk.zero_()

    kpl.compute_batch_device(dets, [b[2].data_ptr() for b in bufs], [b[3][1:].data_ptr() for b in bufs],
                             [len(b[2]) for b in bufs], [b[3][0:1].data_ptr() for b in bufs], st.cuda_stream)
    check((0, 1))
    kpl.compute_batch_device(dets[:1], [bufs[0][2].data_ptr()], [bufs[0][3][1:].data_ptr()], [len(bufs[0][2])],
                             [bufs[0][3][0:1].data_ptr()], st.cuda_stream)
    check((0,))
    assert len(np.unique(want[0][0][np.isfinite(want[0][0])])) > 10 and len(np.unique(want[1][0][np.isfinite(want[1][0])])) > 10
