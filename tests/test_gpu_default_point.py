"""The reference's OWN default operating point (round-3 verdict, missing item 2): TestDetector with no radius options =
ANNULI 5 x BINS 10, radiusFeatures 20, radiusNMS 4 in the cloud's units, threshold 0.85, draws_remove off, on cheff001
(/root/reference/src/main_test_detector.cpp:62-67, :105-106, :123-130): about 30 mesh resolutions, ~2 900 neighbors per
point -- 15 x the largest neighborhood any other test has.  Both neighbor orders, bit for bit against the committed oracle
outputs (tests/golden/cheff001.npz, tools/make_default_case.py), through the C-ABI and through the TestDetector binary."""
import json
import os
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "keypoint-learning_amd", "TestDetector")
FOREST = os.path.join(ROOT, "data", "forests", "cheff_a5b10_t10.yaml.gz")


@pytest.fixture(scope="module")
def gold():
    return np.load(os.path.join(ROOT, "tests", "golden", "cheff001.npz"))


@pytest.mark.parametrize("order", ["canonical", "sorted"])
def test_default_operating_point_matches_the_oracle(kpl, cases, gold, order):
    det = kpl.KeypointLearningDetector()
    assert det._p.n_annulus == 5 and det._p.n_bins == 10          # the class defaults ARE the main's ANNULI / BINS
    det.setNonMaxima(True)
    det.setNonMaxRadius(float(gold["r_nms"]))
    det.setNonMaximaDrawsRemove(False)
    det.setPredictionThreshold(float(gold["thr"]))
    det.setRadiusSearch(float(gold["r_feat"]))
    det.setSortedSearch(order == "sorted")
    assert det.loadForest(FOREST), det.lastError()
    det.setInputCloud(gold["xyz"])
    det.setNormals(gold["nrm"])
    _, scores = det.compute()
    assert cases.same_bits(scores, gold["scores_" + order])
    assert np.array_equal(det.getKeypointsIndices(), gold["kp_" + order])
    st = det.collectStats()
    kf = st["sum_kf"] / max(st["n_scored"], 1)
    assert 2000 < kf < 4000, kf                                    # the regime this test exists for
    # the normals TestDetector computes itself (k = 10) are the fixture's, bit for bit
    nk, _ = det.estimateNormals(gold["xyz"], k=10)
    assert cases.same_bits(nk, gold["nrm"])


def _write_ascii_pcd(path, xyz):
    with open(path, "w") as f:
        f.write("# .PCD v0.7 - Point Cloud Data file format\nVERSION 0.7\nFIELDS x y z\nSIZE 4 4 4\nTYPE F F F\nCOUNT 1 1 1\n"
                "WIDTH %d\nHEIGHT 1\nVIEWPOINT 0 0 0 1 0 0 0\nPOINTS %d\nDATA ascii\n" % (len(xyz), len(xyz)))
        for p in xyz:
            f.write("%.9g %.9g %.9g\n" % (p[0], p[1], p[2]))


def _read_keypoints(path):
    rows, data = [], False
    for line in open(path):
        if data:
            rows.append([float(v) for v in line.split()])
        elif line.startswith("DATA"):
            data = True
    return np.asarray(rows, dtype=np.float32).reshape(-1, 4)


@pytest.mark.parametrize("order", ["canonical", "sorted"])
def test_test_detector_with_no_radius_options_runs_the_reference_defaults(tmp_path, gold, order):
    """only --pathCloud / --pathRF / --pathKP are given (the reference's defaults for those are relative paths into its own
    checkout): every other value must be the reference main's default, and the keypoint file the oracle's."""
    cloud, kp_file = str(tmp_path / "cheff001.pcd"), str(tmp_path / "kp.pcd")
    _write_ascii_pcd(cloud, gold["xyz"])                           # like the reference's data file: ASCII, x y z only
    cmd = [EXE, "--pathCloud", cloud, "--pathRF", FOREST, "--pathKP", kp_file, "--json"] + (["--sortedSearch"] if order == "sorted" else [])
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    info = json.loads(out.stdout.strip().splitlines()[-1])
    assert info["radiusFeatures"] == 20.0 and info["radiusNMS"] == 4.0 and info["annuli"] == 5 and info["bins"] == 10
    assert np.float32(info["threshold"]) == np.float32(0.85)
    kp = gold["kp_" + order]
    got = _read_keypoints(kp_file)
    assert info["keypoints"] == len(kp) == len(got)
    assert np.array_equal(got[:, :3], gold["xyz"][kp])
    assert np.array_equal(got[:, 3], gold["scores_" + order][kp])


@pytest.mark.parametrize("walk", ["auto", "lanes2", "lanes4", "twopass2", "twopass4"])
def test_test_detector_walk_option_changes_no_output(tmp_path, gold, walk):
    """--walk (setFeatureWalk of the drop-in class): whatever walk is forced, the keypoint file is the oracle's; an unknown
    value is refused like any invalid argument"""
    cloud, kp_file = str(tmp_path / "cheff001.pcd"), str(tmp_path / "kp.pcd")
    _write_ascii_pcd(cloud, gold["xyz"])
    out = subprocess.run([EXE, "--pathCloud", cloud, "--pathRF", FOREST, "--pathKP", kp_file, "--json", "--walk", walk],
                         capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    info = json.loads(out.stdout.strip().splitlines()[-1])
    took = {"auto": ("two-pass", 4), "lanes2": ("lanes", 2), "lanes4": ("lanes", 4), "twopass2": ("two-pass", 2), "twopass4": ("two-pass", 4)}[walk]
    assert (info["walk"], info["lanes_per_point"]) == took         # auto: ~2 300 neighbors per point, known from the first call on
    kp = gold["kp_canonical"]
    got = _read_keypoints(kp_file)
    assert len(got) == len(kp)
    assert np.array_equal(got[:, :3], gold["xyz"][kp])
    assert np.array_equal(got[:, 3], gold["scores_canonical"][kp])
    if walk == "auto":
        bad = subprocess.run([EXE, "--pathCloud", cloud, "--pathRF", FOREST, "--walk", "sideways"], capture_output=True, text=True, timeout=900)
        assert bad.returncode != 0 and "--walk" in bad.stderr


def test_detect_views_at_the_default_operating_point_sorted(tmp_path, gold):
    """DetectViews (C++, batches + RCCL gather) with its defaults = the reference main's, sorted search: the first round needs
    KPL_ERR_RETRY for the key array (kpl_sync_status grows it), the keypoint files equal the oracle's sorted result"""
    exe = os.path.join(ROOT, "keypoint-learning_amd", "DetectViews")
    clouds = []
    for k in range(2):
        path = str(tmp_path / ("view%d.pcd" % k))
        _write_ascii_pcd(path, gold["xyz"])
        clouds.append(path)
    prefix = str(tmp_path / "kp")
    out = subprocess.run([exe, "--pathRF", FOREST, "--sortedSearch", "--devices", "all", "--pathKP", prefix] + clouds,
                         capture_output=True, text=True, timeout=900, env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0"))
    assert out.returncode == 0, out.stdout[-1000:] + out.stderr[-3000:]
    kp = gold["kp_sorted"]
    for k in range(2):
        got = _read_keypoints(prefix + "%d.pcd" % k)
        assert len(got) == len(kp)
        assert np.array_equal(got[:, :3], gold["xyz"][kp]) and np.array_equal(got[:, 3], gold["scores_sorted"][kp])
