"""tools/train_detector.py end to end on a small synthetic dataset: features from the device kernel,
forest written in the OpenCV layout, and that forest driving the detector."""
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def write_pcd(path, arr, fields):
    arr = np.ascontiguousarray(arr, dtype=np.float32)
    k = arr.shape[1]
    os.makedirs(os.path.dirname(path), exist_ok=True)
    with open(path, "wb") as f:
        f.write(("# .PCD v0.7 - Point Cloud Data file format\nVERSION 0.7\nFIELDS %s\nSIZE %s\nTYPE %s\nCOUNT %s\nWIDTH %d\n"
                 "HEIGHT 1\nVIEWPOINT 0 0 0 1 0 0 0\nPOINTS %d\nDATA binary\n"
                 % (fields, " ".join(["4"] * k), " ".join(["F"] * k), " ".join(["1"] * k), len(arr), len(arr))).encode())
        f.write(arr.tobytes())


def test_train_then_detect(tmp_path, kpl, oracle, cases, capsys):
    from tools import synth, train_detector
    A, B = 5, 6
    data, ts, rf = tmp_path / "dataset", tmp_path / "ts", tmp_path / "rf"
    views = {}
    rng = np.random.default_rng(8)
    for m, model in enumerate(("modelA", "modelB")):
        for v in range(2):
            xyz, _ = synth.make_cloud(70, 60, seed=50 + 2 * m + v)
            xyz[7] = np.nan                                      # dropped like removeNaNFromPointCloud does
            name = "view%d" % v
            write_pcd(str(data / model / (name + ".pcd")), xyz, "x y z")
            good = xyz[np.isfinite(xyz).all(axis=1)]
            mr = oracle.cloud_resolution(good)
            r = 6.0
            nrm, _ = oracle.estimate_normals(good, k=10)
            g = oracle.Grid(good, r)
            cand = rng.choice(len(good), size=1200, replace=False).astype(np.int32)
            feat = g.features(nrm, A, B, r, cand)
            lab = synth.saliency_labels(feat, A, B, keep_fraction=0.3)
            jitter = rng.normal(0, 0.02, size=(len(cand), 3)).astype(np.float32)        # snapped back to the cloud
            pts = np.concatenate([good[cand] + jitter, np.zeros((len(cand), 1), np.float32)], axis=1)
            write_pcd(str(ts / model / "positives" / (name + ".pcd")), pts[lab == 0], "x y z intensity")
            write_pcd(str(ts / model / "negatives" / (name + ".pcd")), pts[lab == 1], "x y z intensity")
            views[(model, name)] = (good, nrm, cand, lab, mr)
    rc = train_detector.main(["--pathDataset", str(data), "--pathTrainingData", str(ts), "--pathRF", str(rf),
                              "--nameRF", "t.yaml.gz", "--ntrees", "12", "--depth", "9", "--annuli", str(A), "--bins", str(B),
                              "--radiusFeatures", "6.0", "--nnNormals", "10"])
    assert rc == 0
    info = json.loads(capsys.readouterr().out.strip().splitlines()[-1])
    assert info["views"] == 4 and info["rows"] == 4 * 1200 and info["ntrees"] == 12
    assert info["train_error"] < 0.1 and info["test_error"] < 0.3
    assert os.path.exists(rf / "training_parameters.log")
    # the forest drives the detector: keypoints come out where the training set had its positives
    good, nrm, cand, lab, mr = views[("modelA", "view0")]
    det = kpl.KeypointLearningDetector()
    det.setNAnnulus(A); det.setNBins(B); det.setNonMaxima(False); det.setRadiusSearch(6.0); det.setNonMaxRadius(4.0)
    det.setPredictionThreshold(0.0)
    assert det.loadForest(str(rf / "t.yaml.gz")), det.lastError()
    det.setInputCloud(good); det.setNormals(nrm)
    _, scores = det.compute()
    assert scores[cand[lab == 0]].mean() > scores[cand[lab == 1]].mean() + 0.5
    # and the same file through the oracle gives the same scores
    from tools import forest_yaml
    fa = forest_yaml.load_forest(str(rf / "t.yaml.gz"))
    o_scores, _ = oracle.detect(good, nrm, A, B, 6.0, 4.0, 0.0, cases.oracle_forest(fa), non_maxima=False)
    assert cases.same_bits(scores, o_scores)
