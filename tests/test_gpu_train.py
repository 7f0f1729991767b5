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


def test_batched_training_features_equal_the_single_view_entry_point(kpl, oracle, cases):
    """kpl_compute_features_batch_device over 5 views of different sizes, radii and neighbor orders in one launch = what
    kpl_compute_features gives view by view = the oracle's rows; indices out of range and non-finite points give NaN rows."""
    import torch
    from tools import synth
    dev = torch.device("cuda", 0)
    A, B = 5, 6
    dets, keep, idx_p, out_p, ms, outs, want = [], [], [], [], [], [], []
    for k, (nx, ny, rmul, srt) in enumerate([(80, 60, 6.0, False), (50, 40, 8.0, True), (90, 70, 5.0, False), (30, 30, 6.0, False), (64, 48, 7.0, True)]):
        xyz, nrm = synth.make_cloud(nx, ny, seed=20 + k, nan_points=3 if k == 2 else 0)
        xyz, nrm = synth.shuffle_cloud(xyz, nrm, 2000 + k)
        mr = oracle.cloud_resolution(xyz[np.isfinite(xyz).all(axis=1)])
        r = float(np.float32(rmul * mr))
        rng = np.random.default_rng(k)
        idx = rng.choice(len(xyz), size=200 + 37 * k, replace=False).astype(np.int32)
        if k == 0:
            idx[5], idx[9] = -1, len(xyz) + 3                          # out of range: NaN rows
        det = kpl.KeypointLearningDetector()
        det.setNAnnulus(A); det.setNBins(B); det.setRadiusSearch(r); det.setSortedSearch(srt)
        dx, dn, di = torch.from_numpy(xyz).to(dev), torch.from_numpy(nrm).to(dev), torch.from_numpy(idx).to(dev)
        do = torch.full((len(idx), A * B), -7.0, dtype=torch.float32, device=dev)
        det.bindCloudDevice(dx.data_ptr(), 12, dn.data_ptr(), 12, len(xyz))
        dets.append(det); keep.append((dx, dn, di)); idx_p.append(di.data_ptr()); out_p.append(do.data_ptr()); ms.append(len(idx)); outs.append(do)
        single = kpl.KeypointLearningDetector()
        single.setNAnnulus(A); single.setNBins(B); single.setRadiusSearch(r); single.setSortedSearch(srt)
        single.setInputCloud(xyz); single.setNormals(nrm)
        want.append(single.computePointsForTrainingFeatures(idx))
        ok = (idx >= 0) & (idx < len(xyz))
        ok[ok] &= np.isfinite(xyz[idx[ok]]).all(axis=1)
        g = oracle.Grid(xyz, r)
        rows = g.features(nrm, A, B, r, idx[ok], order=oracle.ORDER_SORTED if srt else oracle.ORDER_CANONICAL)
        assert cases.same_bits(want[-1][ok], rows) and np.isnan(want[-1][~ok]).all()
    torch.cuda.synchronize()
    for attempt in range(4):
        kpl.compute_features_batch_device(dets, idx_p, ms, out_p, None)
        if kpl.ERR_RETRY not in [d.syncStatus(None) for d in dets]:
            break
    for o, w in zip(outs, want):
        assert cases.same_bits(o.cpu().numpy(), w)
