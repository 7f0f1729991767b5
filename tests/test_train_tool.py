"""Host-side pieces of tools/train_detector.py (the TrainDetector counterpart,
/root/reference/src/main_train_detector.cpp:153-519) that need no GPU."""
import numpy as np
import pytest

from tests import helpers
from tools import forest_yaml, train_detector


def test_sklearn_forest_flattens_to_the_same_decisions(tmp_path):
    from sklearn.ensemble import RandomForestClassifier
    rng = np.random.default_rng(4)
    x = rng.uniform(0, 1, size=(3000, 12)).astype(np.float32)
    x[:, 3] = np.round(x[:, 3] * 8) / 8                      # repeated values: thresholds fall between float32 neighbors
    y = ((x[:, 0] + x[:, 3] > 1.0) ^ (x[:, 5] > 0.7)).astype(np.int32)
    clf = RandomForestClassifier(n_estimators=9, max_depth=7, random_state=1).fit(x, y)
    fa = train_detector.sklearn_to_arrays(clf, 12)
    votes = train_detector.forest_votes(fa, x)
    per_tree = np.stack([est.predict(x) for est in clf.estimators_]).mean(axis=0)
    assert np.allclose(votes, per_tree)                      # every tree takes the same branch on every row
    # YAML round trip through the writer and the independent reader, then the oracle's predict
    path = tmp_path / "f.yaml.gz"
    forest_yaml.save_forest(fa, str(path))
    fb = forest_yaml.load_forest(str(path))
    of = helpers.oracle_forest(fb)
    for i in range(0, 3000, 97):
        assert of.predict_sum(x[i])[0] / fb.ntrees == pytest.approx(votes[i], abs=1e-6)


def test_uniform_sampling_keeps_the_point_closest_to_each_voxel_centre():
    rng = np.random.default_rng(5)
    xyz = rng.uniform(-3, 3, size=(2000, 3)).astype(np.float32)
    keep = train_detector.uniform_sampling(xyz, 1.0)
    assert np.all(np.diff(keep) > 0)
    vox = np.floor(xyz.astype(np.float64)).astype(np.int64)
    assert len(keep) == len(np.unique(vox, axis=0))
    for i in keep[::17]:
        same = np.all(vox == vox[i], axis=1)
        d = ((xyz[same] - (vox[i] + 0.5)) ** 2).sum(axis=1)
        assert ((xyz[i] - (vox[i] + 0.5)) ** 2).sum() == d.min()


def test_snap_to_cloud():
    xyz = np.float32([[0, 0, 0], [1, 0, 0], [0, 2, 0]])
    q = np.float32([[0.9, 0.1, 0], [0, 1.6, 0.1], [0.1, 0, 0]])
    assert train_detector.snap_to_cloud(xyz, q).tolist() == [1, 2, 0]
    assert len(train_detector.snap_to_cloud(xyz, q[:0])) == 0


def test_ply_reader(tmp_path):
    from tools import cloud_io
    rng = np.random.default_rng(1)
    xyz = rng.normal(size=(50, 3)).astype(np.float32)
    a = tmp_path / "a.ply"
    with open(a, "w") as f:
        f.write("ply\nformat ascii 1.0\nelement vertex 50\nproperty float x\nproperty float y\nproperty float z\n"
                "element face 0\nproperty list uchar int vertex_indices\nend_header\n")
        for p in xyz:
            f.write("%.9g %.9g %.9g\n" % tuple(p))
    assert np.array_equal(cloud_io.read_cloud_xyz(str(a)), xyz)
    b = tmp_path / "b.ply"
    rec = np.zeros(50, dtype=[("x", "<f4"), ("y", "<f4"), ("z", "<f4"), ("red", "u1"), ("nx", "<f8")])
    rec["x"], rec["y"], rec["z"] = xyz[:, 0], xyz[:, 1], xyz[:, 2]
    with open(b, "wb") as f:
        f.write(b"ply\nformat binary_little_endian 1.0\nelement vertex 50\nproperty float x\nproperty float y\n"
                b"property float z\nproperty uchar red\nproperty double nx\nend_header\n")
        f.write(rec.tobytes())
    assert np.array_equal(cloud_io.read_cloud_xyz(str(b)), xyz)
