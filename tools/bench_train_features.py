"""Feature extraction of the TrainDetector counterpart (tools/train_detector.py; /root/reference/src/main_train_detector.cpp:413-446):
views per second, one view per call (kpl_compute_features, the round-4 path) against 8 views per launch
(kpl_compute_features_batch_device).  Views = the three cheff clouds of tests/golden (data/point_cloud_test), 24 in all, 600
training points each, r = 6 mesh resolutions, 5 x 6; rows compared bit for bit between the two paths.

    python tools/bench_train_features.py
"""
import importlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from tools import train_detector  # noqa: E402

kpl = importlib.import_module("keypoint-learning_amd")


def main():
    clouds = [np.load(os.path.join(ROOT, "tests", "golden", "cheff00%d.npz" % k)) for k in (0, 1, 2)]
    A, B = 5, 6
    rng = np.random.default_rng(1)
    views = []
    for k in range(24):
        c = clouds[k % 3]
        xyz, nrm = np.ascontiguousarray(c["xyz"], dtype=np.float32), np.ascontiguousarray(c["nrm"], dtype=np.float32)
        idx = rng.choice(len(xyz), size=600, replace=False).astype(np.int32)
        views.append((xyz, nrm, idx))
    mr = float(clouds[0]["mr"])
    r = float(np.float32(6.0 * mr))
    dets = []
    for _ in range(train_detector.VIEWS_PER_BATCH):
        det = kpl.KeypointLearningDetector()
        det.setNAnnulus(A); det.setNBins(B); det.setRadiusSearch(r)
        dets.append(det)
    rows = {}
    for name in ("one view per call", "8 views per launch"):
        for rep in range(3):                      # (the first repetition grows the tables)
            t0 = time.perf_counter()
            out = []
            if name.startswith("one"):
                for xyz, nrm, idx in views:
                    dets[0].setInputCloud(xyz)
                    dets[0].setNormals(nrm)
                    out.append(dets[0].computePointsForTrainingFeatures(idx))
            else:
                for b0 in range(0, len(views), train_detector.VIEWS_PER_BATCH):
                    got, _ = train_detector.batch_features(kpl, dets, views[b0:b0 + train_detector.VIEWS_PER_BATCH], 0)
                    out.extend(got)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
        rows[name] = (dt, out)
    same = all(np.array_equal(a.view(np.uint32), b.view(np.uint32)) for a, b in zip(rows["one view per call"][1], rows["8 views per launch"][1]))
    print(json.dumps({"what": "training-feature extraction, 24 cheff views x 600 points, upload + index + features + rows back",
                      "one_view_per_call": {"seconds": round(rows["one view per call"][0], 4), "views_per_s": round(24 / rows["one view per call"][0], 1)},
                      "8_views_per_launch": {"seconds": round(rows["8 views per launch"][0], 4), "views_per_s": round(24 / rows["8 views per launch"][0], 1)},
                      "rows_bit_identical": bool(same)}))
    assert same


if __name__ == "__main__":
    main()
