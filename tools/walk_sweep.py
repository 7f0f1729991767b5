"""Which walk of the canonical order is fastest at which neighborhood size (kpl_set_feature_walk; every walk gives the same
bits): feature-stage time of one view per walk and lanes-per-point, radius by radius.  The numbers behind choose_walk
(csrc/api.cpp).

    python tools/walk_sweep.py [nx ny] [r/mr ...]          (default: 250 x 252 and 707 x 707 points, 6 .. 30 mr)
"""
import importlib
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from tools import synth  # noqa: E402

kpl = importlib.import_module("keypoint-learning_amd")
FOREST = os.path.join(ROOT, "data", "forests", "synth200k_a5b6_t10.yaml.gz")
DEV = torch.device("cuda", 0)


def feature_ms(det, n, bufs, reps=12):
    dx, dn, ds, dk = bufs
    st = torch.cuda.current_stream().cuda_stream

    def step():
        det.computeDevice(ds.data_ptr(), dk[1:].data_ptr(), n, dk[0:1].data_ptr(), st)
    step()
    while det.syncStatus(st) == kpl.ERR_RETRY:
        step()
    for _ in range(2):
        step()
    torch.cuda.synchronize()
    det.enableTiming(True)
    for _ in range(reps):
        step()
    torch.cuda.synchronize()
    tm = det.getTiming()
    det.enableTiming(False)
    return tm["feature_ms"] / max(tm["calls"], 1), ds.cpu().numpy()


def main():
    args = [a for a in sys.argv[1:]]
    sizes = [(250, 252), (707, 707)]
    radii = [6.0, 10.0, 14.0, 18.0, 24.0, 30.0]
    if len(args) >= 2:
        sizes = [(int(args[0]), int(args[1]))]
        if len(args) > 2:
            radii = [float(a) for a in args[2:]]
    for nx, ny in sizes:
        xyz, nrm = synth.make_cloud(nx, ny, seed=4)
        xyz, nrm = synth.shuffle_cloud(xyz, nrm, 1004)
        n = len(xyz)
        dx, dn = torch.from_numpy(np.array(xyz)).to(DEV), torch.from_numpy(np.array(nrm)).to(DEV)
        bufs = (dx, dn, torch.empty(n, dtype=torch.float32, device=DEV), torch.zeros(n + 1, dtype=torch.int32, device=DEV))
        for rmul in radii:
            row, ref = {"points": n, "r_over_mr": rmul}, None
            for walk, lanes in ((kpl.WALK_LANES, 2), (kpl.WALK_LANES, 4), (kpl.WALK_TWO_PASS, 2), (kpl.WALK_TWO_PASS, 4)):
                det = kpl.KeypointLearningDetector()
                mr = det.cloudResolution(xyz)
                det.setNAnnulus(5); det.setNBins(6); det.setNonMaxima(True); det.setNonMaximaDrawsRemove(False)
                det.setNonMaxRadius(float(np.float32(4 * mr))); det.setPredictionThreshold(float(np.float32(0.85)))
                det.setRadiusSearch(float(np.float32(rmul * mr)))
                assert det.loadForest(FOREST)
                det.setFeatureWalk(walk, lanes)
                det.bindCloudDevice(dx.data_ptr(), 12, dn.data_ptr(), 12, n)
                ms, sc = feature_ms(det, n, bufs)
                if ref is None:
                    ref = sc
                    st = det.collectStats(torch.cuda.current_stream().cuda_stream)
                    row["K_f"] = round(st["sum_kf"] / max(st["n_scored"], 1), 1)
                assert np.array_equal(np.asarray(sc).view(np.uint32), np.asarray(ref).view(np.uint32)), "walks disagree"
                row[("two_pass" if walk == kpl.WALK_TWO_PASS else "lanes") + str(lanes) + "_ms"] = round(ms, 4)
            print(json.dumps(row), flush=True)


if __name__ == "__main__":
    main()
