"""tools/time_sorted.py [canonical] [noshuffle] [rmul=<radius in mesh resolutions, default 6>]: the feature stage of one batch of 8 synthetic 200 k-point views (the bench workload) in
sorted-search mode, timed with the handle's events.  Timing only -- used with scratch builds of ablated kernels
(KPL_LIB_PATH=build/variants/libkpl_<name>.so, tools/build_variant.sh) to see where the time of the stage goes."""
import importlib
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

kpl = importlib.import_module("keypoint-learning_amd")
from tools import synth  # noqa: E402

FOREST = os.path.join(ROOT, "data", "forests", "synth200k_a5b6_t10.yaml.gz")


def main():
    srt = "canonical" not in sys.argv[1:]
    rmul = float(([a[5:] for a in sys.argv[1:] if a.startswith("rmul=")] or ["6"])[0])
    dev = torch.device("cuda", 0)
    nb, nx, ny = 8, 500, 400
    dets, keep, ps, pk, pc = [], [], [], [], []
    for k in range(nb):
        xyz, nrm = synth.make_cloud(nx, ny, seed=1 + k)
        if "noshuffle" not in sys.argv[1:]:           # (scan order instead of a random one: the caller's arrays, which the sorted mode
            xyz, nrm = synth.shuffle_cloud(xyz, nrm, 1001 + k)      # reads the normals of the neighbors from by ORIGINAL index, become local)
        n = len(xyz)
        det = kpl.KeypointLearningDetector(device=0)
        mr = det.cloudResolution(xyz)
        det.setNAnnulus(5); det.setNBins(6); det.setNonMaxima(True); det.setNonMaxRadius(float(np.float32(4.0 * mr)))
        det.setNonMaximaDrawsRemove(False); det.setPredictionThreshold(float(np.float32(0.85)))
        det.setRadiusSearch(float(np.float32(rmul * mr))); det.setSortedSearch(srt)
        assert det.loadForest(FOREST), det.lastError()
        dx, dn = torch.from_numpy(xyz).to(dev), torch.from_numpy(nrm).to(dev)
        sc = torch.zeros(n, dtype=torch.float32, device=dev)
        kp = torch.zeros(n, dtype=torch.int32, device=dev)
        cn = torch.zeros(1, dtype=torch.int32, device=dev)
        det.bindCloudDevice(dx.data_ptr(), 12, dn.data_ptr(), 12, n)
        dets.append(det); keep.append((dx, dn, sc, kp, cn)); ps.append(sc.data_ptr()); pk.append(kp.data_ptr()); pc.append(cn.data_ptr())
    torch.cuda.synchronize()
    st = torch.cuda.Stream()
    run = lambda: kpl.compute_batch_device(dets, ps, pk, [n] * nb, pc, st.cuda_stream)
    # the cell tables and the key array of the large path grow on the first calls (KPL_ERR_RETRY): EVERY detector is synced
    # after every run (no short circuit) until all of them report OK -- a timed run of a handle that is still growing would
    # measure failing calls
    for attempt in range(8):
        run()
        torch.cuda.synchronize()
        rcs = [d.syncStatus(None) for d in dets]      # (raises on anything but OK / RETRY)
        if kpl.ERR_RETRY not in rcs:
            break
    else:
        raise SystemExit("still growing after 8 attempts: %s" % rcs)
    for _ in range(4):                     # (the handles' hints settle over a few synced calls: list capacity, all-large, word lists)
        run()
        torch.cuda.synchronize()
        rcs = [d.syncStatus(None) for d in dets]
    run()
    torch.cuda.synchronize()
    assert all(d.syncStatus(None) == kpl.OK for d in dets)
    dets[0].enableTiming(True)
    for _ in range(20):
        run()
    torch.cuda.synchronize()
    t = dets[0].getTiming()
    assert all(d.syncStatus(None) == kpl.OK for d in dets), "a timed run failed on the device"
    assert all(int(k[4].item()) >= 0 for k in keep)
    calls = max(t["calls"], 1)
    print(json.dumps({"lib": os.environ.get("KPL_LIB_PATH", "libkpl.so"), "sorted": srt, "rmul": rmul, "launch": dets[0].getLastLaunch(), "feature_ms": round(t["feature_ms"] / calls, 4),
                      "forest_ms": round(t["forest_ms"] / calls, 4), "keypoints": int(sum(int(k[4].item()) for k in keep))}))


if __name__ == "__main__":
    main()
