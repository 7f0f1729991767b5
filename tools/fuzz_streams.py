"""tools/fuzz_streams.py [seconds] [seed]: the STATEFUL counterpart of tools/fuzz_parity.py.  A libkpl handle carries hints from
call to call -- which walk the canonical order takes, how many accept words a point collects, in sorted order the list capacity,
"every point is listed" and the word-list mode, the estimate of a first host call -- and every one of them is a choice of speed
that must never change a bit.  Here ONE handle lives through a random stream of views: the same surface at changing density
(neighbors per point from a handful to several hundred), changing size, radius and neighbor order, through the host entry point
or the device entry point with status reads, one to five calls per view; every call's scores and keypoint list are compared with
the oracle's.  Prints one JSON line; exit code 1 on a mismatch (the offending step is saved under gpurun_out/)."""
import importlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import kplo  # noqa: E402
from tests import helpers  # noqa: E402
from tools import synth  # noqa: E402


def run(budget, seed, verbose=False):
    import torch
    kpl = importlib.import_module("keypoint-learning_amd")
    rng = np.random.default_rng(seed)
    dev = torch.device("cuda", 0)
    A, B = 5, 6
    fa = helpers.trained_forest(A, B)
    of = helpers.oracle_forest(fa)
    cores = helpers.usable_cores()
    t_end = time.time() + budget
    streams = calls = 0
    modes = {}
    while time.time() < t_end:
        streams += 1
        det = kpl.KeypointLearningDetector()
        det.setNAnnulus(A); det.setNBins(B); det.setNonMaxima(True); det.setNonMaximaDrawsRemove(False)
        helpers.load_arrays(det, fa)
        base_n = int(rng.choice([3000, 8000, 20000]))
        nx = int(np.sqrt(base_n * rng.uniform(0.7, 1.4)))
        base_xyz, base_nrm = synth.make_cloud(nx, max(8, base_n // nx), seed=int(rng.integers(1, 1 << 30)),
                                              nan_points=int(rng.integers(0, 4)), nan_normals=int(rng.integers(0, 4)))
        mr = kplo.cloud_resolution(base_xyz)
        radius = float(np.float32(mr * rng.uniform(5, 12)))
        srt = bool(rng.random() < 0.6)
        for step in range(int(rng.integers(4, 10))):
            if time.time() > t_end:
                break
            # what changes from view to view of the stream
            u = rng.random()
            if u < 0.25:
                radius = float(np.float32(mr * rng.uniform(4, 14)))
            elif u < 0.35:
                srt = not srt
            scale = float(rng.choice([1.0, 1.0, 1.0, 2.0, 3.0, 0.7]))             # the same surface, sparser or denser
            keep = slice(0, len(base_xyz) if rng.random() < 0.8 else int(len(base_xyz) * rng.uniform(0.5, 0.9)))
            xyz = np.ascontiguousarray(base_xyz[keep] * np.float32([scale, scale, 1.0]), np.float32)
            nrm = np.ascontiguousarray(base_nrm[keep], np.float32)
            rn, thr = float(np.float32(radius * rng.uniform(0.3, 0.8))), float(np.float32(rng.choice([0.0, 0.5, 0.85])))
            det.setRadiusSearch(radius); det.setNonMaxRadius(rn); det.setPredictionThreshold(thr); det.setSortedSearch(srt)
            want = kplo.detect(xyz, nrm, A, B, radius, rn, thr, of, order=kplo.ORDER_SORTED if srt else kplo.ORDER_CANONICAL, threads=cores)
            device_entry = bool(rng.random() < 0.4)
            n = len(xyz)
            if device_entry:
                dx, dn = torch.from_numpy(xyz).to(dev), torch.from_numpy(nrm).to(dev)
                ds = torch.empty(n, dtype=torch.float32, device=dev)
                dk = torch.zeros(n + 1, dtype=torch.int32, device=dev)
                torch.cuda.synchronize()
                det.bindCloudDevice(dx.data_ptr(), 12, dn.data_ptr(), 12, n)
            else:
                det.setInputCloud(xyz)
                det.setNormals(nrm)
            for rep in range(int(rng.integers(1, 6))):
                calls += 1
                if device_entry:
                    for attempt in range(8):
                        det.computeDevice(ds.data_ptr(), dk[1:].data_ptr(), n, dk[0:1].data_ptr(), None)
                        torch.cuda.synchronize()
                        if det.syncStatus(None) == kpl.OK:
                            break
                    else:
                        raise SystemExit("still RETRY after 8 attempts")
                    scores = ds.cpu().numpy()
                    kp = dk[1:1 + int(dk[0].item())].cpu().numpy()
                else:
                    _, scores = det.compute()
                    kp = det.getKeypointsIndices()
                li = det.getLastLaunch()
                key = "sorted:%d/%d/%d" % (li["walk"], li["sorted_list_keys"], li["sorted_all_large"]) if srt else \
                    "canonical:%d/%d/%d" % (li["walk"], li["lanes_per_point"], li["accept_words"])
                modes[key] = modes.get(key, 0) + 1
                if not (helpers.same_bits(scores, want[0]) and np.array_equal(kp, want[1])):
                    d = os.path.join(ROOT, "gpurun_out")
                    path = os.path.join(d if os.path.isdir(d) else ".", "fuzz_stream_failure.npz")
                    np.savez(path, xyz=xyz, nrm=nrm, radius=radius, rn=rn, thr=thr, srt=srt, scores=scores, kp=kp, o_scores=want[0], o_kp=want[1])
                    print(json.dumps({"MISMATCH": True, "seed": seed, "stream": streams, "step": step, "rep": rep, "launch": li,
                                      "device_entry": device_entry, "n": n, "saved": path}))
                    return 1
        del det
    print(json.dumps({"streams": streams, "calls": calls, "seed": seed, "seconds": budget, "all_bit_exact": True,
                      "launch_modes_seen": dict(sorted(modes.items()))}))
    return 0


if __name__ == "__main__":
    sys.exit(run(float(sys.argv[1]) if len(sys.argv) > 1 else 60.0, int(sys.argv[2]) if len(sys.argv) > 2 else 1))
