"""Times compute() with non_maxima_draws_remove (the CLASS default, /root/reference/include/KeypointLearning.h:81) on a
200 k-point view whose candidates nearly all sit on plateaus (leaf values rounded to tenths: scores k / 10), with the points
in SCAN order (the order of a real range image: every maximum of a plateau waits for its left neighbor and for the row
above) and shuffled, for two draw thresholds; keypoints checked against the oracle once per row.
    python tools/bench_draws.py [--no-parity]
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

from oracle import kplo  # noqa: E402
from tests import helpers  # noqa: E402
from tools import forest_yaml, synth  # noqa: E402

kpl = importlib.import_module("keypoint-learning_amd")


def main():
    dev = torch.device("cuda", 0)
    fa = forest_yaml.load_forest(os.path.join(ROOT, "data", "forests", "synth200k_a5b6_t10.yaml.gz"))
    rows = []
    for order in ("scan", "shuffled"):
        xyz, nrm = synth.make_cloud(500, 400, seed=1)
        if order == "shuffled":
            xyz, nrm = synth.shuffle_cloud(xyz, nrm, 1001)
        n = len(xyz)
        det = kpl.KeypointLearningDetector()
        mr = det.cloudResolution(xyz)
        r, rn = float(np.float32(6 * mr)), float(np.float32(4 * mr))
        det.setNAnnulus(5); det.setNBins(6); det.setNonMaxima(True); det.setNonMaxRadius(rn)
        det.setPredictionThreshold(0.5); det.setRadiusSearch(r)
        helpers.load_arrays(det, fa)
        dx, dn = torch.from_numpy(np.array(xyz)).to(dev), torch.from_numpy(np.array(nrm)).to(dev)
        dk = torch.zeros(n + 1, dtype=torch.int32, device=dev)
        det.bindCloudDevice(dx.data_ptr(), 12, dn.data_ptr(), 12, n)
        for draws, mul in ((False, 0.0), (True, 2.0), (True, 4.0)):
            dthr = float(np.float32(mul * mr))
            det.setNonMaximaDrawsRemove(draws); det.setNonMaximaDrawsThreshold(dthr)

            def step():
                det.computeDevice(None, dk[1:].data_ptr(), n, dk[0:1].data_ptr())
            step()
            while det.syncStatus(None) == kpl.ERR_RETRY:
                step()
            times = []
            for _ in range(15):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(4):
                    step()
                torch.cuda.synchronize()
                times.append((time.perf_counter() - t0) / 4)
            cnt = int(dk[0].item())
            row = {"order": order, "draws_remove": draws, "draws_threshold_mr": mul, "compute_ms": round(float(np.median(times)) * 1e3, 4),
                   "keypoints": cnt}
            if "--no-parity" not in sys.argv:
                _, o_kp = kplo.detect(xyz, nrm, 5, 6, r, rn, 0.5, helpers.oracle_forest(fa), draws_remove=draws, draws_threshold=dthr,
                                      threads=helpers.usable_cores())
                row["parity"] = bool(np.array_equal(dk[1:1 + cnt].cpu().numpy(), o_kp))
                assert row["parity"], row
            rows.append(row)
            print(json.dumps(row), flush=True)


if __name__ == "__main__":
    main()
