"""tools/graph_latency.py: one view alone on the GPU -- compute() as 11 launches against the same call replayed from a captured
hipGraph (kpl_compute_device only enqueues, so a caller may capture it: tests/test_gpu_golden.py).  cfg1 (cheff000, 62 k points)
and cfg2 (200 k synthetic points), wall time per call incl. the wait for the result count; keypoint lists compared."""
import importlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

kpl = importlib.import_module("keypoint-learning_amd")
from tools import synth  # noqa: E402

FOREST = os.path.join(ROOT, "data", "forests", "synth200k_a5b6_t10.yaml.gz")


def view(name):
    if name == "cfg1":
        z = np.load(os.path.join(ROOT, "tests", "golden", "cheff000.npz"))
        return z["xyz"], z["nrm"], float(z["r_feat"]), float(z["r_nms"])
    xyz, nrm = synth.make_cloud(500, 400, seed=1)
    xyz, nrm = synth.shuffle_cloud(xyz, nrm, 1001)
    det = kpl.KeypointLearningDetector(device=0)
    mr = det.cloudResolution(xyz)
    return xyz, nrm, float(np.float32(6 * mr)), float(np.float32(4 * mr))


def main():
    dev = torch.device("cuda", 0)
    for name in ("cfg1", "cfg2"):
        xyz, nrm, r, rn = view(name)
        n = len(xyz)
        det = kpl.KeypointLearningDetector(device=0)
        det.setNAnnulus(5); det.setNBins(6); det.setNonMaxima(True); det.setNonMaxRadius(rn); det.setNonMaximaDrawsRemove(False)
        det.setPredictionThreshold(float(np.float32(0.85))); det.setRadiusSearch(r)
        assert det.loadForest(FOREST)
        dx, dn = torch.from_numpy(np.ascontiguousarray(xyz)).to(dev), torch.from_numpy(np.ascontiguousarray(nrm)).to(dev)
        dk = torch.zeros(n + 1, dtype=torch.int32, device=dev)
        det.bindCloudDevice(dx.data_ptr(), 12, dn.data_ptr(), 12, n)
        st = torch.cuda.Stream()
        call = lambda s: det.computeDevice(None, dk[1:].data_ptr(), n, dk[0:1].data_ptr(), s)
        for _ in range(4):
            call(st.cuda_stream)
            st.synchronize()
            det.syncStatus(st.cuda_stream)
        ref = dk[:1 + int(dk[0].item())].cpu().numpy().copy()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=st):
            call(torch.cuda.current_stream().cuda_stream)
        rows = {}
        for mode in ("launches", "graph"):
            best = []
            for rep in range(5):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(50):
                    if mode == "graph":
                        graph.replay()
                        torch.cuda.current_stream().synchronize()
                    else:
                        call(st.cuda_stream)
                        st.synchronize()
                best.append((time.perf_counter() - t0) / 50 * 1e3)
            rows[mode + "_ms"] = round(min(best), 4)
            torch.cuda.synchronize()
            assert np.array_equal(dk[:1 + int(dk[0].item())].cpu().numpy(), ref), mode
        print(json.dumps({"view": name, "points": n, **rows, "keypoints": int(ref[0])}), flush=True)


if __name__ == "__main__":
    main()
