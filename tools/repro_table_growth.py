import importlib, os, sys, time
import numpy as np
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT)
from tools import case_blob
from tests import helpers
kpl = importlib.import_module("keypoint-learning_amd")
c = case_blob.load_case(os.path.join(ROOT, "tests", "golden", "fuzz_r04_9004.npz"))
rA, rB = c["r"] * 1.137, c["r"] * 1.026
def oracle(r):
    from oracle import kplo
    forest = kplo.Forest(c["root"], c["var"], c["thrs"], c["left"], c["right"], c["value"], c["A"] * c["B"])
    return kplo.detect(c["xyz"], c["nrm"], c["A"], c["B"], r, c["rn"], c["thr"], forest, non_maxima=c["nms"], draws_remove=False)
t0 = time.time(); oB = oracle(rB); print("oracle B %.1f s, %d keypoints" % (time.time() - t0, len(oB[1])), flush=True)
bad = 0
N = int(sys.argv[1]) if len(sys.argv) > 1 else 20
for it in range(N):
    det = kpl.KeypointLearningDetector()
    det.setNAnnulus(c["A"]); det.setNBins(c["B"]); det.setNonMaxima(c["nms"]); det.setNonMaxRadius(c["rn"])
    det.setNonMaximaDrawsRemove(False); det.setPredictionThreshold(c["thr"])
    det.loadForestArrays(c["root"], c["var"], c["thrs"], c["left"], c["right"], c["value"], c["A"] * c["B"])
    det.setInputCloud(c["xyz"]); det.setNormals(c["nrm"])
    det.setRadiusSearch(rA); det.compute()          # grows the tables to ~2.2e8 cells
    det.setRadiusSearch(rB); _, sc = det.compute()  # needs ~2.4e8: the cell table grows again, nothing else does
    ok = helpers.same_bits(sc, oB[0]) and np.array_equal(det.getKeypointsIndices(), oB[1])
    if not ok:
        bad += 1
        print("iteration %d: MISMATCH (%d NaN scores of %d, %d keypoints)" % (it, int(np.isnan(sc).sum()), len(sc), len(det.getKeypointsIndices())), flush=True)
    det.close()
print("growth repro: %d iterations, %d mismatches" % (N, bad))
