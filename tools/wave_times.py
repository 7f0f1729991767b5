"""Diagnostic: distribution of per-wave durations of the score kernel (needs a KPL_ABLATE=16 build)."""
import importlib, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
kpl = importlib.import_module("keypoint-learning_amd")
from tools import synth
xyz, nrm = synth.make_cloud(500, 400, seed=1)
xyz, nrm = synth.shuffle_cloud(xyz, nrm, 1001)
mr = 0.8397691220715642
det = kpl.KeypointLearningDetector()
det.setNAnnulus(5); det.setNBins(6); det.setNonMaxima(True); det.setNonMaxRadius(4 * mr)
det.setNonMaximaDrawsRemove(False); det.setPredictionThreshold(0.85); det.setRadiusSearch(6 * mr)
det.loadForest("data/forests/synth200k_a5b6_t10.yaml.gz")
det.setInputCloud(xyz); det.setNormals(nrm)
for _ in range(3):
    _, cyc = det.compute()
c = np.sort(cyc)
print("per-point value: min %.0f p10 %.0f p50 %.0f p90 %.0f p99 %.0f max %.0f mean %.0f"
      % (c[0], c[len(c)//10], c[len(c)//2], c[9*len(c)//10], c[99*len(c)//100], c[-1], c.mean()))
