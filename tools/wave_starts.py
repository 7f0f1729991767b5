"""Diagnostic: spread of wave start times of the score kernel (needs a KPL_ABLATE=48 build)."""
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
    _, st = det.compute()
st = np.unique(st)                       # one value per wave (roughly)
st = np.sort((st - st.min()) % (1 << 24))
print("waves %d; start offsets (10 ns ticks): p10 %.0f p50 %.0f p90 %.0f p99 %.0f max %.0f"
      % (len(st), st[len(st)//10], st[len(st)//2], st[9*len(st)//10], st[99*len(st)//100], st[-1]))
