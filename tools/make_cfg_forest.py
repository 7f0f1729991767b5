"""Regenerates data/forests/synth200k_a5b6_t10.yaml.gz, the stand-in for the reference's missing SHOT forest
(data/forest/SHOT-LaserScanner.yaml.gz is listed in the reference's README but absent from the checkout):
10 extremely-randomised trees trained on the oracle's 5 x 6 features of the BASELINE.json configs[1] view
(every second point), label 0 (= keypoint) for the 12 % most non-flat feature rows.  Deterministic: the
output is byte-identical to the committed file.
    python tools/make_cfg_forest.py [out.yaml.gz]
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import kplo  # noqa: E402
from tools import forest_yaml, synth  # noqa: E402


def main():
    out = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "data", "forests", "synth200k_a5b6_t10.yaml.gz")
    xyz, nrm = synth.make_cloud(500, 400, seed=1)
    xyz, nrm = synth.shuffle_cloud(xyz, nrm, 1001)
    mr = kplo.cloud_resolution(xyz)
    A, B = 5, 6
    r = float(np.float32(6 * mr))
    g = kplo.Grid(xyz, r)
    feat = g.features(nrm, A, B, r, np.arange(len(xyz)))
    lab = synth.saliency_labels(feat, A, B)
    sub = np.arange(0, len(xyz), 2)
    fa = synth.train_extra_trees(feat[sub], lab[sub], ntrees=10, max_depth=25, min_samples=6, seed=2, candidates=16)
    forest_yaml.save_forest(fa, out)
    print(out, fa.ntrees, "trees", fa.nnodes, "nodes", os.path.getsize(out), "bytes")


if __name__ == "__main__":
    main()
