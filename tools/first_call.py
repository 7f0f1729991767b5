"""What ONE TestDetector run of the reference's default operating point costs on a fresh process (the drop-in use: one
compute() per handle): writes cheff001 (tests/golden/cheff001.npz) as the ASCII PCD the reference ships, runs the TestDetector
binary with no radius options, once per --walk value, and prints its JSON lines (compute_first_s = the first call: scratch
allocation, the walk chosen from the bounding-box estimate; compute_s = the best of the later ones)."""
import json
import os
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "keypoint-learning_amd", "TestDetector")
FOREST = os.path.join(ROOT, "data", "forests", "cheff_a5b10_t10.yaml.gz")


def main():
    z = np.load(os.path.join(ROOT, "tests", "golden", "cheff001.npz"))
    with tempfile.TemporaryDirectory() as tmp:
        cloud = os.path.join(tmp, "cheff001.pcd")
        xyz = z["xyz"]
        with open(cloud, "w") as f:
            f.write("# .PCD v0.7 - Point Cloud Data file format\nVERSION 0.7\nFIELDS x y z\nSIZE 4 4 4\nTYPE F F F\nCOUNT 1 1 1\n"
                    "WIDTH %d\nHEIGHT 1\nVIEWPOINT 0 0 0 1 0 0 0\nPOINTS %d\nDATA ascii\n" % (len(xyz), len(xyz)))
            for p in xyz:
                f.write("%.9g %.9g %.9g\n" % (p[0], p[1], p[2]))
        for walk in ["auto", "lanes2", "twopass4"]:
            for rep in range(2):
                extra = [a for a in sys.argv[1:] if a != "--trace"]
                out = subprocess.run([EXE, "--pathCloud", cloud, "--pathRF", FOREST, "--json", "--walk", walk] + extra,
                                     capture_output=True, text=True, timeout=600)
                if "--trace" in sys.argv[1:]:      # a scratch libkpl built with -DKPL_TRACE_HOST (LD_PRELOAD) stamps its steps on stderr
                    print(out.stderr, flush=True)
                if out.returncode != 0:
                    print(out.stderr[-2000:])
                    return 1
                row = json.loads(out.stdout.strip().splitlines()[-1])
                row["walk_option"] = walk
                print(json.dumps(row), flush=True)
    return 0


if __name__ == "__main__":
    sys.exit(main())
