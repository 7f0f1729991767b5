"""A saved fuzz case (tools/fuzz_parity.py's fuzz_failure.npz layout, e.g. tests/golden/fuzz_31337.npz) as the flat
binary blob tests/csrc/hammer_case.cpp reads, with the ORACLE's scores and keypoints as the expectation.
    python tools/case_blob.py tests/golden/fuzz_31337.npz gpurun_out/fuzz_31337.blob
"""
import os
import struct
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import kplo  # noqa: E402


def load_case(path):
    d = np.load(path)
    c = {k: d[k] for k in d.files}
    for k in ("A", "B"):
        c[k] = int(c[k])
    for k in ("r", "rn", "thr", "dthr"):
        c[k] = float(c[k])
    for k in ("nms", "draws", "srt"):
        c[k] = bool(c[k])
    return c


def oracle_result(c, threads=0):
    forest = kplo.Forest(c["root"], c["var"], c["thrs"], c["left"], c["right"], c["value"], c["A"] * c["B"])
    return kplo.detect(c["xyz"], c["nrm"], c["A"], c["B"], c["r"], c["rn"], c["thr"], forest, non_maxima=c["nms"],
                       draws_remove=c["draws"], draws_threshold=c["dthr"],
                       order=kplo.ORDER_SORTED if c["srt"] else kplo.ORDER_CANONICAL)


def write_blob(c, scores, kp, out):
    n = len(c["xyz"])
    with open(out, "wb") as f:
        f.write(b"KPLCASE1")
        f.write(struct.pack("<10i", n, c["A"], c["B"], int(c["nms"]), int(c["draws"]), int(c["srt"]), len(c["root"]),
                            len(c["var"]), c["A"] * c["B"], len(kp)))
        f.write(struct.pack("<4d", c["r"], c["rn"], c["thr"], c["dthr"]))
        for a, t in ((c["xyz"], np.float32), (c["nrm"], np.float32), (c["root"], np.int32), (c["var"], np.int32),
                     (c["thrs"], np.float32), (c["left"], np.int32), (c["right"], np.int32), (c["value"], np.float64),
                     (scores, np.float32), (kp, np.int32)):
            f.write(np.ascontiguousarray(a, dtype=t).tobytes())


def main():
    src, out = sys.argv[1], sys.argv[2]
    c = load_case(src)
    scores, kp = oracle_result(c)
    os.makedirs(os.path.dirname(os.path.abspath(out)), exist_ok=True)
    write_blob(c, scores, kp, out)
    print("%s: %d points, %d x %d, %d trees / %d nodes, oracle: %d keypoints, max score %.4f -> %s"
          % (src, len(c["xyz"]), c["A"], c["B"], len(c["root"]), len(c["var"]), len(kp),
             float(np.nanmax(scores)) if len(scores) else float("nan"), out))


if __name__ == "__main__":
    main()
