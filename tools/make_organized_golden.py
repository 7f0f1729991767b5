"""tests/golden/organized_case.npz: an organized cloud (WIDTH 160 x HEIGHT 120: a pinhole view of a bumpy surface with a
depth step, NaN holes and a few pixels whose x is not finite) and the normals the detector's own fallback computes on it
when the caller gave none (/root/reference/include/impl/KeypointLearning.hpp:138-145: pcl::IntegralImageNormalEstimation,
SIMPLE_3D_GRADIENT, smoothing size 5, viewpoint = sensor origin), as the ORACLE restates them (kplo.integral_image_normals).
PCL is absent here ("parity unpinned"): this is the array a PCL 1.8 run has to reproduce (tests/golden/README.md item 6);
until then a regression anchor for the oracle and the device kernels alike.
    python tools/make_organized_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import kplo  # noqa: E402
from tests.test_oracle_organized_normals import depth_image  # noqa: E402


def main():
    W, H = 160, 120
    xyz = depth_image(W, H, seed=160 * 120, step=71, holes=25)
    xyz[5 * W + 17, 0] = np.inf            # x not finite, z finite: out of the integral image, into the depth-change map
    xyz[77 * W + 140, 0] = -np.inf
    out = {"xyz": xyz, "width": np.int32(W), "height": np.int32(H), "smoothing": np.float32(5.0)}
    for name, vp in (("origin", (0.0, 0.0, 0.0)), ("off", (0.3, -0.2, -1.0))):
        nrm, curv = kplo.integral_image_normals(xyz, W, H, 5.0, vp)
        assert np.isnan(curv).all()
        out["viewpoint_" + name] = np.float32(vp)
        out["normals_" + name] = nrm
    path = os.path.join(ROOT, "tests", "golden", "organized_case.npz")
    np.savez_compressed(path, **out)
    fin = np.isfinite(out["normals_origin"]).all(axis=1)
    print("%s: %d x %d, %d pixels with a normal, %d without, %d KiB" % (path, W, H, int(fin.sum()), int((~fin).sum()),
                                                                         os.path.getsize(path) // 1024))


if __name__ == "__main__":
    main()
