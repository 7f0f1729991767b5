"""PCD reader (ascii / binary, fields x y z float32) and k-NN PCA normals -- tooling for fixtures.

The TestDetector of the reference loads a PCD and estimates normals with k = 10 before calling
the detector (/root/reference/src/main_test_detector.cpp:143,162-169); this is the tooling-side
equivalent used to prepare fixtures, not part of the accelerated path.
"""
import numpy as np


def read_pcd_xyz(path):
    with open(path, "rb") as f:
        raw = f.read()
    header_end = raw.index(b"DATA")
    line_end = raw.index(b"\n", header_end)
    header = raw[:line_end].decode("ascii", errors="replace").splitlines()
    meta = {ln.split()[0]: ln.split()[1:] for ln in header if ln and not ln.startswith("#")}
    fields, n = meta["FIELDS"], int(meta["POINTS"][0])
    sizes = [int(s) for s in meta["SIZE"]]
    kind = meta["DATA"][0]
    body = raw[line_end + 1:]
    if kind == "ascii":
        arr = np.loadtxt(body.decode("ascii").splitlines(), dtype=np.float32).reshape(n, len(fields))
    elif kind == "binary":
        assert all(s == 4 for s in sizes)
        arr = np.frombuffer(body[:n * 4 * len(fields)], dtype=np.float32).reshape(n, len(fields))
    else:
        raise ValueError("unsupported PCD DATA " + kind)
    ix = [fields.index(c) for c in ("x", "y", "z")]
    return np.ascontiguousarray(arr[:, ix], dtype=np.float32)


def pca_normals(xyz, k=10, viewpoint=(0.0, 0.0, 0.0), flip=False):
    """Normals = eigenvector of the smallest eigenvalue of the k-NN covariance, oriented toward the
    viewpoint (pcl::NormalEstimation semantics, tolerance-level only), optionally flipped like
    TestDetector's --flipNormals."""
    from scipy.spatial import cKDTree
    x = xyz.astype(np.float64)
    _, idx = cKDTree(x).query(x, k=k)
    nb = x[idx]
    c = nb - nb.mean(axis=1, keepdims=True)
    cov = np.einsum("nki,nkj->nij", c, c) / k
    w, v = np.linalg.eigh(cov)
    nrm = v[:, :, 0]
    to_vp = np.asarray(viewpoint)[None, :] - x
    sgn = np.where((nrm * to_vp).sum(axis=1) < 0, -1.0, 1.0)
    nrm = nrm * sgn[:, None]
    if flip:
        nrm = -nrm
    return nrm.astype(np.float32)
