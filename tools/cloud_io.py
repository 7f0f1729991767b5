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


def read_ply_xyz(path):
    """PLY (ascii or binary_little_endian) vertex x y z as float32[n,3] -- the second format the reference's
    TrainDetector accepts (/root/reference/src/main_train_detector.cpp:303-310)."""
    with open(path, "rb") as f:
        raw = f.read()
    end = raw.index(b"end_header")
    line_end = raw.index(b"\n", end)
    header = raw[:line_end].decode("ascii", errors="replace").splitlines()
    fmt, n, props, in_vertex = None, 0, [], False
    sizes = {"char": 1, "uchar": 1, "int8": 1, "uint8": 1, "short": 2, "ushort": 2, "int16": 2, "uint16": 2,
             "int": 4, "uint": 4, "int32": 4, "uint32": 4, "float": 4, "float32": 4, "double": 8, "float64": 8}
    codes = {"char": "i1", "uchar": "u1", "int8": "i1", "uint8": "u1", "short": "<i2", "ushort": "<u2", "int16": "<i2",
             "uint16": "<u2", "int": "<i4", "uint": "<u4", "int32": "<i4", "uint32": "<u4", "float": "<f4",
             "float32": "<f4", "double": "<f8", "float64": "<f8"}
    for ln in header:
        t = ln.split()
        if not t:
            continue
        if t[0] == "format":
            fmt = t[1]
        elif t[0] == "element":
            in_vertex = t[1] == "vertex"
            if in_vertex:
                n = int(t[2])
        elif t[0] == "property" and in_vertex:
            if t[1] == "list":
                raise ValueError("list property in the vertex element")
            props.append((t[2], t[1]))
    names = [p[0] for p in props]
    ix = [names.index(c) for c in ("x", "y", "z")]
    body = raw[line_end + 1:]
    if fmt == "ascii":
        rows = np.loadtxt(body.decode("ascii").splitlines()[:n], dtype=np.float64).reshape(n, len(props))
        return np.ascontiguousarray(rows[:, ix], dtype=np.float32)
    if fmt != "binary_little_endian":
        raise ValueError("unsupported PLY format " + str(fmt))
    dt = np.dtype([(nm, codes[ty]) for nm, ty in props])
    arr = np.frombuffer(body[:n * dt.itemsize], dtype=dt)
    return np.ascontiguousarray(np.stack([arr["x"], arr["y"], arr["z"]], axis=1), dtype=np.float32)


def read_cloud_xyz(path):
    return read_ply_xyz(path) if str(path).lower().endswith(".ply") else read_pcd_xyz(path)


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
