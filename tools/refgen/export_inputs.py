"""tools/refgen/export_inputs.py [out_dir]: writes every INPUT of the committed fixtures in the formats the reference itself
reads, plus a manifest of the runs whose outputs a machine with PCL 1.8 + OpenCV 3.2 has to regenerate
(tools/refgen/refgen_driver.cpp reads the manifest, tools/refgen/compare.py judges what it wrote).

  <out>/clouds/<name>.pcd            x y z, float32, DATA binary (bit-exact; non-finite points as they are), VIEWPOINT as set
  <out>/clouds/<name>_normals.pcd    normal_x normal_y normal_z curvature, float32, DATA binary
  <out>/clouds/<name>_query.txt      the query indices of the feature rows, one per line
  <out>/forests/*.yaml.gz            the forest files of the tree (tests/golden/small_forest.yaml.gz, data/forests/*)
  <out>/manifest.txt                 one run per line: key=value pairs, radii / thresholds as C99 hex floats of the DOUBLES
                                     the fixtures were made with (a float CLI value promoted to double, as
                                     /root/reference/src/main_test_detector.cpp:113-115,126-130 does)

Reads only committed fixtures (tests/golden/*.npz): nothing under oracle/ is imported, nothing of /root/reference is
needed.  The EXPECTED outputs stay in the .npz files; compare.py maps every run id onto its arrays (EXPECT below)."""
import os
import shutil
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
GOLD = os.path.join(ROOT, "tests", "golden")
FORESTS = {"small_forest.yaml.gz": os.path.join(GOLD, "small_forest.yaml.gz"),
           "synth200k_a5b6_t10.yaml.gz": os.path.join(ROOT, "data", "forests", "synth200k_a5b6_t10.yaml.gz"),
           "cheff_a5b10_t10.yaml.gz": os.path.join(ROOT, "data", "forests", "cheff_a5b10_t10.yaml.gz")}


def write_pcd(path, columns, names, width=None, height=1, viewpoint=(0.0, 0.0, 0.0)):
    """float32 columns -> PCD v0.7, DATA binary (the floats travel as their bits)"""
    arr = np.ascontiguousarray(np.stack([np.asarray(c, dtype=np.float32) for c in columns], axis=1))
    n = len(arr)
    width = n if width is None else width
    assert width * height == n
    head = ("# .PCD v0.7 - Point Cloud Data file format\nVERSION 0.7\nFIELDS %s\nSIZE %s\nTYPE %s\nCOUNT %s\n"
            "WIDTH %d\nHEIGHT %d\nVIEWPOINT %.9g %.9g %.9g 1 0 0 0\nPOINTS %d\nDATA binary\n"
            % (" ".join(names), " ".join("4" for _ in names), " ".join("F" for _ in names), " ".join("1" for _ in names),
               width, height, viewpoint[0], viewpoint[1], viewpoint[2], n))
    with open(path, "wb") as f:
        f.write(head.encode("ascii"))
        f.write(arr.astype("<f4").tobytes())


def hexd(x):
    return float(x).hex()


# run id -> what compare.py holds it against: (npz file, {output kind: array name}, "bitwise" | "distribution" | "tolerance").
# "bitwise" rows are the SORTED-search runs (setSearchMethod(KdTree(true))): every float addition of the histogram is then
# ordered by FLANN's published (distance, index) sort, not by its tree layout.  "distribution" rows use the reference's
# default unsorted tree, whose traversal order this repo cannot know: they are expected to agree statistically only
# (tests/golden/README.md).  "tolerance" rows are the normal estimators (PCL: float one-pass covariance + analytic
# eigen-solver; here: double two-pass + Jacobi).
EXPECT = {}


def rows_small(out, z):
    xyz, nrm = z["xyz"], z["nrm"]
    write_pcd(os.path.join(out, "clouds", "small.pcd"), xyz.T, ["x", "y", "z"])
    write_pcd(os.path.join(out, "clouds", "small_normals.pcd"), list(nrm.T) + [np.zeros(len(nrm), np.float32)],
              ["normal_x", "normal_y", "normal_z", "curvature"])
    np.savetxt(os.path.join(out, "clouds", "small_query.txt"), z["query"], fmt="%d")
    finite = bool(np.isfinite(xyz).all() and np.isfinite(nrm).all())
    base = "cloud=clouds/small.pcd normals=clouds/small_normals.pcd all_finite=%d r_feat=%s r_nms=%s" % (
        finite, hexd(z["r_feat"]), hexd(z["r_nms"]))
    rows = []
    for srt, tag, npz, mode in ((1, "sorted", "sorted_case.npz", "bitwise"), (0, "default", "small_case.npz", "distribution")):
        for A, B in ((5, 6), (5, 10), (8, 10)):
            name = "small_feat_%dx%d" % (A, B) if srt else "feat_%dx%d" % (A, B)
            if srt and (A, B) == (5, 10):
                continue                                     # (sorted_case.npz holds 5x6 and 8x10)
            rid = "small_%s_features_%dx%d" % (tag, A, B)
            rows.append("id=%s mode=features %s annuli=%d bins=%d sorted=%d query=clouds/small_query.txt" % (rid, base, A, B, srt))
            EXPECT[rid] = (npz, {"features": name}, mode)
    # scores + keypoints: 5 x 6 with the small forest
    fbase = base + " annuli=5 bins=6 forest=forests/small_forest.yaml.gz"
    rid = "small_sorted_detect_thr050"
    rows.append("id=%s mode=detect %s sorted=1 thr=%s draws_remove=0 draws_thr=%s" % (
        rid, fbase, hexd(np.float32(0.5)), hexd(z["draws_threshold"])))
    EXPECT[rid] = ("sorted_case.npz", {"scores": "small_scores", "keypoints": "small_kp_thr050"}, "bitwise")
    for thr in (0.0, 0.5, 0.85):
        for dr in (0, 1):
            rid = "small_default_detect_thr%03d_dr%d" % (int(thr * 100), dr)
            rows.append("id=%s mode=detect %s sorted=0 thr=%s draws_remove=%d draws_thr=%s" % (
                rid, fbase, hexd(np.float32(thr)), dr, hexd(z["draws_threshold"])))
            EXPECT[rid] = ("small_case.npz", {"scores": "scores", "keypoints": "kp_thr%03d_dr%d" % (int(thr * 100), dr)}, "distribution")
    return rows


def rows_cheff(out):
    rows = []
    c0 = np.load(os.path.join(GOLD, "cheff000.npz"))
    c1 = np.load(os.path.join(GOLD, "cheff001.npz"))
    for name, z in (("cheff000", c0), ("cheff001", c1)):
        write_pcd(os.path.join(out, "clouds", name + ".pcd"), z["xyz"].T, ["x", "y", "z"])
        write_pcd(os.path.join(out, "clouds", name + "_normals.pcd"), list(z["nrm"].T) + [np.zeros(len(z["nrm"]), np.float32)],
                  ["normal_x", "normal_y", "normal_z", "curvature"])
    # config 1 (6 mr / 4 mr / 0.85, 5 x 6) on cheff000
    b0 = ("cloud=clouds/cheff000.pcd normals=clouds/cheff000_normals.pcd all_finite=1 annuli=5 bins=6 "
          "forest=forests/synth200k_a5b6_t10.yaml.gz r_feat=%s r_nms=%s thr=%s draws_remove=0 draws_thr=%s" % (
              hexd(c0["r_feat"]), hexd(c0["r_nms"]), hexd(c0["thr"]), hexd(0.0)))
    rows.append("id=cheff000_sorted_detect mode=detect %s sorted=1" % b0)
    EXPECT["cheff000_sorted_detect"] = ("sorted_case.npz", {"scores": "cheff_scores", "keypoints": "cheff_kp"}, "bitwise")
    rows.append("id=cheff000_default_detect mode=detect %s sorted=0" % b0)
    EXPECT["cheff000_default_detect"] = ("cheff000.npz", {"scores": "scores", "keypoints": "kp"}, "distribution")
    # the reference main's own defaults on its default cloud (src/main_test_detector.cpp:62-67,105-106)
    b1 = ("cloud=clouds/cheff001.pcd normals=clouds/cheff001_normals.pcd all_finite=1 annuli=5 bins=10 "
          "forest=forests/cheff_a5b10_t10.yaml.gz r_feat=%s r_nms=%s thr=%s draws_remove=0 draws_thr=%s" % (
              hexd(c1["r_feat"]), hexd(c1["r_nms"]), hexd(c1["thr"]), hexd(0.0)))
    rows.append("id=cheff001_sorted_detect mode=detect %s sorted=1" % b1)
    EXPECT["cheff001_sorted_detect"] = ("cheff001.npz", {"scores": "scores_sorted", "keypoints": "kp_sorted"}, "bitwise")
    rows.append("id=cheff001_default_detect mode=detect %s sorted=0" % b1)
    EXPECT["cheff001_default_detect"] = ("cheff001.npz", {"scores": "scores_canonical", "keypoints": "kp_canonical"}, "distribution")
    return rows


def rows_normals(out, small):
    z = np.load(os.path.join(GOLD, "normals_case.npz"))
    vp = z["viewpoint"]
    write_pcd(os.path.join(out, "clouds", "small_vp.pcd"), small["xyz"].T, ["x", "y", "z"], viewpoint=vp)
    base = "cloud=clouds/small_vp.pcd all_finite=0 viewpoint=%s,%s,%s" % (hexd(vp[0]), hexd(vp[1]), hexd(vp[2]))
    rows = ["id=small_normals_k10 mode=normals_k %s k=%d" % (base, int(z["k"])),
            "id=small_normals_radius mode=normals_radius %s r_feat=%s" % (base, hexd(z["radius"]))]
    EXPECT["small_normals_k10"] = ("normals_case.npz", {"normals": "nrm_k", "curvature": "curv_k"}, "tolerance")
    EXPECT["small_normals_radius"] = ("normals_case.npz", {"normals": "nrm_r", "curvature": "curv_r"}, "tolerance")
    return rows


def rows_organized(out):
    z = np.load(os.path.join(GOLD, "organized_case.npz"))
    w, h = int(z["width"]), int(z["height"])
    rows = []
    for tag in ("origin", "off"):
        vp = z["viewpoint_" + tag]
        write_pcd(os.path.join(out, "clouds", "organized_%s.pcd" % tag), z["xyz"].T, ["x", "y", "z"], width=w, height=h, viewpoint=vp)
        rid = "organized_normals_" + tag
        rows.append("id=%s mode=normals_organized cloud=clouds/organized_%s.pcd all_finite=0 smoothing=%s viewpoint=%s,%s,%s" % (
            rid, tag, hexd(z["smoothing"]), hexd(vp[0]), hexd(vp[1]), hexd(vp[2])))
        EXPECT[rid] = ("organized_case.npz", {"normals": "normals_" + tag}, "bitwise_or_one_ulp")
    return rows


def export(out):
    for sub in ("clouds", "forests", "results"):
        os.makedirs(os.path.join(out, sub), exist_ok=True)
    for name, src in FORESTS.items():
        shutil.copyfile(src, os.path.join(out, "forests", name))
    small = np.load(os.path.join(GOLD, "small_case.npz"))
    rows = rows_small(out, small) + rows_cheff(out) + rows_normals(out, small) + rows_organized(out)
    with open(os.path.join(out, "manifest.txt"), "w") as f:
        f.write("# refgen manifest: one run per line, key=value; doubles as C99 hex floats (strtod reads them)\n")
        for r in rows:
            f.write(r + "\n")
    return rows


def expectations():
    """run id -> (npz, {kind: array}, mode) without writing anything (compare.py)"""
    if not EXPECT:
        import tempfile
        with tempfile.TemporaryDirectory() as tmp:
            export(tmp)
    return dict(EXPECT)


if __name__ == "__main__":
    out = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "build", "refgen")
    rows = export(out)
    print("%d runs -> %s/manifest.txt" % (len(rows), out))
