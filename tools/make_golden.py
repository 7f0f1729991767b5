"""Generates the committed fixtures under tests/golden/ (run in the build container only).

  pair_kat.json      findAnnulusPair / findBinPair known answers produced by the REFERENCE's own
                     functions (oracle/_ref, compiled from /root/reference/src/KeypointLearning.cpp
                     in place) -- the one part of the path that can be pinned to reference code.
  small_case.npz     seeded 40x40 cloud, forest, and the oracle's features / scores / keypoints
  normals_case.npz   the oracle's normals (k-search and radius search) of that cloud
                     (regression anchor for the oracle and expected values for the HIP path).
  cheff000.npz       the reference's data/point_cloud_test/cheff000.pcd (a data file) with k=10
                     PCA normals and the oracle's keypoints for config 1.
Fixtures are data: inputs and expected outputs.  No reference source text is stored.
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import kplo            # noqa: E402
from tools import cloud_io, forest_yaml, synth  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")
REF = "/root/reference"


def f32(x):
    return float(np.float32(x))


def make_pair_kat():
    if not kplo.RefPairs.available():
        kplo.build(ref=True)
    ref = kplo.RefPairs()
    rows_a, rows_b = [], []
    # annulus: n, support, distances incl. exact edges, 0, just below support
    for n, support in [(5, f32(3.9)), (5, f32(6 * 0.6571)), (8, f32(5.038615)), (1, f32(2.0)), (3, f32(20.0))]:
        dim = np.float32(support) / np.float32(n)
        ds = [0.0, 1e-30, f32(support) * 0.999999, f32(np.nextafter(np.float32(support), np.float32(0)))]
        for k in range(n + 1):
            e = np.float32(k) * dim
            ds += [f32(e), f32(np.nextafter(e, np.float32(0))), f32(np.nextafter(e, np.float32(1e9))),
                   f32(e + dim / 2), f32(e + dim / 4)]
        ds += [f32(v) for v in np.linspace(0, support, 23)]
        for d in ds:
            if 0 <= d <= support * 1.0000001:
                i, p, w = ref.annulus(n, d, support)
                if 0 <= i < n:
                    rows_a.append([n, float(d), float(support), i, p, float(w)])
    for n in (6, 10, 1, 2, 7):
        dim = np.float32(2) / np.float32(n)
        cs = [-0.1, 0.0, 2.0, 2.5, 1.0, 1e-30, 1.9999999]
        for k in range(n + 1):
            e = np.float32(k) * dim
            cs += [f32(e), f32(np.nextafter(e, np.float32(-1))), f32(np.nextafter(e, np.float32(9))),
                   f32(e + dim / 2), f32(e + dim / 3)]
        cs += [f32(v) for v in np.linspace(-0.2, 2.2, 29)]
        for c in cs:
            i, p, w = ref.bin(n, c)
            rows_b.append([n, float(c), i, p, float(w)])
    out = {"provenance": "oracle/_ref/libkpl_ref_pairs.so = /root/reference/src/KeypointLearning.cpp:41-92 "
                         "compiled in place (g++ -O2 -ffp-contract=off, float abs overload)",
           "annulus": rows_a, "bin": rows_b}
    with open(os.path.join(GOLD, "pair_kat.json"), "w") as f:
        json.dump(out, f)
    print("pair_kat.json:", len(rows_a), "annulus rows,", len(rows_b), "bin rows")


def make_small_case():
    xyz, nrm = synth.make_cloud(40, 40, seed=7, nan_points=5, nan_normals=7)
    xyz, nrm = synth.shuffle_cloud(xyz, nrm, 1007)
    mr = kplo.cloud_resolution(xyz)
    out = {"xyz": xyz, "nrm": nrm, "mr": np.float64(mr)}
    q = np.arange(0, len(xyz), len(xyz) // 32)[:32].astype(np.int32)
    out["query"] = q
    r = f32(6 * mr)
    g = kplo.Grid(xyz, r)
    for A, B in ((5, 6), (5, 10), (8, 10)):
        out["feat_%dx%d" % (A, B)] = g.features(nrm, A, B, r, q)
    A, B = 5, 6
    allq = np.arange(len(xyz), dtype=np.int32)
    feat = g.features(nrm, A, B, r, allq)
    ok = np.isfinite(feat).all(axis=1)
    lab = synth.saliency_labels(feat[ok], A, B, keep_fraction=0.2)
    fa = synth.train_extra_trees(feat[ok], lab, ntrees=10, max_depth=8, seed=8)
    forest_yaml.save_forest(fa, os.path.join(GOLD, "small_forest.yaml.gz"))
    of = kplo.Forest(fa.root, fa.var, fa.thr, fa.left, fa.right, fa.value, fa.var_count)
    out["r_feat"], out["r_nms"] = np.float64(r), np.float64(f32(4 * mr))
    for thr in (0.0, 0.5, 0.85):
        for dr in (0, 1):
            sc, kp = kplo.detect(xyz, nrm, A, B, r, f32(4 * mr), f32(thr), of, draws_remove=bool(dr),
                                 draws_threshold=f32(2 * mr))
            out["kp_thr%03d_dr%d" % (int(thr * 100), dr)] = kp
            out["scores"] = sc
    out["draws_threshold"] = np.float64(f32(2 * mr))
    np.savez_compressed(os.path.join(GOLD, "small_case.npz"), **out)
    print("small_case.npz:", len(xyz), "points", {k: len(v) for k, v in out.items() if k.startswith("kp_")})


def make_normals_case():
    """normals of the small case's cloud: k-search 10 (TestDetector, main_test_detector.cpp:162-169) and radius
    search (the detector's fallback, impl/KeypointLearning.hpp:125-148), viewpoint off the origin"""
    z = np.load(os.path.join(GOLD, "small_case.npz"))
    xyz, mr = z["xyz"], float(z["mr"])
    vp = np.float32([3.0, -2.0, 40.0])
    nk, ck = kplo.estimate_normals(xyz, k=10, viewpoint=vp)
    r = f32(3 * mr)
    nr, cr = kplo.estimate_normals(xyz, k=0, radius=r, viewpoint=vp)
    np.savez_compressed(os.path.join(GOLD, "normals_case.npz"), viewpoint=vp, k=np.int32(10), nrm_k=nk, curv_k=ck,
                        radius=np.float64(r), nrm_r=nr, curv_r=cr)
    print("normals_case.npz:", len(xyz), "points,", int(np.isfinite(nk).all(1).sum()), "finite k-normals")


def make_cheff():
    src = os.path.join(REF, "data", "point_cloud_test", "cheff000.pcd")
    xyz = cloud_io.read_pcd_xyz(src)
    nrm = cloud_io.pca_normals(xyz, k=10, flip=True)
    mr = kplo.cloud_resolution(xyz)
    A, B = 5, 6
    r, rn, thr = f32(6 * mr), f32(4 * mr), f32(0.85)
    fa = forest_yaml.load_forest(os.path.join(ROOT, "data", "forests", "synth200k_a5b6_t10.yaml.gz"))
    of = kplo.Forest(fa.root, fa.var, fa.thr, fa.left, fa.right, fa.value, fa.var_count)
    sc, kp = kplo.detect(xyz, nrm, A, B, r, rn, thr, of)
    np.savez_compressed(os.path.join(GOLD, "cheff000.npz"), xyz=xyz, nrm=nrm, mr=np.float64(mr),
                        r_feat=np.float64(r), r_nms=np.float64(rn), thr=np.float64(thr),
                        kp=kp, scores=sc)
    print("cheff000.npz:", len(xyz), "points, mr %.4f, %d keypoints" % (mr, len(kp)),
          os.path.getsize(os.path.join(GOLD, "cheff000.npz")) // 1024, "KiB")


def make_sorted_case():
    """the same two clouds in SORTED-search mode (neighbors in ascending (distance, index) order: what a
    pcl::search::KdTree(true) handed to setSearchMethod gives the feature loop).  This is the file a PCL + OpenCV run
    could regenerate and compare BIT FOR BIT (tests/golden/README.md); until then a regression anchor."""
    z = np.load(os.path.join(GOLD, "small_case.npz"))
    xyz, nrm, q = z["xyz"], z["nrm"], z["query"]
    r, rn = float(z["r_feat"]), float(z["r_nms"])
    g = kplo.Grid(xyz, r)
    out = {}
    for A, B in ((5, 6), (8, 10)):
        out["small_feat_%dx%d" % (A, B)] = g.features(nrm, A, B, r, q, order=kplo.ORDER_SORTED)
    fa = forest_yaml.load_forest(os.path.join(GOLD, "small_forest.yaml.gz"))
    of = kplo.Forest(fa.root, fa.var, fa.thr, fa.left, fa.right, fa.value, fa.var_count)
    sc, kp = kplo.detect(xyz, nrm, 5, 6, r, rn, f32(0.5), of, order=kplo.ORDER_SORTED)
    out["small_scores"], out["small_kp_thr050"] = sc, kp
    c = np.load(os.path.join(GOLD, "cheff000.npz"))
    fa = forest_yaml.load_forest(os.path.join(ROOT, "data", "forests", "synth200k_a5b6_t10.yaml.gz"))
    of = kplo.Forest(fa.root, fa.var, fa.thr, fa.left, fa.right, fa.value, fa.var_count)
    sc, kp = kplo.detect(c["xyz"], c["nrm"], 5, 6, float(c["r_feat"]), float(c["r_nms"]), float(c["thr"]), of,
                         order=kplo.ORDER_SORTED, threads=8)
    out["cheff_scores"], out["cheff_kp"] = sc, kp
    np.savez_compressed(os.path.join(GOLD, "sorted_case.npz"), **out)
    print("sorted_case.npz:", {k: v.shape for k, v in out.items()}, os.path.getsize(os.path.join(GOLD, "sorted_case.npz")) // 1024, "KiB")


if __name__ == "__main__":
    if "--sorted-only" in sys.argv:          # needs only the committed fixtures, not the reference checkout
        make_sorted_case()
        sys.exit(0)
    if not os.path.isdir(REF):
        print("reference checkout absent: nothing to do")
        sys.exit(0)
    os.makedirs(GOLD, exist_ok=True)
    make_pair_kat()
    make_small_case()
    make_normals_case()
    make_cheff()
    make_sorted_case()
