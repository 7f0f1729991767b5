"""tools/refgen/compare.py <dir> [--explain] [--json]: holds what tools/refgen/refgen_driver wrote into <dir>/results/ against the
expected arrays of tests/golden/*.npz, array by array and bit by bit.

Per array: elements compared, identical bits, first mismatch (index, both values, distance in ulps), a histogram of the ulp
distances.  Per run a verdict; at the end the lines a reader is after:

  * "RTrees::load accepted the file: yes/no" per forest file (the YAML dialect of csrc/forest.cpp and tools/forest_yaml.py is
    restated from memory of OpenCV 3.x; no sample forest survives in the reference checkout);
  * for the SORTED-search runs (the only ones that can agree in every bit, tests/golden/README.md): pinned or not;
  * with --explain, for feature rows that differ: which of the arithmetic choices DESIGN.md section 2 had to make without
    Eigen at hand reproduces the reference's bits -- row normalisation by true division or by multiplication with the
    reciprocal (Eigen 3.2 vs 3.3), the dot product as x + (y + z) or (x + y) + z, the squared norm summed forwards or pairwise.
    The explainer is a second, independent restatement of hpp:321-376 / cpp:41-92 in numpy float32 (brute-force sorted
    neighbors, no grid), so it also cross-checks the fixtures themselves.

  python tools/refgen/compare.py --self-test    feeds the comparison the fixtures' OWN arrays laid out as a reference run would
                                                write them (must come out identical), then a one-ulp perturbation and a row
                                                normalised with the reciprocal (must be found and named).
Test infrastructure: nothing here is imported by the product."""
import json
import os
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
import export_inputs  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")
F32 = np.float32


# ------------------------------------------------------------------------------------------------ bit comparison
def ulp_distance(a, b):
    """distance in units in the last place between float32 arrays (NaN == NaN: 0; NaN vs number: 2^31)"""
    a, b = np.ascontiguousarray(a, F32), np.ascontiguousarray(b, F32)
    ia, ib = a.view(np.int32).astype(np.int64), b.view(np.int32).astype(np.int64)
    ia = np.where(ia < 0, -(ia & 0x7fffffff), ia)          # sign-magnitude -> a line
    ib = np.where(ib < 0, -(ib & 0x7fffffff), ib)
    d = np.abs(ia - ib)
    na, nb = np.isnan(a), np.isnan(b)
    d = np.where(na & nb, 0, d)
    d = np.where(na ^ nb, 1 << 31, d)
    return d


def compare_f32(name, got, want):
    got, want = np.asarray(got, F32).ravel(), np.asarray(want, F32).ravel()
    rep = {"array": name, "expected_len": int(want.size), "got_len": int(got.size)}
    if got.size != want.size:
        rep.update(identical=False, note="length differs")
        return rep
    d = ulp_distance(got, want)
    same = (got.view(np.uint32) == want.view(np.uint32)) | (np.isnan(got) & np.isnan(want))
    rep["bit_identical"] = int(same.sum())
    rep["identical"] = bool(same.all())
    edges = [0, 1, 2, 3, 5, 17, 1 << 31, (1 << 31) + 1]
    hist = np.histogram(d, bins=edges)[0]
    rep["ulp_histogram"] = dict(zip(["0", "1", "2", "3-4", "5-16", ">16", "nan_vs_number"], [int(v) for v in hist]))
    if not rep["identical"]:
        k = int(np.argmin(same))
        rep["first_mismatch"] = {"index": k, "got": float(got[k]), "got_hex": float(got[k]).hex(),
                                 "expected": float(want[k]), "expected_hex": float(want[k]).hex(), "ulps": int(d[k])}
        rep["max_ulps"] = int(d[d < (1 << 31)].max()) if (d < (1 << 31)).any() else None
    return rep


def compare_i32(name, got, want):
    got, want = np.asarray(got, np.int32).ravel(), np.asarray(want, np.int32).ravel()
    rep = {"array": name, "expected_len": int(want.size), "got_len": int(got.size),
           "identical": bool(got.size == want.size and (got == want).all())}
    if not rep["identical"]:
        sg, sw = set(got.tolist()), set(want.tolist())
        rep["common"] = len(sg & sw)
        rep["only_reference"] = sorted(sg - sw)[:8]
        rep["only_fixture"] = sorted(sw - sg)[:8]
    return rep


def read_raw(path, dtype):
    return np.fromfile(path, dtype=dtype) if os.path.exists(path) else None


# ------------------------------------------------------------------------ second restatement (numpy float32), with switches
def soft_pair(n, v, dim):
    """the shared body of findAnnulusPair / findBinPair (src/KeypointLearning.cpp:43-64, :75-91), float32 op by op"""
    k = int(np.floor(v / dim))
    if k == n:
        k -= 1
    ctr = F32(k) * dim + dim / F32(2)
    w = (v - ctr) / dim
    k2 = k + 1 if w > 0 else k - 1
    if k2 == -1:
        k2 = 0
    if k2 == n:
        k2 = k
    return k, k2, F32(abs(w))


def feature_row(xyz, nrm, i, A, B, r_feat, sorted_order=True, normalise="divide", dot="x+(y+z)", norm_sum="forward"):
    """hpp:321-376 for point i with brute-force neighbors; only the sorted order is defined without the engine's grid"""
    assert sorted_order
    p = xyz[i]
    R2 = F32(float(r_feat) * float(r_feat))                        # (float)(r * r), the product in double (PCL)
    dx, dy, dz = p[0] - xyz[:, 0], p[1] - xyz[:, 1], p[2] - xyz[:, 2]
    with np.errstate(invalid="ignore"):
        d2 = (dx * dx + dy * dy) + dz * dz                          # float32: ((dx dx) + dy dy) + dz dz
        inside = np.flatnonzero(d2 < R2)                            # strict; non-finite points drop out (NaN < x is false)
    order = inside[np.lexsort((inside, d2[inside]))]                # FLANN: ascending (distance, index)
    support = F32(r_feat)
    adim, bdim = support / F32(A), F32(2) / F32(B)
    H = np.zeros((A, B), F32)
    n_p = nrm[i]
    for j in order[1:]:                                             # hpp:336: element 0 is dropped
        n_q = nrm[j]
        if not np.isfinite(n_q).all():
            continue
        if dot == "x+(y+z)":
            dp = n_p[0] * n_q[0] + (n_p[1] * n_q[1] + n_p[2] * n_q[2])
        else:
            dp = (n_p[0] * n_q[0] + n_p[1] * n_q[1]) + n_p[2] * n_q[2]
        cosine = F32(1) - dp
        a, a2, aw = soft_pair(A, np.sqrt(d2[j]), adim)
        c = min(max(cosine, F32(0)), F32(2))
        b, b2, bw = soft_pair(B, c, bdim)
        H[a, b] += (F32(1) - bw) * (F32(1) - aw)
        H[a, b2] += bw * (F32(1) - aw)
        H[a2, b] += (F32(1) - bw) * aw
        H[a2, b2] += bw * aw
    for a in range(A):
        sq = H[a] * H[a]
        if norm_sum == "forward":
            s = F32(0)
            for v in sq:
                s = s + v
        else:                                                       # pairwise halves, as a vectorised reduction would add
            v = sq.copy()
            while len(v) > 1:
                if len(v) % 2:
                    v = np.append(v, F32(0))
                v = v[0::2] + v[1::2]
            s = v[0]
        nm = np.sqrt(s)
        if nm > 0:
            H[a] = H[a] / nm if normalise == "divide" else H[a] * (F32(1) / nm)
    return H.reshape(-1)


HYPOTHESES = [{"normalise": n, "dot": d, "norm_sum": s}
              for n in ("divide", "reciprocal") for d in ("x+(y+z)", "(x+y)+z") for s in ("forward", "pairwise")]


def explain_features(row, got, limit=6):
    """which arithmetic choice reproduces the reference's differing feature rows"""
    z = np.load(os.path.join(GOLD, "small_case.npz"))
    xyz, nrm, query = z["xyz"], z["nrm"], z["query"]
    A, B, r = int(row["annuli"]), int(row["bins"]), float.fromhex(row["r_feat"])
    F = A * B
    got = np.asarray(got, F32).reshape(-1, F)
    votes, checked = {}, 0
    for k, i in enumerate(query):
        if checked >= limit:
            break
        base = feature_row(xyz, nrm, int(i), A, B, r)
        if np.array_equal(base.view(np.uint32), got[k].view(np.uint32)):
            continue
        checked += 1
        for h in HYPOTHESES:
            if np.array_equal(feature_row(xyz, nrm, int(i), A, B, r, **h).view(np.uint32), got[k].view(np.uint32)):
                key = json.dumps(h, sort_keys=True)
                votes[key] = votes.get(key, 0) + 1
    return {"rows_examined": checked, "reproduced_by": votes or "none of the %d combinations" % len(HYPOTHESES)}


# --------------------------------------------------------------------------------------------------- per-run judgement
def manifest_rows(d):
    rows = {}
    with open(os.path.join(d, "manifest.txt")) as f:
        for line in f:
            if line.strip() and not line.startswith("#"):
                kv = dict(t.split("=", 1) for t in line.split() if "=" in t)
                rows[kv["id"]] = kv
    return rows


def summary_lines(d):
    out = {}
    p = os.path.join(d, "results", "summary.txt")
    if os.path.exists(p):
        for line in open(p):
            t = line.split()
            if len(t) >= 2:
                out.setdefault(t[0], []).append(" ".join(t[1:]))
    return out


def angle_stats(got, want):
    got, want = np.asarray(got, np.float64).reshape(-1, 3), np.asarray(want, np.float64).reshape(-1, 3)
    fin = np.isfinite(got).all(1) & np.isfinite(want).all(1)
    c = np.clip((got[fin] * want[fin]).sum(1) / np.maximum(np.linalg.norm(got[fin], axis=1) * np.linalg.norm(want[fin], axis=1), 1e-30), -1, 1)
    ang = np.arccos(c)
    return {"nan_pattern_equal": bool((np.isfinite(got).all(1) == np.isfinite(want).all(1)).all()), "compared": int(fin.sum()),
            "max_angle_rad": float(ang.max()) if ang.size else 0.0, "median_angle_rad": float(np.median(ang)) if ang.size else 0.0,
            "flipped": int((c < 0).sum())}


def judge(d, explain=False):
    expect = export_inputs.expectations()
    rows, summ = manifest_rows(d), summary_lines(d)
    report = {"engine": (summ.get("#") or ["unknown"])[0], "runs": {}, "forest_files": {}}
    for rid, (npz, arrays, mode) in sorted(expect.items()):
        row = rows.get(rid)
        if row is None:
            continue
        z = np.load(os.path.join(GOLD, npz))
        res = os.path.join(d, "results", rid)
        run = {"mode": mode, "fixture": npz, "arrays": []}
        for line in summ.get(rid, []):
            if line.startswith("forest_accepted="):
                report["forest_files"][row.get("forest", "?")] = line.split("=")[1]
            if line.startswith("error="):
                run["error"] = line
        missing = False
        for kind, name in arrays.items():
            want = z[name]
            if kind == "scores":
                got = read_raw(res + ".scores.f32", F32)
                if got is None:
                    missing = True
                    continue
                # runForest skips points with a non-finite xyz or normal (hpp:277): the reference's response cloud is
                # compacted; the fixture keeps the input index space with NaN in those places
                rep = compare_f32(name, got, want[np.isfinite(want)])
                if mode == "distribution" and not rep.get("identical", False) and got.size == np.isfinite(want).sum():
                    w = want[np.isfinite(want)]
                    rep["equal_scores_fraction"] = float((got == w).mean())
                    rep["mean_abs_difference"] = float(np.abs(got - w).mean())
                    rep["mean_score"] = {"reference": float(got.mean()), "fixture": float(w.mean())}
            elif kind == "keypoints":
                got = read_raw(res + ".keypoints.i32", np.int32)
                if got is None:
                    missing = True
                    continue
                rep = compare_i32(name, got, want)
                if row.get("all_finite") == "0":
                    rep["note"] = ("the cloud holds non-finite points: the reference indexes its COMPACTED response with tree "
                                   "indices (hpp:213 vs :277), this repo keeps the input index space -- documented divergence")
            elif kind == "features":
                got = read_raw(res + ".features.f32", F32)
                if got is None:
                    missing = True
                    continue
                rep = compare_f32(name, got, want)
                if explain and mode == "bitwise" and not rep["identical"] and got.size == want.size:
                    rep["explain"] = explain_features(row, got)
            elif kind in ("normals", "curvature"):
                got = read_raw(res + "." + kind + ".f32", F32)
                if got is None:
                    missing = True
                    continue
                rep = compare_f32(name, got, want)
                if kind == "normals" and got.size == want.size:
                    rep["angles"] = angle_stats(got, want)
            run["arrays"].append(rep)
        if missing and not run["arrays"]:
            continue                                                   # run not made (driver given a subset)
        ok = all(a.get("identical") for a in run["arrays"]) and not missing
        if mode == "bitwise":
            run["verdict"] = "PINNED: identical in every bit" if ok else "DIFFERS"
        elif mode == "bitwise_or_one_ulp":
            worst = max([a.get("max_ulps") or 0 for a in run["arrays"]] + [0])
            nanok = all(a["ulp_histogram"].get("nan_vs_number", 0) == 0 for a in run["arrays"] if "ulp_histogram" in a)
            run["verdict"] = ("PINNED: identical in every bit" if ok else
                              "within one ulp, same NaN pattern (Eigen 3.2's /= multiplies by the reciprocal)" if worst <= 1 and nanok
                              else "DIFFERS")
        elif mode == "tolerance":
            ang = [a["angles"] for a in run["arrays"] if "angles" in a]
            run["verdict"] = ("identical" if ok else "tolerance: max angle %.3g rad, NaN pattern %s" % (
                ang[0]["max_angle_rad"], "equal" if ang[0]["nan_pattern_equal"] else "DIFFERS")) if ang else "no normals"
        else:
            run["verdict"] = "identical (unexpected for an unsorted tree: FLANN's order equals the canonical one here)" if ok else \
                "differs, as expected for the unsorted default tree (statistics above)"
        report["runs"][rid] = run
    bit_runs = [r for r in report["runs"].values() if r["mode"] == "bitwise"]
    report["sorted_runs_pinned"] = "%d of %d" % (sum(r["verdict"].startswith("PINNED") for r in bit_runs), len(bit_runs))
    return report


def print_report(rep):
    print("engine:", rep["engine"])
    for rid, run in rep["runs"].items():
        print("%-40s [%s] %s" % (rid, run["mode"], run["verdict"]))
        for a in run["arrays"]:
            line = "    %-18s %d / %d identical" % (a["array"], a.get("bit_identical", a["got_len"] if a["identical"] else -1), a["expected_len"]) \
                if "bit_identical" in a else "    %-18s %s (%d vs %d entries)" % (a["array"], "identical" if a["identical"] else "differs", a["got_len"], a["expected_len"])
            if "first_mismatch" in a:
                m = a["first_mismatch"]
                line += "; first mismatch [%d] %s vs %s (%d ulps); ulps %s" % (m["index"], m["got_hex"], m["expected_hex"], m["ulps"], a["ulp_histogram"])
            print(line)
            for extra in ("explain", "angles", "note", "equal_scores_fraction"):
                if extra in a:
                    print("        %s: %s" % (extra, a[extra]))
    for f, v in rep["forest_files"].items():
        print("RTrees::load accepted the file %s: %s" % (f, v))
    print("sorted-search runs pinned:", rep["sorted_runs_pinned"])
    ex = [a["explain"] for r in rep["runs"].values() for a in r["arrays"] if "explain" in a]
    if ex:
        print("Eigen normalize(): reciprocal or division? ->", ex)
    elif all(r["verdict"].startswith("PINNED") for r in rep["runs"].values() if r["mode"] == "bitwise" and any("feat" in a["array"] for a in r["arrays"])):
        print("Eigen normalize(): the feature rows agree in every bit with TRUE DIVISION, x + (y + z), forward sums (DESIGN.md section 2)")


# --------------------------------------------------------------------------------------------------------- self-test
def fake_reference_results(d, only_bitwise=False):
    """lays the fixtures' own expected arrays out the way refgen_driver writes a reference run"""
    expect = export_inputs.expectations()
    with open(os.path.join(d, "results", "summary.txt"), "w") as s:
        s.write("# engine=self-test (the fixtures' own arrays)\n")
        for rid, (npz, arrays, mode) in expect.items():
            if only_bitwise and mode != "bitwise":
                continue
            z = np.load(os.path.join(GOLD, npz))
            for kind, name in arrays.items():
                a = z[name]
                if kind == "scores":
                    a[np.isfinite(a)].astype("<f4").tofile(os.path.join(d, "results", rid + ".scores.f32"))
                    s.write("%s forest_accepted=yes\n" % rid)
                elif kind == "keypoints":
                    a.astype("<i4").tofile(os.path.join(d, "results", rid + ".keypoints.i32"))
                else:
                    a.astype("<f4").tofile(os.path.join(d, "results", "%s.%s.f32" % (rid, kind)))


def self_test():
    with tempfile.TemporaryDirectory() as d:
        export_inputs.export(d)
        fake_reference_results(d)
        rep = judge(d, explain=True)
        bad = [rid for rid, r in rep["runs"].items() if not all(a["identical"] for a in r["arrays"])]
        assert not bad, "the fixtures' own arrays do not compare identical: %s" % bad
        assert rep["sorted_runs_pinned"].split()[0] == rep["sorted_runs_pinned"].split()[2] != "0", rep["sorted_runs_pinned"]
        # (1) the numpy restatement reproduces the fixture's sorted feature rows bit for bit (it never saw the oracle)
        z, s = np.load(os.path.join(GOLD, "small_case.npz")), np.load(os.path.join(GOLD, "sorted_case.npz"))
        for (A, B) in ((5, 6), (8, 10)):
            for k in (0, 7, 19, 31):
                mine = feature_row(z["xyz"], z["nrm"], int(z["query"][k]), A, B, float(z["r_feat"]))
                want = s["small_feat_%dx%d" % (A, B)][k]
                assert np.array_equal(mine.view(np.uint32), want.view(np.uint32)), ("numpy restatement differs from the fixture", A, B, k)
        # (2) a one-ulp perturbation is found, located and measured
        p = os.path.join(d, "results", "cheff000_sorted_detect.scores.f32")
        a = np.fromfile(p, F32)
        a[1234] = np.nextafter(a[1234], F32(2))
        a.tofile(p)
        r2 = judge(d)["runs"]["cheff000_sorted_detect"]
        sc = [x for x in r2["arrays"] if x["array"] == "cheff_scores"][0]
        assert r2["verdict"] == "DIFFERS" and sc["first_mismatch"]["index"] == 1234 and sc["first_mismatch"]["ulps"] == 1, sc
        # (3) rows normalised with the reciprocal are recognised as such
        rid = "small_sorted_features_5x6"
        rows = [feature_row(z["xyz"], z["nrm"], int(i), 5, 6, float(z["r_feat"]), normalise="reciprocal") for i in z["query"]]
        np.asarray(rows, F32).tofile(os.path.join(d, "results", rid + ".features.f32"))
        r3 = judge(d, explain=True)["runs"][rid]
        ex = r3["arrays"][0].get("explain")
        assert r3["verdict"] == "DIFFERS" and ex and isinstance(ex["reproduced_by"], dict), r3
        assert all(json.loads(k)["normalise"] == "reciprocal" for k in ex["reproduced_by"]), ex
        # (4) a missing results directory entry is "not made", never "identical"
        os.remove(os.path.join(d, "results", "cheff001_sorted_detect.scores.f32"))
        os.remove(os.path.join(d, "results", "cheff001_sorted_detect.keypoints.i32"))
        assert "cheff001_sorted_detect" not in judge(d)["runs"]
    print("compare.py self-test ok")
    return 0


if __name__ == "__main__":
    if "--self-test" in sys.argv:
        sys.exit(self_test())
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    if not args:
        print(__doc__)
        sys.exit(2)
    rep = judge(args[0], explain="--explain" in sys.argv)
    if "--json" in sys.argv:
        print(json.dumps(rep))
    else:
        print_report(rep)
    bit = [r for r in rep["runs"].values() if r["mode"] == "bitwise"]
    sys.exit(0 if bit and all(r["verdict"].startswith("PINNED") for r in bit) else 1)
