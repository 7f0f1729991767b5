"""The TestDetector counterpart (C++, include/KeypointLearning.h facade over the C-ABI) end to end
on a PCD file: same keypoints as the committed fixture."""
import json
import os
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "keypoint-learning_amd", "TestDetector")
GOLD = os.path.join(ROOT, "tests", "golden")


def write_pcd(path, xyz, nrm=None, binary=True):
    fields = "x y z" + (" normal_x normal_y normal_z" if nrm is not None else "")
    k = 6 if nrm is not None else 3
    arr = np.concatenate([xyz, nrm], axis=1).astype(np.float32) if nrm is not None else xyz.astype(np.float32)
    hdr = ("# .PCD v0.7 - Point Cloud Data file format\nVERSION 0.7\nFIELDS %s\nSIZE %s\nTYPE %s\nCOUNT %s\n"
           "WIDTH %d\nHEIGHT 1\nVIEWPOINT 0 0 0 1 0 0 0\nPOINTS %d\nDATA %s\n"
           % (fields, " ".join(["4"] * k), " ".join(["F"] * k), " ".join(["1"] * k), len(arr), len(arr),
              "binary" if binary else "ascii"))
    with open(path, "wb") as f:
        f.write(hdr.encode())
        if binary:
            f.write(arr.tobytes())
        else:
            for row in arr:
                f.write((" ".join("%.9g" % v for v in row) + "\n").encode())


def lzf_compress(data):
    """Small greedy LZF encoder (test helper): hash of 3 bytes -> last position, literals otherwise."""
    out, lit, table, i, n = bytearray(), bytearray(), {}, 0, len(data)

    def flush():
        for k in range(0, len(lit), 32):
            chunk = lit[k:k + 32]
            out.append(len(chunk) - 1)
            out.extend(chunk)
        lit.clear()
    while i < n:
        key = bytes(data[i:i + 3])
        ref = table.get(key) if len(key) == 3 else None
        if len(key) == 3:
            table[key] = i
        if ref is not None and 0 < i - ref <= 8192:
            length = 3
            while i + length < n and length < 264 and data[ref + length] == data[i + length]:
                length += 1
            flush()
            dist, l2 = i - ref - 1, length - 2
            if l2 < 7:
                out.append((l2 << 5) | (dist >> 8))
            else:
                out.append((7 << 5) | (dist >> 8))
                out.append(l2 - 7)
            out.append(dist & 0xff)
            i += length
        else:
            lit.append(data[i])
            i += 1
    flush()
    return bytes(out)


def write_pcd_compressed(path, xyz, nrm):
    import struct
    arr = np.concatenate([xyz, nrm], axis=1).astype(np.float32)
    payload = np.ascontiguousarray(arr.T).tobytes()             # field-major
    comp = lzf_compress(payload)
    hdr = ("# .PCD v0.7 - Point Cloud Data file format\nVERSION 0.7\nFIELDS x y z normal_x normal_y normal_z\nSIZE 4 4 4 4 4 4\n"
           "TYPE F F F F F F\nCOUNT 1 1 1 1 1 1\nWIDTH %d\nHEIGHT 1\nVIEWPOINT 0 0 0 1 0 0 0\nPOINTS %d\nDATA binary_compressed\n"
           % (len(arr), len(arr)))
    with open(path, "wb") as f:
        f.write(hdr.encode())
        f.write(struct.pack("<II", len(comp), len(payload)))
        f.write(comp)
    return len(comp), len(payload)


def test_cli_reads_binary_compressed_pcd(tmp_path):
    z = np.load(os.path.join(GOLD, "cheff000.npz"))
    n = 12000
    xyz, nrm = z["xyz"][:n].copy(), z["nrm"][:n].copy()
    xyz[100:4000] = xyz[100]                                     # long repeats: exercises back references
    a, b = tmp_path / "c.pcd", tmp_path / "b.pcd"
    csize, usize = write_pcd_compressed(a, xyz, nrm)
    assert csize < usize
    write_pcd(b, xyz, nrm, True)
    outs = []
    for pcd in (a, b):
        out = tmp_path / (pcd.name + ".kp")
        cmd = [EXE, "--pathCloud", str(pcd), "--pathRF", os.path.join(ROOT, "data", "forests", "synth200k_a5b6_t10.yaml.gz"),
               "--pathKP=%s" % out, "--radiusFeatures", "%.9g" % float(z["r_feat"]), "--radiusNMS", "%.9g" % float(z["r_nms"]),
               "-t", "0.85", "--annuli", "5", "--bins", "6", "--json"]
        res = subprocess.run(cmd, capture_output=True, text=True, timeout=300)
        assert res.returncode == 0, res.stderr
        outs.append(open(out).read())
    assert outs[0] == outs[1] and outs[0].count("\n") > 12
    bad = tmp_path / "bad.pcd"
    raw = open(a, "rb").read()
    open(bad, "wb").write(raw[:-50])
    res = subprocess.run([EXE, "--pathCloud", str(bad), "--pathRF", os.path.join(GOLD, "small_forest.yaml.gz")],
                         capture_output=True, text=True, timeout=300)
    assert res.returncode != 0 and "corrupt binary_compressed" in res.stderr


@pytest.mark.parametrize("binary", [True, False])
def test_cli_on_cheff_view(tmp_path, binary):
    assert os.path.exists(EXE), "TestDetector is not built (python keypoint-learning_amd/build.py)"
    z = np.load(os.path.join(GOLD, "cheff000.npz"))
    n = 20000 if not binary else len(z["xyz"])          # keep the ascii file small
    pcd, out = tmp_path / "view.pcd", tmp_path / "kp.pcd"
    write_pcd(pcd, z["xyz"][:n], z["nrm"][:n], binary)
    cmd = [EXE, "--pathCloud", str(pcd), "--pathRF", os.path.join(ROOT, "data", "forests", "synth200k_a5b6_t10.yaml.gz"),
           "--pathKP=%s" % out, "--radiusFeatures", "%.9g" % float(z["r_feat"]), "--radiusNMS", "%.9g" % float(z["r_nms"]),
           "-t", "0.85", "--annuli", "5", "--bins", "6", "--json"]
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=300)
    assert res.returncode == 0, res.stderr
    info = json.loads(res.stdout.strip().splitlines()[-1])
    assert info["points"] == n
    kp = np.loadtxt(out, skiprows=11, dtype=np.float32).reshape(-1, 4)
    assert info["keypoints"] == len(kp)
    if binary:
        assert len(kp) == len(z["kp"])
        assert np.array_equal(kp[:, :3], z["xyz"][z["kp"]])
        assert np.array_equal(kp[:, 3], z["scores"][z["kp"]])


def test_cli_estimates_normals_and_resolution(tmp_path):
    """No normals in the file: the CLI gets them from kpl_estimate_normals (k = 10) like the reference main and
    the radii from kpl_cloud_resolution; the whole run equals the oracle pipeline on the same file."""
    from oracle import kplo
    from tests import helpers
    from tools import forest_yaml
    z = np.load(os.path.join(GOLD, "small_case.npz"))
    xyz = np.ascontiguousarray(z["xyz"][np.isfinite(z["xyz"]).all(axis=1)])
    pcd, out = tmp_path / "small.pcd", tmp_path / "kp.pcd"
    write_pcd(pcd, xyz, None, True)
    forest = os.path.join(GOLD, "small_forest.yaml.gz")
    cmd = [EXE, "--pathCloud", str(pcd), "--pathRF", forest, "--radiusFeatures", "6", "--pathKP=%s" % out,
           "--radiusNMS", "4", "--radiusInMr", "--annuli", "5", "--bins", "6", "-t", "0.5", "--flipNormals", "--json"]
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=300)
    assert res.returncode == 0, res.stderr
    info = json.loads(res.stdout.strip().splitlines()[-1])
    mr = kplo.cloud_resolution(xyz)
    assert info["mr"] == pytest.approx(mr, rel=1e-8)          # printed with 9 digits
    nrm, _ = kplo.estimate_normals(xyz, k=10, viewpoint=(0, 0, 0))
    nrm = -nrm                                                  # --flipNormals
    r, rn = float(np.float32(6 * mr)), float(np.float32(4 * mr))
    fa = forest_yaml.load_forest(forest)
    o_sc, o_kp = kplo.detect(xyz, nrm, 5, 6, r, rn, float(np.float32(0.5)), helpers.oracle_forest(fa))
    kp = np.loadtxt(out, skiprows=11, dtype=np.float32).reshape(-1, 4)
    assert info["points"] == len(xyz) and info["keypoints"] == len(o_kp) > 0
    assert np.array_equal(kp[:, :3], xyz[o_kp]) and np.array_equal(kp[:, 3], o_sc[o_kp])
    res = subprocess.run([EXE, "--pathCloud", str(pcd), "--printResolution"], capture_output=True, text=True, timeout=300)
    assert res.returncode == 0 and float(res.stdout.strip().splitlines()[-1]) == mr
    # no setNormals at all: the detector's own fallback (radius search with the feature radius, hpp:125-148)
    cmd = [c for c in cmd if c != "--flipNormals"] + ["--detectorNormals"]
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=300)
    assert res.returncode == 0, res.stderr
    assert "Computing normals for KPL" in res.stdout
    info = json.loads(res.stdout.strip().splitlines()[-1])
    nrm, _ = kplo.estimate_normals(xyz, k=0, radius=r, viewpoint=(0, 0, 0))
    o_sc, o_kp = kplo.detect(xyz, nrm, 5, 6, r, rn, float(np.float32(0.5)), helpers.oracle_forest(fa))
    kp = np.loadtxt(out, skiprows=11, dtype=np.float32).reshape(-1, 4)
    assert info["keypoints"] == len(o_kp) > 0
    assert np.array_equal(kp[:, :3], xyz[o_kp]) and np.array_equal(kp[:, 3], o_sc[o_kp])


def test_cli_subsampling_equals_the_tools_uniform_sampling(tmp_path):
    """--subSampling --leaf: the C++ UniformSampling of TestDetector (one point per leaf-sized voxel, the one
    closest to the voxel centre) keeps exactly the points tools/train_detector.uniform_sampling keeps; the rest
    of the run (normals on the subsampled cloud, radii from the resolution of the FULL cloud like
    /root/reference/src/main_test_detector.cpp:143-157) equals the oracle pipeline on those points."""
    from oracle import kplo
    from tests import helpers
    from tools import forest_yaml
    from tools.train_detector import uniform_sampling
    z = np.load(os.path.join(GOLD, "small_case.npz"))
    xyz = np.ascontiguousarray(z["xyz"][np.isfinite(z["xyz"]).all(axis=1)])
    pcd, out = tmp_path / "small.pcd", tmp_path / "kp.pcd"
    write_pcd(pcd, xyz, None, True)
    forest = os.path.join(GOLD, "small_forest.yaml.gz")
    cmd = [EXE, "--pathCloud", str(pcd), "--pathRF", forest, "--radiusFeatures", "6", "--pathKP=%s" % out,
           "--radiusNMS", "4", "--radiusInMr", "--subSampling", "--leaf", "1.5", "--annuli", "5", "--bins", "6",
           "-t", "0.5", "--flipNormals", "--json"]
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=300)
    assert res.returncode == 0, res.stderr
    info = json.loads(res.stdout.strip().splitlines()[-1])
    mr = kplo.cloud_resolution(xyz)                               # of the full cloud
    leaf = float(np.float32(1.5 * mr))
    keep = uniform_sampling(xyz, leaf)
    sub = np.ascontiguousarray(xyz[keep])
    assert 0 < len(sub) < len(xyz) and info["points"] == len(sub)
    nrm, _ = kplo.estimate_normals(sub, k=10, viewpoint=(0, 0, 0))
    nrm = -nrm                                                    # --flipNormals
    r, rn = float(np.float32(6 * mr)), float(np.float32(4 * mr))
    fa = forest_yaml.load_forest(forest)
    o_sc, o_kp = kplo.detect(sub, nrm, 5, 6, r, rn, float(np.float32(0.5)), helpers.oracle_forest(fa))
    kp = np.loadtxt(out, skiprows=11, dtype=np.float32).reshape(-1, 4)
    assert info["keypoints"] == len(o_kp) > 0
    assert np.array_equal(kp[:, :3], sub[o_kp]) and np.array_equal(kp[:, 3], o_sc[o_kp])


def test_protected_members_of_the_class(tmp_path):
    """runForest / computePointFeatures (protected in the reference, include/KeypointLearning.h:164-177) through a
    subclass: same scores and feature rows as the public path."""
    z = np.load(os.path.join(GOLD, "small_case.npz"))
    ok = np.isfinite(z["xyz"]).all(axis=1)
    pcd = tmp_path / "small.pcd"
    write_pcd(pcd, z["xyz"][ok], z["nrm"][ok], True)          # some normals are non-finite: those points are skipped
    cmd = [EXE, "--pathCloud", str(pcd), "--pathRF", os.path.join(GOLD, "small_forest.yaml.gz"), "--radiusFeatures", "6",
           "--radiusNMS", "4", "--radiusInMr", "--annuli", "5", "--bins", "6", "-t", "0.5", "--checkProtected"]
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=300)
    assert res.returncode == 0, res.stdout + res.stderr
    assert "checkProtected: 0 mismatches" in res.stdout


def test_detect_views_batches_like_single_runs(tmp_path):
    """DetectViews: plain C++ over the C-ABI, caller-owned device buffers (hipMalloc), normals estimated on the
    device into pcl::Normal records, three views in ONE kpl_compute_batch_device call -- same keypoints as the
    oracle pipeline per view."""
    from oracle import kplo
    from tests import helpers
    from tools import forest_yaml, synth
    exe = os.path.join(ROOT, "keypoint-learning_amd", "DetectViews")
    assert os.path.exists(exe)
    forest = os.path.join(ROOT, "data", "forests", "synth200k_a5b6_t10.yaml.gz")
    fa = forest_yaml.load_forest(forest)
    files, expect = [], []
    for k, (nx, ny) in enumerate([(60, 50), (45, 70), (80, 30)]):
        xyz, _ = synth.make_cloud(nx, ny, seed=60 + k)
        pcd = tmp_path / ("v%d.pcd" % k)
        write_pcd(pcd, xyz, None, k != 1)                     # one of them ascii
        if k == 1:
            xyz = np.loadtxt(pcd, skiprows=11, dtype=np.float32).reshape(-1, 3)   # what the ascii round trip keeps
        mr = kplo.cloud_resolution(xyz)
        nrm, _ = kplo.estimate_normals(xyz, k=10)
        r, rn = float(np.float32(6 * mr)), float(np.float32(4 * mr))
        _, kp = kplo.detect(xyz, -nrm, 5, 6, r, rn, float(np.float32(0.85)), helpers.oracle_forest(fa))
        files.append(str(pcd)); expect.append((len(xyz), kp))
    cmd = [exe, "--pathRF", forest, "--radiusFeatures", "6", "--radiusNMS", "4", "--radiusInMr", "--annuli", "5", "--bins", "6",
           "-t", "0.85", "--flipNormals", "--pathKP", str(tmp_path / "kp")] + files
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=300)
    assert res.returncode == 0, res.stdout + res.stderr
    rows = [json.loads(ln) for ln in res.stdout.strip().splitlines() if ln.startswith("{")]     # (RCCL prints a banner)
    assert len(rows) == 4 and rows[-1]["devices"] == 1 and rows[-1]["views"] == 3       # per view + the summary line
    for row, (n, kp) in zip(rows, expect):
        assert row["points"] == n and row["keypoints"] == len(kp) > 0 and row["index_checksum"] == int(kp.astype(np.int64).sum())
    assert os.path.exists(tmp_path / "kp2.pcd")


def test_cli_errors():
    res = subprocess.run([EXE, "--pathRF", "/nonexistent.yaml.gz"], capture_output=True, text=True)
    assert res.returncode != 0 and "impossible to load random forest" in res.stderr
    res = subprocess.run([EXE, "--subSampling"], capture_output=True, text=True)
    assert "Subsampling needs leaf." in res.stdout


def test_cli_organized_cloud_without_normals(tmp_path):
    """An ORGANIZED .pcd (WIDTH x HEIGHT, NaN holes) and no setNormals: the detector's fallback is
    pcl::IntegralImageNormalEstimation, SIMPLE_3D_GRADIENT, smoothing size 5 (hpp:138-145) -- here
    kpl_estimate_normals_organized.  The whole run equals the oracle pipeline on the same file."""
    from oracle import kplo
    from tests import helpers
    from tests.test_oracle_organized_normals import depth_image
    from tools import forest_yaml
    W, H = 96, 72
    xyz = depth_image(W, H, seed=5, step=50, holes=12, bumps=0.12, period=5.0)
    pcd, out = tmp_path / "org.pcd", tmp_path / "kp.pcd"
    write_pcd(pcd, xyz, None, True)
    hdr = open(pcd, "rb").read().replace(b"WIDTH %d\nHEIGHT 1\n" % (W * H), b"WIDTH %d\nHEIGHT %d\n" % (W, H))
    open(pcd, "wb").write(hdr)
    mr = kplo.cloud_resolution(xyz)
    r, rn = float(np.float32(6 * mr)), float(np.float32(4 * mr))
    nrm, _ = kplo.integral_image_normals(xyz, W, H, 5.0)
    assert 0 < np.isnan(nrm[:, 0]).sum() < len(xyz)
    # a forest whose thresholds come from this cloud's own features, so that the responses differ
    from tools import synth
    ok = np.flatnonzero(np.isfinite(nrm[:, 0]))[::7].astype(np.int32)
    feat = kplo.Grid(xyz, r).features(nrm, 5, 6, r, ok)
    forest = str(tmp_path / "forest.yaml.gz")
    forest_yaml.save_forest(synth.random_forest(30, ntrees=12, max_depth=9, seed=23, target_nodes_per_tree=120, feat=feat), forest)
    cmd = [EXE, "--pathCloud", str(pcd), "--pathRF", forest, "--radiusFeatures", "6", "--pathKP=%s" % out,
           "--radiusNMS", "4", "--radiusInMr", "--annuli", "5", "--bins", "6", "-t", "0.3", "--json", "--detectorNormals"]
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=300)
    assert res.returncode == 0, res.stderr + res.stdout
    assert "Computing normals for KPL" in res.stdout
    info = json.loads(res.stdout.strip().splitlines()[-1])
    fa = forest_yaml.load_forest(forest)
    o_sc, o_kp = kplo.detect(xyz, nrm, 5, 6, r, rn, float(np.float32(0.3)), helpers.oracle_forest(fa))
    kp = np.loadtxt(out, skiprows=11, dtype=np.float32).reshape(-1, 4)
    assert info["points"] == len(xyz) and info["keypoints"] == len(o_kp) > 0
    assert np.array_equal(kp[:, :3], xyz[o_kp]) and np.array_equal(kp[:, 3], o_sc[o_kp])


def test_cli_sorted_search_method(tmp_path):
    """--sortedSearch: the C++ class is handed pcl::search::KdTree(true) through the inherited setSearchMethod and
    scores with the neighbors in FLANN's sorted order: same keypoints and responses as the oracle's sorted mode
    (and not those of the canonical order)."""
    from oracle import kplo
    from tests import helpers
    from tools import forest_yaml
    z = np.load(os.path.join(GOLD, "cheff000.npz"))
    n = 30000
    xyz, nrm = np.ascontiguousarray(z["xyz"][:n]), np.ascontiguousarray(z["nrm"][:n])
    pcd, out = tmp_path / "view.pcd", tmp_path / "kp.pcd"
    write_pcd(pcd, xyz, nrm, True)
    forest = os.path.join(ROOT, "data", "forests", "synth200k_a5b6_t10.yaml.gz")
    r, rn, thr = float(z["r_feat"]), float(z["r_nms"]), float(np.float32(0.85))
    cmd = [EXE, "--pathCloud", str(pcd), "--pathRF", forest, "--pathKP=%s" % out, "--radiusFeatures", "%.9g" % r,
           "--radiusNMS", "%.9g" % rn, "-t", "0.85", "--annuli", "5", "--bins", "6", "--json", "--sortedSearch"]
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=300)
    assert res.returncode == 0, res.stderr
    of = helpers.oracle_forest(forest_yaml.load_forest(forest))
    r32, rn32 = float(np.float32(r)), float(np.float32(rn))         # the CLI parses its radii as float
    s_sc, s_kp = kplo.detect(xyz, nrm, 5, 6, r32, rn32, thr, of, order=kplo.ORDER_SORTED, threads=helpers.usable_cores())
    c_sc, c_kp = kplo.detect(xyz, nrm, 5, 6, r32, rn32, thr, of, threads=helpers.usable_cores())
    kp = np.loadtxt(out, skiprows=11, dtype=np.float32).reshape(-1, 4)
    assert len(kp) == len(s_kp) > 0
    assert np.array_equal(kp[:, :3], xyz[s_kp]) and np.array_equal(kp[:, 3], s_sc[s_kp])
    assert not np.array_equal(s_sc, c_sc)


def test_cli_normals_flip_towards_the_pcd_viewpoint(tmp_path):
    """A PCD whose VIEWPOINT is not the origin: pcl::PCDReader puts it into sensor_origin_, and both PCL normal
    estimators flip towards it by default.  TestDetector's own k = 10 normals and the detector's fallback
    (--detectorNormals) must do the same: the run equals the oracle pipeline with that viewpoint, and differs from
    the one with the viewpoint at the origin."""
    from oracle import kplo
    from tests import helpers
    from tools import forest_yaml
    z = np.load(os.path.join(GOLD, "small_case.npz"))
    xyz = np.ascontiguousarray(z["xyz"][np.isfinite(z["xyz"]).all(axis=1)])
    vp = (float(xyz[:, 0].mean()), float(xyz[:, 1].mean()), float(xyz[:, 2].max() + 50.0))    # above the surface (the origin is below most of it)
    pcd, out = tmp_path / "small.pcd", tmp_path / "kp.pcd"
    write_pcd(pcd, xyz, None, True)
    raw = open(pcd, "rb").read().replace(b"VIEWPOINT 0 0 0 1 0 0 0", b"VIEWPOINT %.9g %.9g %.9g 1 0 0 0" % vp)
    open(pcd, "wb").write(raw)
    vp32 = tuple(float(np.float32(float("%.9g" % v))) for v in vp)
    forest = os.path.join(GOLD, "small_forest.yaml.gz")
    fa = forest_yaml.load_forest(forest)
    mr = kplo.cloud_resolution(xyz)
    r, rn = float(np.float32(6 * mr)), float(np.float32(4 * mr))
    base = [EXE, "--pathCloud", str(pcd), "--pathRF", forest, "--radiusFeatures", "6", "--pathKP=%s" % out,
            "--radiusNMS", "4", "--radiusInMr", "--annuli", "5", "--bins", "6", "-t", "0.5", "--json"]
    for extra, kw in (([], dict(k=10)), (["--detectorNormals"], dict(k=0, radius=r))):
        res = subprocess.run(base + extra, capture_output=True, text=True, timeout=300)
        assert res.returncode == 0, res.stderr
        nrm, _ = kplo.estimate_normals(xyz, viewpoint=vp32, **kw)
        nrm0, _ = kplo.estimate_normals(xyz, viewpoint=(0, 0, 0), **kw)
        assert np.mean(np.sign(nrm[:, 2]) != np.sign(nrm0[:, 2])) > 0.2          # the viewpoint does flip many of them
        o_sc, o_kp = kplo.detect(xyz, nrm, 5, 6, r, rn, float(np.float32(0.5)), helpers.oracle_forest(fa))
        kp = np.loadtxt(out, skiprows=11, dtype=np.float32).reshape(-1, 4)
        assert len(kp) == len(o_kp) > 0
        assert np.array_equal(kp[:, :3], xyz[o_kp]) and np.array_equal(kp[:, 3], o_sc[o_kp])


def test_detect_views_all_devices_equals_test_detector(tmp_path):
    """DetectViews --devices all: the C++ multi-GPU runner (one host thread per device, views dealt round robin,
    results gathered with ONE ncclAllGather of the packed [count][indices][responses] buffers, written out from
    device 0's copy).  On this box "all" is one device, RCCL still runs.  Every keypoint file must equal, byte for
    byte, the one TestDetector writes for the same view -- in the canonical and in the sorted neighbor order."""
    from tools import synth
    exe = os.path.join(ROOT, "keypoint-learning_amd", "DetectViews")
    forest = os.path.join(ROOT, "data", "forests", "synth200k_a5b6_t10.yaml.gz")
    files = []
    for k, (nx, ny) in enumerate([(60, 50), (45, 70), (80, 30), (64, 64), (33, 90), (70, 41), (52, 52), (48, 60), (90, 25), (40, 40), (75, 44)]):
        xyz, nrm = synth.make_cloud(nx, ny, seed=160 + k)
        pcd = tmp_path / ("v%02d.pcd" % k)
        write_pcd(pcd, xyz, nrm if k % 3 else None, True)            # every third file without normals: estimated (k = 10)
        files.append(str(pcd))
    common = ["--pathRF", forest, "--radiusFeatures", "6", "--radiusNMS", "4", "--radiusInMr", "--annuli", "5", "--bins", "6", "-t", "0.85"]
    for extra in ([], ["--sortedSearch"]):
        res = subprocess.run([exe] + common + extra + ["--devices", "all", "--rounds", "3", "--pathKP", str(tmp_path / "multi")] + files,
                             capture_output=True, text=True, timeout=600)
        assert res.returncode == 0, res.stdout + res.stderr
        rows = [json.loads(ln) for ln in res.stdout.strip().splitlines() if ln.startswith("{")]
        summary = rows[-1]
        assert len(rows) == len(files) + 1 and summary["views"] == len(files) and summary["devices"] >= 1
        assert "ncclAllGather" in summary["exchange"] and summary["Mpoints_per_s"] > 0
        for k, f in enumerate(files):
            out = tmp_path / "single.pcd"
            one = subprocess.run([EXE, "--pathCloud", f, "--pathKP=%s" % out, "--json"] + common + extra, capture_output=True, text=True, timeout=300)
            assert one.returncode == 0, one.stderr
            info = json.loads(one.stdout.strip().splitlines()[-1])
            assert rows[k]["keypoints"] == info["keypoints"] > 0 and rows[k]["points"] == info["points"]
            assert open(out).read() == open(str(tmp_path / "multi") + "%d.pcd" % k).read(), (k, extra)


def test_cli_host_staging_gives_the_same_file(tmp_path):
    """--hostStaging: setHostStaging(true) before the setters; compute() goes through the pinned staging path."""
    z = np.load(os.path.join(GOLD, "cheff000.npz"))
    pcd = tmp_path / "view.pcd"
    write_pcd(pcd, z["xyz"], z["nrm"], True)
    outs = []
    for extra in ([], ["--hostStaging"]):
        out = tmp_path / ("kp%d.pcd" % len(outs))
        cmd = [EXE, "--pathCloud", str(pcd), "--pathRF", os.path.join(ROOT, "data", "forests", "synth200k_a5b6_t10.yaml.gz"),
               "--pathKP=%s" % out, "--radiusFeatures", "%.9g" % float(z["r_feat"]), "--radiusNMS", "%.9g" % float(z["r_nms"]),
               "-t", "0.85", "--annuli", "5", "--bins", "6", "--json"] + extra
        res = subprocess.run(cmd, capture_output=True, text=True, timeout=300)
        assert res.returncode == 0, res.stderr
        outs.append(open(out).read())
    assert outs[0] == outs[1] and outs[0].count("\n") > 100
