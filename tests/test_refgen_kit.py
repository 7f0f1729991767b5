"""tools/refgen -- the kit with which a machine that has PCL 1.8 + OpenCV 3.2 regenerates the expected outputs of
tests/golden/*.npz from the reference itself (tests/golden/README.md).  Here, without PCL: the exporter writes what it
says, the driver parses in both of its builds, and the comparison finds what it must find."""
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools", "refgen"))
import compare  # noqa: E402
import export_inputs  # noqa: E402
from tools import cloud_io  # noqa: E402

DRIVER = os.path.join(ROOT, "tools", "refgen", "refgen_driver.cpp")


def test_exporter_writes_bit_exact_inputs_and_a_manifest(tmp_path):
    rows = export_inputs.export(str(tmp_path))
    man = compare.manifest_rows(str(tmp_path))
    assert len(man) == len(rows) >= 20
    expect = export_inputs.expectations()
    assert set(man) == set(expect)                                   # every run has expected arrays, and the other way round
    for rid, row in man.items():
        for key in ("cloud", "normals", "forest", "query"):
            if key in row:
                assert os.path.exists(os.path.join(str(tmp_path), row[key])), (rid, key)
        npz, arrays, mode = expect[rid]
        z = np.load(os.path.join(ROOT, "tests", "golden", npz))
        assert all(name in z.files for name in arrays.values()), rid
    # the floats travel as their bits (DATA binary), radii as hex doubles
    z = np.load(os.path.join(ROOT, "tests", "golden", "cheff001.npz"))
    back = cloud_io.read_pcd_xyz(os.path.join(str(tmp_path), "clouds", "cheff001.pcd"))
    assert np.array_equal(back.view(np.uint32), z["xyz"].view(np.uint32))
    assert float.fromhex(man["cheff001_sorted_detect"]["r_feat"]) == float(z["r_feat"]) == 20.0
    assert float.fromhex(man["cheff001_sorted_detect"]["thr"]) == float(np.float32(0.85))
    small = np.load(os.path.join(ROOT, "tests", "golden", "small_case.npz"))
    assert float.fromhex(man["small_sorted_features_5x6"]["r_feat"]) == float(small["r_feat"])
    # the sorted-search runs are the ones held to "every bit"
    assert sum(1 for v in expect.values() if v[2] == "bitwise") >= 5


def _syntax(flags):
    res = subprocess.run(["g++", "-std=c++14", "-fsyntax-only", "-Wall", "-Wextra", "-Werror", "-I", os.path.join(ROOT, "include")] + flags + [DRIVER],
                         capture_output=True, text=True, timeout=300)
    assert res.returncode == 0, res.stderr[-4000:]


def test_driver_parses_against_this_repos_drop_in_header():
    _syntax(["-DREFGEN_WITH_KPL"])


def test_driver_parses_in_its_reference_build_against_declarations():
    """the reference build's code path (pcl::io, pcl::NormalEstimation, pcl::IntegralImageNormalEstimation, cv::Mat) against
    declaration-only PCL / OpenCV headers, with this repo's header -- the same class API -- standing where the reference's
    stands; also the first time -DKPL_USE_OPENCV of include/KeypointLearning.h meets a compiler"""
    _syntax(["-DKPL_USE_PCL", "-DKPL_USE_OPENCV", "-I", os.path.join(ROOT, "tests", "csrc", "pcl_decl"),
             "-I", os.path.join(ROOT, "tests", "csrc", "opencv_decl")])


def test_compare_self_test():
    assert compare.self_test() == 0


def test_numpy_restatement_agrees_with_the_sorted_cheff_fixture_on_sample_rows():
    """compare.py's explainer is a second restatement of hpp:321-376 (numpy float32, brute-force sorted neighbors); through the
    10-tree fixture forest its rows must give the committed sorted-mode scores of cheff000"""
    from tools import forest_yaml
    c = np.load(os.path.join(ROOT, "tests", "golden", "cheff000.npz"))
    s = np.load(os.path.join(ROOT, "tests", "golden", "sorted_case.npz"))
    fa = forest_yaml.load_forest(os.path.join(ROOT, "data", "forests", "synth200k_a5b6_t10.yaml.gz"))
    for i in (0, 12345, 40000, 62233):
        row = compare.feature_row(c["xyz"], c["nrm"], i, 5, 6, float(c["r_feat"]))
        total = 0.0
        for t in range(len(fa.root)):
            node = int(fa.root[t])
            while fa.var[node] >= 0:
                node = int(fa.left[node]) if row[fa.var[node]] <= fa.thr[node] else int(fa.right[node])
            total += float(fa.value[node])
        score = np.float32(1) - np.float32(total) / (np.float32(len(fa.root)) * np.float32(1.0))
        assert np.float32(score) == s["cheff_scores"][i], i
