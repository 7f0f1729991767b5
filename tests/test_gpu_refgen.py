"""The reference-regeneration kit end to end, with this repo's engine in the reference's place: the SAME driver source that a
PCL + OpenCV machine compiles against the reference's header (tools/refgen/CMakeLists.txt) is compiled against
include/KeypointLearning.h (-DREFGEN_WITH_KPL, keypoint-learning_amd/build.py), run on the exported inputs on the GPU, and
tools/refgen/compare.py must find every array of every run identical to tests/golden/*.npz -- the sorted-search runs, the
canonical ones (the engine IS what made those fixtures' expectations), both normal estimators and the organized fallback."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools", "refgen"))
import compare  # noqa: E402
import export_inputs  # noqa: E402

EXE = os.path.join(ROOT, "tools", "refgen", "refgen_driver_kpl")


@pytest.mark.gpu
def test_driver_over_libkpl_reproduces_every_fixture(tmp_path):
    assert os.path.exists(EXE), "tools/refgen/refgen_driver_kpl is not built (keypoint-learning_amd/build.py)"
    d = str(tmp_path)
    rows = export_inputs.export(d)
    res = subprocess.run([EXE, d], capture_output=True, text=True, timeout=900)
    assert res.returncode == 0, (res.stdout[-2000:], res.stderr[-2000:])
    rep = compare.judge(d, explain=True)
    assert "libkpl" in rep["engine"]
    assert len(rep["runs"]) == len(rows)
    bad = {rid: [a for a in r["arrays"] if not a["identical"]] for rid, r in rep["runs"].items()
           if not all(a["identical"] for a in r["arrays"])}
    assert not bad, bad
    assert rep["sorted_runs_pinned"].split()[0] == rep["sorted_runs_pinned"].split()[2]
    assert set(rep["forest_files"].values()) == {"yes"} and len(rep["forest_files"]) == 3
