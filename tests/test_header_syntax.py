"""include/KeypointLearning.h must at least PARSE and type-check in all its configurations: with the in-repo shim
(what TestDetector builds with) and with -DKPL_USE_PCL against declaration-only PCL headers
(tests/csrc/pcl_decl: names and signatures of PCL 1.8's public API, no bodies).  A syntax check, not parity."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TU = os.path.join(ROOT, "tests", "csrc", "use_pcl_syntax.cpp")


def _syntax(flags):
    cmd = ["g++", "-std=c++14", "-fsyntax-only", "-Wall", "-Wextra", "-Werror", "-I", os.path.join(ROOT, "include")] + flags + [TU]
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=300)
    assert res.returncode == 0, res.stderr[-4000:]


def test_header_with_the_in_repo_shim():
    _syntax([])


def test_header_with_kpl_use_pcl_against_declarations():
    _syntax(["-DKPL_USE_PCL", "-I", os.path.join(ROOT, "tests", "csrc", "pcl_decl")])
