"""The forest reader of libkpl (csrc/forest.cpp: what replaces cv::ml::RTrees::load,
/root/reference/include/impl/KeypointLearning.hpp:162) under AddressSanitizer + UBSan on the CPU:
truncated, corrupted, cut and duplicated YAML must be accepted or rejected, never crash."""
import os
import shutil
import subprocess

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


@pytest.mark.skipif(shutil.which("g++") is None, reason="needs g++")
def test_mutated_forests_never_crash_the_reader(tmp_path):
    exe = tmp_path / "fuzz_forest"
    src = os.path.join(ROOT, "keypoint-learning_amd", "csrc")
    subprocess.check_call(["g++", "-O1", "-g", "-std=c++17", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined",
                           "-I", src, os.path.join(HERE, "csrc", "fuzz_forest.cpp"), os.path.join(src, "forest.cpp"), "-lz",
                           "-o", str(exe)])
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=0")
    out = subprocess.run([str(exe), os.path.join(HERE, "golden", "small_forest.yaml.gz"), "1500"], capture_output=True,
                         text=True, timeout=600, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    accepted, rejected = int(out.stdout.split()[1]), int(out.stdout.split()[3])
    assert accepted + rejected == 1500 and rejected > accepted > 0


@pytest.mark.skipif(shutil.which("g++") is None, reason="needs g++")
def test_flat_layouts_walk_like_the_model(tmp_path):
    """flatten_forest (csrc/forest.cpp) for random forests of both device layouts -- 8-slot blocks below the top part,
    and the chained level-major layout of forests with >= 40 trees and >= 32 variables -- walked on the host with the
    kernels' child rules: the model's leaf and depth for every tree, the chain links t -> t + 16 -> ... -> resting
    leaf, under ASan / UBSan."""
    exe = tmp_path / "check_flat_forest"
    src = os.path.join(ROOT, "keypoint-learning_amd", "csrc")
    subprocess.check_call(["g++", "-O1", "-g", "-std=c++17", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined",
                           "-I", src, os.path.join(HERE, "csrc", "check_flat_forest.cpp"), os.path.join(src, "forest.cpp"), "-lz",
                           "-o", str(exe)])
    out = subprocess.run([str(exe), "25"], capture_output=True, text=True, timeout=600,
                         env=dict(os.environ, ASAN_OPTIONS="detect_leaks=0"))
    assert out.returncode == 0, out.stdout + out.stderr[-2000:]
    chained, blocked = int(out.stdout.split()[1]), int(out.stdout.split()[3])
    assert chained >= 5 and blocked >= 5
