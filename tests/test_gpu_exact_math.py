"""csrc/exact_math.h: the 9-instruction correctly rounded square root of the feature kernels equals hipcc's sqrtf
for EVERY non-negative finite float (2.1e9 values, checked on the device by tools/check_exact_math)."""
import json
import os
import subprocess

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_sqrt_rn_equals_sqrtf_for_every_float():
    exe = os.path.join(ROOT, "tools", "check_exact_math")
    assert os.path.exists(exe), "tools/check_exact_math is not built (python keypoint-learning_amd/build.py)"
    res = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stdout + res.stderr
    j = json.loads([ln for ln in res.stdout.splitlines() if ln.startswith("{")][-1])
    assert j["checked"] == 0x7f800000 and j["mismatches"] == 0
