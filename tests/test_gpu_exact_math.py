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


def test_div_rn_equals_division_for_every_numerator_of_many_divisors():
    """div_rn(a, b, RN(1/b)) == a / b ON THE DEVICE (v_fma_f32, v_mul_f32 against hipcc's IEEE division) for 107 divisors --
    2 / B for B = 1 .. 32, 48 random support / A, 27 mantissa edge cases -- and EVERY numerator the soft assignment can
    form: [2^-100, 16 b] and [-b, -2^-100] (tests/test_division.py does 8 divisors on the CPU with C fmaf)."""
    exe = os.path.join(ROOT, "tools", "check_exact_math")
    res = subprocess.run([exe, "div"], capture_output=True, text=True, timeout=900)
    assert res.returncode == 0, res.stdout + res.stderr
    j = json.loads([ln for ln in res.stdout.splitlines() if ln.startswith("{")][-1])
    assert j["mode"] == "div" and j["divisors"] >= 64 and j["checked"] > 5e10 and j["mismatches"] == 0
