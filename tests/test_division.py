"""The HIP kernels divide by the per-launch constants `annulus dimension` and `bin dimension`
with a 3-instruction sequence (reciprocal multiply + two FMAs) instead of hipcc's IEEE division
expansion.  It must be bit-identical to true division for every numerator the path can produce.
Checked here exhaustively on the CPU (same arithmetic, C fmaf)."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def divlib(tmp_path_factory):
    out = tmp_path_factory.mktemp("div") / "libdivcheck.so"
    subprocess.check_call(["gcc", "-O2", "-ffp-contract=off", "-mfma", "-shared", "-fPIC", "-o", str(out),
                           os.path.join(HERE, "csrc", "div_check.c"), "-lm"])
    lib = C.CDLL(str(out))
    lib.check_range.argtypes = [C.c_float, C.c_uint32, C.c_uint32, C.c_uint32, C.POINTER(C.c_uint32)]
    lib.check_range.restype = C.c_long
    return lib


def bits(x):
    return int(np.float32(x).view(np.uint32))


DIMS = [np.float32(5.038615) / np.float32(5), np.float32(2) / np.float32(6), np.float32(2) / np.float32(10),
        np.float32(3.9) / np.float32(5), np.float32(20.0) / np.float32(8), np.float32(2) / np.float32(7),
        np.float32(0.01234) / np.float32(3), np.float32(2) / np.float32(1)]


@pytest.mark.parametrize("b", DIMS)
def test_every_positive_numerator_up_to_16_dims(divlib, b):
    """distance / dim and cosine / dim: every float in [2^-100, 16*dim]."""
    bad = C.c_uint32(0)
    n = divlib.check_range(float(b), bits(2.0 ** -100), bits(np.float32(16) * b), 1, C.byref(bad))
    assert n == 0, "first mismatch at bits 0x%08x" % bad.value


@pytest.mark.parametrize("b", DIMS[:4])
def test_every_negative_numerator_down_to_minus_dim(divlib, b):
    """(value - center) / dim is in [-0.5, 0.5] * dim: every negative float down to -dim."""
    bad = C.c_uint32(0)
    n = divlib.check_range(float(b), bits(-(2.0 ** -100)), bits(-b), 1, C.byref(bad))
    assert n == 0, "first mismatch at bits 0x%08x" % bad.value


def test_tiny_numerators_floor_to_zero(divlib):
    """Below 2^-100 the quotient may differ in its last (denormal) bits; the path only floors it."""
    for b in DIMS[:3]:
        rb = np.float32(1) / b
        for a in (np.float32(0), np.float32(1e-45), np.float32(1e-40), np.float32(2.0 ** -101)):
            q = np.float32(a * rb)
            assert np.floor(q) == 0 and np.floor(np.float32(a / b)) == 0
