"""The C-ABI library: loads, exports every symbol include/kpl.h declares, and fails loudly (no CPU
fallback) when no HIP device is usable.  No compute calls here."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "kpl.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(kpl_[a-z_0-9]+)\s*\(", text)))


def test_header_symbols_are_exported_and_bound(kpl):
    lib = kpl.load_library()
    names = declared_symbols()
    assert len(names) >= 25
    for n in names:
        assert hasattr(lib, n), "libkpl.so does not export " + n
    assert set(names) == set(kpl.SYMBOLS), "python binding and header disagree"
    # test hooks (include/kpl_debug.h) live in a library of their own (tests/csrc/libkpl_testhooks.so), not in the shipped one
    assert not any(n.startswith("kpl_debug") for n in names)
    assert not hasattr(lib, "kpl_debug_set_scan_poll_limit")


def test_version_and_status_strings(kpl):
    lib = kpl.load_library()
    assert lib.kpl_version() == 150
    # the binary that was loaded is the one built from THESE sources (build.py compiles the hash in)
    import importlib.util
    spec = importlib.util.spec_from_file_location("kpl_build", os.path.join(ROOT, "keypoint-learning_amd", "build.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    assert lib.kpl_source_hash().decode() == mod.source_hash()
    assert lib.kpl_status_string(0) == b"ok"
    assert b"forest" in lib.kpl_status_string(kpl.ERR_NO_FOREST)


def test_default_params_match_the_reference_ctor(kpl):
    # /root/reference/include/KeypointLearning.h:81
    p = kpl.Params()
    kpl.load_library().kpl_default_params(C.byref(p))
    assert (p.n_annulus, p.n_bins, p.non_maxima, p.non_maxima_draws_remove) == (5, 10, 1, 1)
    assert p.prediction_th == 0.5 and p.non_max_radius == 0.0 and p.radius_search == 0.0
    assert p.neighbor_order == kpl.NEIGHBORS_CANONICAL


def test_null_handles_do_not_crash(kpl):
    lib = kpl.load_library()
    assert lib.kpl_create(None, 0) == kpl.ERR_INVALID_ARG
    lib.kpl_destroy(None)
    assert lib.kpl_set_params(None, None) == kpl.ERR_INVALID_ARG
    assert lib.kpl_load_forest_file(None, b"x") == kpl.ERR_INVALID_ARG
    assert lib.kpl_detect(None, None, 12, None, 12, 0, None, None, 0, None) == kpl.ERR_INVALID_ARG
    assert lib.kpl_last_error(None) == b"null handle"


def test_no_gpu_means_a_loud_failure_not_a_fallback(kpl):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    h = C.c_void_p()
    assert kpl.load_library().kpl_create(C.byref(h), 0) == kpl.ERR_DEVICE
    with pytest.raises(kpl.KplError):
        kpl.KeypointLearningDetector()


def test_product_never_touches_the_oracle():
    """oracle/ is test infrastructure: nothing under the product package may name it."""
    pkg = os.path.join(ROOT, "keypoint-learning_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".h")):
                text = open(os.path.join(dirpath, f), errors="replace").read()
                assert "kplo" not in text and "kpl_oracle" not in text and "oracle/" not in text.replace(
                    "the oracle's", ""), f
    lib = C.CDLL(os.path.join(pkg, "libkpl.so"))
    assert not hasattr(lib, "kplo_detect")
