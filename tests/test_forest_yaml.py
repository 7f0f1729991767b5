"""OpenCV RTrees YAML(.gz): libkpl's C++ reader (host-only entry points, no GPU needed) against the
independent Python reader/writer in tools/forest_yaml.py, on generated files and on hand-written
snippets exercising the cv::FileStorage dialect."""
import gzip

import numpy as np
import pytest

from tools import forest_yaml

FIELDS = "root var thr left right value".split()


def both(kpl, data):
    a = kpl.forest_export_arrays(data)
    b = forest_yaml.load_forest(data)
    for k in FIELDS:
        assert np.array_equal(a[k], getattr(b, k)), k
    assert a["var_count"] == b.var_count
    return a


def test_roundtrip_generated_forest(kpl, cases):
    fa = cases.trained_forest()
    text = forest_yaml.forest_to_yaml(fa).encode()
    a = both(kpl, text)
    for k in FIELDS:
        assert np.array_equal(a[k], getattr(fa, k)), k
    info = kpl.forest_inspect(gzip.compress(text))          # gzip sniffed by magic
    assert info["ntrees"] == fa.ntrees and info["nnodes"] == fa.nnodes and info["var_count"] == 30
    assert info["max_depth"] == int(fa.depth.max()) + 1
    both(kpl, forest_yaml.forest_to_yaml(fa, legacy_keys=True, top_key="my_random_trees").encode())


HAND = b"""%YAML:1.0
---
opencv_ml_rtrees:
   format: 3
   is_classifier: 1
   var_all: 4
   var_count: 3
   training_params:
      use_surrogates: 0
      priors: !!opencv-matrix
         rows: 1
         cols: 2
         dt: d
         data: [ 1., 1. ]
   var_type: [ 0, 0,
       0, 1 ]
   class_labels: [ 0, 1 ]
   ntrees: 2
   trees:
      -
         nodes:
            -
               depth: 0
               value: 1.
               norm_class_idx: 1
               splits:
                  - { var:2, quality:1.5e+01, le:.5 }
            -
               depth: 1
               value: 0.
               norm_class_idx: 0
            -
               depth: 1
               value: 1.
               norm_class_idx: 1
               splits:
                  - { var:0, quality:3., le:2.50000000e-01 }
            -
               depth: 2
               value: 1.
            -
               depth: 2
               value: 0.
      -
         nodes:
            - { depth:0, value:0., norm_class_idx:0, splits:[ { var:1, quality:1., gt:-1.25e-01 } ] }
            - { depth:1, value:1. }
            - { depth:1, value:0. }
"""


def test_hand_written_dialect(kpl):
    a = both(kpl, HAND)
    assert a["root"].tolist() == [0, 5]
    assert a["var"].tolist() == [2, -1, 0, -1, -1, 1, -1, -1]
    assert a["thr"][[0, 2, 5]].tolist() == [0.5, 0.25, -0.125]
    assert a["left"][[0, 2]].tolist() == [1, 3] and a["right"][[0, 2]].tolist() == [2, 4]
    # 'gt' = inversed split: the children swap roles
    assert (a["left"][5], a["right"][5]) == (7, 6)
    assert a["value"].tolist() == [1, 0, 1, 1, 0, 0, 1, 0]
    assert kpl.forest_inspect(HAND)["max_depth"] == 3


@pytest.mark.parametrize("breakage,frag", [
    (lambda s: s.replace(b"le:.5", b"in:[ 1, 2 ]"), "categorical"),
    (lambda s: s.replace(b"   trees:", b"   shrubs:"), "trees"),
    (lambda s: s[: s.index(b"            -\n               depth: 2")], "unfilled"),
    (lambda s: s.replace(b"ntrees: 2", b"ntrees: 3"), "ntrees"),
    (lambda s: s.replace(b"value: 0.\n               norm_class_idx: 0\n", b"value: 0.25000001\n", 1), ""),
    (lambda s: b"\x1f\x8bgarbage", "gzip"),
    (lambda s: b"", ""),
])
def test_malformed_files_are_rejected_with_a_message(kpl, breakage, frag):
    with pytest.raises(kpl.KplError) as e:
        kpl.forest_inspect(breakage(HAND)) if breakage(HAND) else kpl.forest_inspect(b"")
    assert e.value.status in (kpl.ERR_FOREST_PARSE, kpl.ERR_INVALID_ARG)
    assert frag in str(e.value)


def test_leaf_value_must_be_float_exact(kpl):
    ok = HAND.replace(b"value: 0.\n               norm_class_idx: 0\n", b"value: 2.5\n", 1)
    assert kpl.forest_inspect(ok)["nnodes"] == 8


def test_var_count_limits(kpl):
    with pytest.raises(kpl.KplError):
        kpl.forest_inspect(HAND.replace(b"var_count: 3", b"var_count: 2"))     # var 2 >= var_count
    with pytest.raises(kpl.KplError):
        kpl.forest_inspect(HAND.replace(b"var_count: 3", b"var_count: 300"))   # > 255 unsupported


def test_committed_fixture_files_parse(kpl):
    import os
    root = os.path.dirname(os.path.dirname(__file__))
    for rel in ("data/forests/synth200k_a5b6_t10.yaml.gz", "tests/golden/small_forest.yaml.gz"):
        data = open(os.path.join(root, rel), "rb").read()
        both(kpl, data)
