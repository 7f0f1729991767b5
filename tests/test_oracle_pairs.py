"""Soft assignment (findAnnulusPair / findBinPair, /root/reference/src/KeypointLearning.cpp:41-92):
the oracle against (a) the known answers of SURVEY.md 8(a), (b) the committed table generated from
the reference's own functions, (c) the reference's functions themselves when oracle/_ref exists."""
import json
import os

import numpy as np
import pytest

GOLD = os.path.join(os.path.dirname(__file__), "golden", "pair_kat.json")

# values printed by the compiled reference translation unit
SURVEY_ANNULUS = [  # n=5, support=3.9f
    (0.0, 0, 0, 0.5), (0.3, 0, 0, 0.115384616), (0.4, 0, 1, 0.0128204999), (0.79, 1, 0, 0.487179548),
    (1.99, 2, 3, 0.0512819998), (3.7, 4, 4, 0.243589506), (3.899, 4, 4, 0.498717546)]
SURVEY_BIN = [      # n=6
    (-0.1, 0, 0, 0.5), (0.0, 0, 0, 0.5), (0.05, 0, 0, 0.350000024), (0.3, 0, 1, 0.400000006),
    (1.0, 3, 2, 0.499999881), (1.9, 5, 5, 0.199999809), (2.0, 5, 5, 0.499999881), (2.5, 5, 5, 0.499999881)]


def test_survey_known_answers(oracle):
    for d, i, p, w in SURVEY_ANNULUS:
        gi, gp, gw = oracle.find_annulus_pair(5, float(np.float32(d)), float(np.float32(3.9)))
        assert (gi, gp) == (i, p) and gw == np.float32(w), (d, gi, gp, gw)
    for c, i, p, w in SURVEY_BIN:
        gi, gp, gw = oracle.find_bin_pair(6, float(np.float32(c)))
        assert (gi, gp) == (i, p) and gw == np.float32(w), (c, gi, gp, gw)


def test_abs_is_float_abs(oracle):
    # the `abs` trap of SURVEY.md: an int abs() would give w == 0 here
    assert oracle.find_annulus_pair(5, 0.3, float(np.float32(3.9)))[2] == np.float32(0.115384616)


def test_committed_reference_table(oracle):
    kat = json.load(open(GOLD))
    assert len(kat["annulus"]) >= 200 and len(kat["bin"]) >= 200
    for n, d, s, i, p, w in kat["annulus"]:
        gi, gp, gw = oracle.find_annulus_pair(n, d, s)
        assert (gi, gp) == (i, p) and gw == np.float32(w), (n, d, s)
    for n, c, i, p, w in kat["bin"]:
        gi, gp, gw = oracle.find_bin_pair(n, c)
        assert (gi, gp) == (i, p) and gw == np.float32(w), (n, c)


def test_against_reference_functions_when_built(oracle):
    if not oracle.RefPairs.available():
        pytest.skip("oracle/_ref not built (reference checkout absent)")
    ref = oracle.RefPairs()
    rng = np.random.RandomState(3)
    for n, support in [(5, 3.9), (8, 5.04), (3, 20.0)]:
        s = float(np.float32(support))
        for d in rng.uniform(0, s, 4000).astype(np.float32):
            if d >= s:
                continue
            assert oracle.find_annulus_pair(n, float(d), s) == ref.annulus(n, float(d), s)
    for n in (6, 10, 1):
        for c in rng.uniform(-0.3, 2.3, 4000).astype(np.float32):
            assert oracle.find_bin_pair(n, float(c)) == ref.bin(n, float(c))
