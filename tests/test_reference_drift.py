"""The restatement in oracle/kpl_oracle.c against the reference text it cites (tools/check_reference_drift.py).
Runs where /root/reference exists (the build container); skipped on the GPU box."""
import importlib.util
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(not os.path.isdir("/root/reference"), reason="/root/reference is not on this host")
def test_oracle_still_matches_the_cited_reference_lines(capsys):
    spec = importlib.util.spec_from_file_location("drift", os.path.join(ROOT, "tools", "check_reference_drift.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    rc = mod.main()
    assert rc == 0, capsys.readouterr().out
