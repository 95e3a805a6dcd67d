"""A short run of tests/fuzz_parity.py inside the suite: random small instances of both gadgets (random counts, weights incl.
0 / 2^128 - 1, points incl. R = P, R = -P, y = 0, bytes >= q, rz patterns, seed pairs), device-built whole SNARKs, commitments and
is_sat byte-compared with the oracle on the Python model's instances.  The long runs are in profiles/r04_fuzz.txt."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_random_instances_match_the_oracle():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "fuzz_parity.py"), "40", "31337"], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    assert "40 random instances (seed 31337), 0 mismatches" in out.stdout
