"""The scripts under examples/ run and recover what they simulate (GPU)."""
import os
import re
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(name, *args):
    r = subprocess.run([sys.executable, os.path.join(REPO, "examples", name)] + list(args), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    return r.stdout


def test_lds_example():
    out = _run("lds_example.py", "200", "40")
    m = re.search(r"rms of y - <C><x> : ([0-9.e+-]+)\s+rms of y : ([0-9.e+-]+)", out)
    assert m and float(m.group(1)) < 0.2 * float(m.group(2)), out


def test_pca_example():
    out = _run("pca_missing_data.py", "300", "30")
    m = re.search(r"rms error of the imputed entries: ([0-9.e+-]+)\s+\(spread of the data: ([0-9.e+-]+)", out)
    assert m and float(m.group(1)) < 0.5 * float(m.group(2)), out


def test_lds_knowns_example():
    out = _run("lds_knowns_in_a.py", "200", "30")
    assert "known row kept exactly: True" in out, out
    m = re.search(r"rms of y - <C><x> : ([0-9.e+-]+)\s+rms of y : ([0-9.e+-]+)", out)
    assert m and float(m.group(1)) < 0.5 * float(m.group(2)), out
