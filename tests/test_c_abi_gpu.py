"""The C ABI from a host program written in C (tests/c/abi_smoke.c): compiled with gcc against include/pyvb_hip.h and
libpyvb_hip.so, run on the GPU, compared with the Python front end on the same inputs.  (CPU part: it compiles and links.)"""
import os
import re
import subprocess

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _build(tmp_path):
    exe = str(tmp_path / "abi_smoke")
    lib = os.path.join(REPO, "pyvb_amd")
    cmd = ["gcc", "-std=c99", "-Wall", "-Werror", "-I", os.path.join(REPO, "include"), os.path.join(REPO, "tests", "c", "abi_smoke.c"),
           "-o", exe, "-L", lib, "-lpyvb_hip", "-lm", "-Wl,-rpath," + lib, "-Wl,-rpath,/opt/rocm/lib"]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    return exe


def test_header_compiles_as_c99_and_the_library_links(tmp_path):
    _build(tmp_path)


@pytest.mark.gpu
def test_c_host_program_matches_the_python_front_end(tmp_path):
    from pyvb_amd import synth
    from pyvb_amd.lds import LDSBatch
    exe = _build(tmp_path)
    N, T, D, K = 3, 120, 5, 7
    Y, st0, pri = synth.make_problem(T, D, K, N, seed=321)
    path = tmp_path / "problem.bin"
    with open(path, "wb") as f:
        np.array([N, T, D, K], dtype=np.float64).tofile(f)
        for a in (Y, st0["X"], st0["A_mean"], st0["A_colvar"], st0["C_mean"], st0["C_colvar"], st0["Q_b"], st0["R_b"]):
            np.ascontiguousarray(a, dtype=np.float64).tofile(f)
    r = subprocess.run([exe, str(path)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    got = [float(v) for v in re.findall(r"lower bound (\S+)", r.stdout)]
    b = LDSBatch.from_problem(Y, st0, pri)
    b.iterate(2)
    hist = b.elbo_history(2).sum(1)
    assert got == [hist[0], hist[1]], (got, hist)                 # same library, same inputs: bitwise
    x = float(re.search(r"x\[0\]\[T-1\]\[0\] (\S+)", r.stdout).group(1))
    assert x == b.get_state(("X",))["X"][0, T - 1, 0]
    b.close()
