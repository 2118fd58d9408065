"""One-off fuzz of the LDS path: random (T, D, K, N) against the oracle, two iterations each.
usage: python profiles/fuzz_shapes.py [n_cases] [seed]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pyvb_amd import synth
from pyvb_amd.lds import LDSBatch
from oracle import lds_closed_form as O

ncase = int(sys.argv[1]) if len(sys.argv) > 1 else 30
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
worst = 0.0
for case in range(ncase):
    D = int(rng.choice([1, 2, 3, 15, 16, 17, 31, 32, 33, 47, 48, 49, 50, 57, 63, 64]))
    K = int(rng.choice([1, 2, 5, 16, 17, 32, 33, 48, 49, 60, 64]))
    T = int(rng.choice([2, 3, 4, 17, 18, 19, 33, 50, 97, 160, 257, 514, 600, 1111, 3000]))
    N = int(rng.choice([1, 2, 3]))
    noise = "gamma" if rng.random() < 0.25 else "diagonal_gamma"
    Y, st0, pri = synth.make_problem(T, D, K, N, seed=1000 + case)
    try:
        b = LDSBatch.from_problem(Y, st0, pri, noise=noise) if noise == "gamma" else LDSBatch.from_problem(Y, st0, pri)
    except TypeError:
        noise = "diagonal_gamma"
        b = LDSBatch.from_problem(Y, st0, pri)
    st = O.expand_state(st0, pri, T)
    err = 0.0
    for it in range(2):
        ref = O.iterate(st, pri, Y) if noise != "gamma" else None
        if ref is None:
            break
        b.iterate(1)
        X = b.get_state(("X",))["X"]
        err = max(err, float(np.abs(X - st["X"]).max() / max(np.abs(st["X"]).max(), 1e-300)))
        e = b.elbo().sum(1); r = ref.sum(1)
        err = max(err, float(np.max(np.abs(e - r) / np.abs(r))))
    b.close()
    worst = max(worst, err)
    print("T=%4d D=%2d K=%2d N=%d %-15s rel err %.2e" % (T, D, K, N, noise, err), flush=True)
    assert err < 1e-8, "mismatch"
print("worst", worst)
