"""One-off fuzz of the LDS path: random (T, D, K, N, noise kind incl. Wishart) against the oracle, two iterations each.
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
    noise = str(rng.choice(["diagonal_gamma", "diagonal_gamma", "gamma", "wishart"]))
    Y, st0, pri = synth.make_problem(T, D, K, N, seed=1000 + case)
    pri["noise"] = noise
    if noise == "gamma":
        for k in ("Q_a0", "Q_b0", "R_a0", "R_b0"):
            pri[k] = np.float64(1e-3)
    elif noise == "wishart":
        W = rng.standard_normal((D, D)); pri["Q_b0"] = 0.05 * (W @ W.T + D * np.eye(D)); pri["Q_a0"] = np.float64(0.5 * D + 1.0)
        W = rng.standard_normal((K, K)); pri["R_b0"] = 0.05 * (W @ W.T + K * np.eye(K)); pri["R_a0"] = np.float64(0.5 * K + 0.5)
    b = LDSBatch.from_problem(Y, st0, pri)
    st = O.expand_state(st0, pri, T)
    err = 0.0
    for it in range(2):
        ref = O.iterate(st, pri, Y)
        b.iterate(1)
        X = b.get_state(("X",))["X"]
        err = max(err, float(np.abs(X - st["X"]).max() / max(np.abs(st["X"]).max(), 1e-300)))
        e = b.elbo().sum(1); r = ref.sum(1)
        err = max(err, float(np.max(np.abs(e - r) / np.abs(r))))
        h = b.elbo_history(1)
        assert np.allclose(h[-1], b.elbo().sum(0), rtol=1e-12, atol=0), "history row differs from the parts"
    b.close()
    worst = max(worst, err)
    print("T=%4d D=%2d K=%2d N=%d %-15s rel err %.2e" % (T, D, K, N, noise, err), flush=True)
    assert err < 1e-8, "mismatch"
print("worst", worst)
