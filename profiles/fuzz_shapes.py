"""One-off fuzz of the LDS path: random (T, D, K, N, noise kind incl. Wishart) against the oracle, two iterations each.
usage: python profiles/fuzz_shapes.py [n_cases] [seed] [big]      (big: the second shape class, 64 < max(D, K) <= 128, short chains --
the oracle is O(D^3) per node --, known entries of A / C and outputs with NaN mixed in)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pyvb_amd import synth
from pyvb_amd.lds import LDSBatch
from oracle import lds_closed_form as O

ncase = int(sys.argv[1]) if len(sys.argv) > 1 else 30
BIG = len(sys.argv) > 3 and sys.argv[3] == "big"
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
worst = 0.0
for case in range(ncase):
    D = int(rng.choice([1, 2, 3, 15, 16, 17, 31, 32, 33, 47, 48, 49, 50, 57, 63, 64]))
    K = int(rng.choice([1, 2, 5, 16, 17, 32, 33, 48, 49, 60, 64]))
    T = int(rng.choice([2, 3, 4, 17, 18, 19, 33, 50, 97, 160, 257, 514, 600, 1111, 3000]))
    N = int(rng.choice([1, 2, 3]))
    noise = str(rng.choice(["diagonal_gamma", "diagonal_gamma", "gamma", "wishart"]))
    if BIG:
        D = int(rng.choice([3, 40, 65, 66, 80, 96, 100, 127, 128])); K = int(rng.choice([2, 33, 65, 70, 97, 128]))
        if max(D, K) <= 64:
            D = 65
        T = int(rng.choice([2, 3, 4, 5, 9, 18, 35])); N = int(rng.choice([1, 2]))
        noise = str(rng.choice(["diagonal_gamma", "gamma"]))            # Wishart noise stops at 64 on the fused kernels
    Y, st0, pri = synth.make_problem(T, D, K, N, seed=1000 + case)
    if BIG and max(D, K) > 102:       # quirk Q2: det(1e-3 I) underflows from 103 dimensions on
        pri["A_prior_prec"] = np.full_like(pri["A_prior_prec"], 1e-2); pri["C_prior_prec"] = np.full_like(pri["C_prior_prec"], 1e-2)
    if BIG and case % 3 == 1 and T >= 4:
        mask = rng.random((N, T, K)) < 0.15
        mask[:, 1] = True; mask[:, 0, 0] = True; mask[:, 3] = False
        Y = np.where(mask, np.nan, Y)
        st0["Yq"] = rng.standard_normal((N, T, K)); st0["Yrowvar"] = 1.0 / rng.uniform(0.5, 1.5, size=(N, T))
    if BIG and case % 3 == 2:
        pri["A_obs"] = np.where(rng.random((D, D)) < 0.1, rng.standard_normal((D, D)) * 0.2, np.nan)
        pri["C_obs"] = np.where(rng.random((K, D)) < 0.1, rng.standard_normal((K, D)), np.nan)
    pri["noise"] = noise
    if noise == "gamma":
        for k in ("Q_a0", "Q_b0", "R_a0", "R_b0"):
            pri[k] = np.float64(1e-3)
    elif noise == "wishart":
        W = rng.standard_normal((D, D)); pri["Q_b0"] = 0.05 * (W @ W.T + D * np.eye(D)); pri["Q_a0"] = np.float64(0.5 * D + 1.0)
        W = rng.standard_normal((K, K)); pri["R_b0"] = 0.05 * (W @ W.T + K * np.eye(K)); pri["R_a0"] = np.float64(0.5 * K + 0.5)
    b = LDSBatch.from_problem(Y, st0, pri)
    missing = bool(np.isnan(Y).any())
    st = O.expand_state(st0, pri, T, Y) if missing else O.expand_state(st0, pri, T)
    err = 0.0
    for it in range(2):
        if missing:                 # the output nodes are updated first: before that their bound terms are NaN on both sides
            O.update_Y(st, pri); b.update_Y()
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
