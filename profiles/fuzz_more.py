"""One-off fuzz of the paths fuzz_shapes.py does not cover: outputs with missing entries (with and without update_Y between
iterations), known entries of A / C, the VB-PCA path at random shapes, all against the oracles.
usage: python profiles/fuzz_more.py [n_cases] [seed]"""
import importlib.util, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from pyvb_amd import synth
from pyvb_amd.lds import LDSBatch
from pyvb_amd.pca import PCABatch
from oracle import lds_closed_form as O
from oracle import pca_closed_form as P

spec = importlib.util.spec_from_file_location("make_golden", os.path.join(ROOT, "tests", "golden", "make_golden.py"))
G = importlib.util.module_from_spec(spec); spec.loader.exec_module(G)
ncase = int(sys.argv[1]) if len(sys.argv) > 1 else 30
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
rel = lambda a, b: float(np.abs(np.asarray(a) - np.asarray(b)).max() / max(np.abs(b).max(), 1e-300))
worst = 0.0
for case in range(ncase):
    D = int(rng.choice([1, 2, 3, 7, 16, 17, 31, 33, 48, 64])); K = int(rng.choice([1, 2, 5, 16, 17, 33, 50, 64]))
    T = int(rng.choice([2, 3, 5, 18, 33, 97, 257, 700])); N = int(rng.choice([1, 2, 3]))
    kind = str(rng.choice(["diagonal_gamma", "gamma"]))
    Y, st0, pri = synth.make_problem(T, D, K, N, seed=3000 + case)
    pri["noise"] = kind
    if kind == "gamma":
        for k in ("Q_a0", "Q_b0", "R_a0", "R_b0"):
            pri[k] = np.float64(1e-3)
    missing = rng.random() < 0.6
    knowns = rng.random() < 0.4
    if missing:
        mask = rng.random((N, T, K)) < rng.choice([0.05, 0.3, 0.9])
        if T > 2:
            mask[:, 1] = True
        Y = np.where(mask, np.nan, Y)
        st0["Yq"] = rng.standard_normal((N, T, K)); st0["Yrowvar"] = 1.0 / rng.uniform(0.5, 1.5, size=(N, T))
        missing = bool(np.isnan(Y).any())
    if knowns:
        A_obs = np.where(rng.random((D, D)) < 0.2, rng.standard_normal((D, D)), np.nan)
        C_obs = np.where(rng.random((K, D)) < 0.2, rng.standard_normal((K, D)), np.nan)
        if D > 1:
            C_obs[:, 0] = rng.standard_normal(K)        # a fully known column
        pri["A_obs"], pri["C_obs"] = A_obs, C_obs
    b = LDSBatch.from_problem(Y, st0, pri)
    st = O.expand_state(st0, pri, T, Y)
    err = 0.0
    for it in range(3):
        upd = missing and it != 1
        ref = O.iterate(st, pri, Y, update_outputs=False)
        b.iterate(1)
        if upd:
            O.update_Y(st, pri); b.update_Y()
        err = max(err, rel(b.get_state(("X",))["X"], st["X"]))
    S = O.statistics(st, Y)
    ref = O.elbo_parts(st, pri, S, T)
    got = b.elbo()
    err = max(err, float(np.max(np.abs(got - ref).sum(1) / np.abs(ref).sum(1))))
    g = b.get_state(("A_mean", "C_mean", "Q_b", "R_b"))
    err = max(err, rel(g["A_mean"], st["A_mean"]), rel(g["C_mean"], st["C_mean"]))
    b.close()
    worst = max(worst, err)
    print("LDS T=%4d D=%2d K=%2d N=%d %-15s missing=%d knowns=%d rel err %.2e" % (T, D, K, N, kind, missing, knowns, err), flush=True)
    assert err < 1e-8, "mismatch"
for case in range(ncase // 2):
    d = int(rng.choice([1, 2, 5, 16, 17, 70, 128, 255, 256])); q = int(rng.choice([1, 2, 3, 16, 17, 31, 32])); q = min(q, max(1, d))
    N = int(rng.choice([2, 3, 17, 64, 65, 500, 4097]))
    init, pri = G.pca_problem(N, d, q, seed=5000 + case, p_missing=float(rng.choice([0.0, 0.1, 0.6])))
    st = P.make_state(init, pri, N, d, q)
    b = PCABatch.from_problem(init, pri)
    err = 0.0
    for it in range(2):
        ref = P.iterate(st, pri); b.iterate(1)
        got = b.elbo()
        err = max(err, float(np.abs(got - ref).sum() / np.abs(ref).sum()))
    g = b.get_state()
    err = max(err, rel(g["W_mean"], st["W_mean"]), rel(g["Z"], st["Z"]), rel(g["X"], st["X"]))
    b.close()
    worst = max(worst, err)
    print("PCA N=%5d d=%3d q=%2d rel err %.2e" % (N, d, q, err), flush=True)
    assert err < 1e-8, "mismatch"
print("worst", worst)
