"""Random update orders through the C ABI of the VB-PCA path (PCABatch) against the oracle: W, Z, X over random row ranges,
Mu, Beta, the lower bound, in any order; explicit and constructor-drawn initial rows.   python profiles/fuzz_batch_pca.py [cases] [seed]"""
import importlib.util, os, sys
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import numpy as np
from pyvb_amd.pca import PCABatch
from oracle import pca_closed_form as P

spec = importlib.util.spec_from_file_location("make_golden", os.path.join(os.path.dirname(HERE), "tests", "golden", "make_golden.py"))
G = importlib.util.module_from_spec(spec); spec.loader.exec_module(G)


def rel(a, b):
    return float(np.abs(np.asarray(a) - np.asarray(b)).max() / max(np.abs(np.asarray(b)).max(), 1e-300))


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 30
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
    only = int(sys.argv[3]) if len(sys.argv) > 3 else None
    worst = 0.0
    for case in range(cases):
        N = int(rng.choice([5, 16, 17, 100, 1000, 5000])); d = int(rng.choice([2, 7, 16, 33, 64, 250])); q = int(rng.integers(1, min(d, 20) + 1))
        init, pri = G.pca_problem(N, d, q, int(rng.integers(1 << 30)), p_missing=float(rng.choice([0.0, 0.1, 0.5])))
        drawn = bool(rng.random() < 0.5)
        if drawn:       # rows as their constructors would have drawn them
            init["X_full"] = np.where(init["obs"].all(1)[:, None], init["X"], rng.standard_normal((N, d)))
            init["X_var0"] = np.where(init["obs"].all(1), 0.0, 1.0 / rng.random(N))
        ops = [str(rng.choice(["W", "Z", "X", "Xall", "X0", "Mu", "Beta", "elbo"], p=[.15, .15, .15, .15, .05, .15, .15, .05])) for _ in range(14)]
        ranges = []
        for op in ops:
            lo = 0 if op != "X" else int(rng.integers(0, N))
            ranges.append((lo, N if op == "Xall" else (1 if op == "X0" else int(rng.integers(lo, N + 1)))))
        if only is not None and case != only:
            continue
        b = PCABatch.from_problem(init, pri)
        st = P.make_state(init, pri, N, d, q)
        err = 0.0
        for k_op, op in enumerate(ops):
            if op == "W": P.update_W(st, pri); b.update_W()
            elif op == "Z": P.update_Z(st, pri); b.update_Z()
            elif op in ("X", "Xall", "X0"):
                lo, hi = ranges[k_op]
                P.update_X(st, pri, lo, hi); b.update_X(lo, hi)
            elif op == "Mu": P.update_Mu(st, pri); b.update_Mu()
            elif op == "Beta": P.update_Beta(st, pri); b.update_Beta()
            elif op == "elbo":
                none = (~st["obs"]).all(1)
                if np.isnan(st["qld_W"]).any() or np.isnan(st["qld_Z"]) or np.isnan(st["qld_Mu"]) or (st["X_var"][none, 0] == 0).any():
                    continue        # a node without q_ln_det yet: the reference raises AttributeError
                ref = P.elbo_parts(st, pri)             # the five class sums cancel: compare on their scale
                got = b.elbo()
                if only is not None:
                    print("  elbo got %s\n       ref %s" % (got.tolist(), ref.tolist()), flush=True)
                err = max(err, float(np.abs(got - ref).max() / np.abs(ref).sum()))
            g = b.get_state()
            if only is not None:
                print("  after %-5s %s X %.1e Z %.1e W %.1e Mu %.1e beta %.1e" % (op, ranges[k_op] if op[0] == "X" else "", rel(g["X"], st["X"]), rel(g["Z"], st["Z"]),
                      rel(g["W_mean"], st["W_mean"]), rel(g["Mu_mean"], st["Mu_mean"]), abs(g["beta_b"] - st["beta_b"]) / abs(st["beta_b"])), flush=True)
            err = max(err, rel(g["X"], st["X"]), rel(g["Z"], st["Z"]), rel(g["W_mean"], st["W_mean"]), rel(g["Mu_mean"], st["Mu_mean"]),
                      abs(g["beta_b"] - st["beta_b"]) / abs(st["beta_b"]))
            assert err < 1e-7, "case %d N=%d d=%d q=%d drawn=%d after %s of %s: rel err %.3e" % (case, N, d, q, drawn, op, ops, err)
        b.close()
        print("case %2d N=%4d d=%3d q=%2d drawn=%d  %s  worst %.2e" % (case, N, d, q, drawn, " ".join(ops), err), flush=True)
        worst = max(worst, err)
    print("worst", worst)


if __name__ == "__main__":
    main()
