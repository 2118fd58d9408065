"""Random update orders through the C ABI (LDSBatch) against the oracle: sweeps in either direction any number of times, the
sweep written out as T single pyvb_lds_update_x calls, column updates over random ranges, noise updates, the lower bound --
in any order after the first sweep.      python profiles/fuzz_batch.py [cases] [seed] [only | big]
(big: shapes of the second class, 64 < max(D, K) <= 128, short chains -- the oracle is O(D^3) per node -- so that the blocked column kernel,
the 128-wide inversions and sweeps meet random update orders, column ranges, known entries and outputs with NaN)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pyvb_amd import synth
from pyvb_amd.lds import LDSBatch
from oracle import lds_closed_form as O


def rel(a, b):
    return float(np.abs(np.asarray(a) - np.asarray(b)).max() / max(np.abs(np.asarray(b)).max(), 1e-300))


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 30
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
    big = len(sys.argv) > 3 and sys.argv[3] == "big"
    only = int(sys.argv[3]) if len(sys.argv) > 3 and not big else None
    worst = 0.0
    for case in range(cases):
        T = int(rng.choice([2, 3, 5, 17, 40, 130, 600])); D = int(rng.integers(1, 20)); K = int(rng.integers(1, 20)); N = int(rng.integers(1, 4))
        if rng.random() < 0.15:
            D, K = int(rng.choice([33, 64])), int(rng.choice([48, 64]))
        kind = str(rng.choice(["diagonal_gamma", "gamma", "wishart"], p=[0.5, 0.25, 0.25]))
        if big:
            D = int(rng.choice([3, 20, 65, 70, 96, 100, 127, 128])); K = int(rng.choice([2, 33, 65, 90, 128]))
            if max(D, K) <= 64:
                K = 70
            T = int(rng.choice([2, 3, 5, 9, 20])); N = int(rng.integers(1, 3))
            kind = str(rng.choice(["diagonal_gamma", "gamma"]))
        Y, st0, pri = synth.make_problem(T, D, K, N, seed=int(rng.integers(1 << 30)))
        if big and max(D, K) > 102:       # quirk Q2: det(1e-3 I) underflows from 103 dimensions on
            pri["A_prior_prec"] = np.full_like(pri["A_prior_prec"], 1e-2); pri["C_prior_prec"] = np.full_like(pri["C_prior_prec"], 1e-2)
        pri["noise"] = kind
        if kind == "gamma":
            for k in ("Q_a0", "Q_b0", "R_a0", "R_b0"):
                pri[k] = np.float64(1e-3)
        if kind == "wishart":       # proper priors: v0 > (dim - 1) / 2, dense w0
            W = rng.standard_normal((D, D)); pri["Q_b0"] = 0.05 * (W @ W.T + D * np.eye(D)); pri["Q_a0"] = np.float64(0.5 * D + 1.0)
            W = rng.standard_normal((K, K)); pri["R_b0"] = 0.05 * (W @ W.T + K * np.eye(K)); pri["R_a0"] = np.float64(0.5 * K + 0.5)
        missing = kind != "wishart" and rng.random() < 0.35
        knowns = kind != "wishart" and rng.random() < 0.3
        if missing:         # outputs with missing entries are nodes of their own (op "Y": [y.update() for y in Ys if not y.observed])
            mask = rng.random((N, T, K)) < rng.choice([0.05, 0.3, 0.9])
            if T > 2:
                mask[:, 1] = True
            Y = np.where(mask, np.nan, Y)
            st0["Yq"] = rng.standard_normal((N, T, K)); st0["Yrowvar"] = 1.0 / rng.uniform(0.5, 1.5, size=(N, T))
        if knowns:          # known entries of A and C (examples/LDS_knowns_in_A.py:73-74)
            A_obs = np.where(rng.random((D, D)) < 0.2, rng.standard_normal((D, D)), np.nan)
            C_obs = np.where(rng.random((K, D)) < 0.2, rng.standard_normal((K, D)), np.nan)
            if D > 1:
                C_obs[:, 0] = rng.standard_normal(K)        # a fully known column
            pri["A_obs"], pri["C_obs"] = A_obs, C_obs
        ops = ["fwd" if rng.random() < 0.7 else "bwd"] + [str(rng.choice(["fwd", "bwd", "xs", "A", "C", "Acols", "Ccols", "Q", "R", "elbo", "Y"],
                                                                         p=[.13, .13, .07, .1, .1, .11, .11, .08, .08, .04, .05])) for _ in range(14)]
        ranges = [(lambda c0: (c0, int(rng.integers(c0 + 1, D + 1))))(int(rng.integers(0, D))) for _ in ops]
        if only is not None and case != only:
            continue
        b = LDSBatch.from_problem(Y, st0, pri)
        st = O.expand_state(st0, pri, T, Y)
        err = 0.0
        for k_op, op in enumerate(ops):
            if op in ("fwd", "bwd"):
                O.sweep(st, pri, Y, "forward" if op == "fwd" else "backward"); b.sweep("forward" if op == "fwd" else "backward")
            elif op == "xs":
                for t in range(T):
                    O.update_x(st, pri, Y, t); b.update_x(t)
            elif op in ("A", "C", "Acols", "Ccols"):
                S = O.statistics(st, Y)
                cols = None
                if op.endswith("cols"):
                    cols = ranges[k_op]
                (O.update_A if op[0] == "A" else O.update_C)(st, pri, S, cols)
                b.update_columns(op[0], *(cols or (0, D)))
            elif op in ("Q", "R"):
                S = O.statistics(st, Y)
                (O.update_Q if op == "Q" else O.update_R)(st, pri, S, T); (b.update_Q if op == "Q" else b.update_R)()
            elif op == "Y":
                if missing:
                    O.update_Y(st, pri); b.update_Y()
            elif op == "elbo":
                if missing and np.isnan(st["Yqld"][np.isnan(Y).all(2)]).any():
                    continue                # an output nothing of which is observed has no q_ln_det before its first update
                if "qld_A" not in st or np.isnan(st["qld_A"]).any() or np.isnan(st["qld_C"]).any():
                    continue                # a column without q_ln_det yet: the reference raises AttributeError
                parts = O.elbo_parts(st, pri, O.statistics(st, Y), T)
                got = b.elbo()
                err = max(err, rel(got.sum(1), parts.sum(1)))
            g = b.get_state()
            parts = {"X": rel(g["X"], st["X"]), "A": rel(g["A_mean"], st["A_mean"]), "C": rel(g["C_mean"], st["C_mean"])}
            if kind == "wishart":
                w = b.get_wishart_state()
                parts["Qw"], parts["Rw"] = rel(w["Q_w"], st["Q_b"]), rel(w["R_w"], st["R_b"])
            if only is not None:
                print("  after %-5s %s" % (op, "  ".join("%s %.1e" % kv for kv in parts.items())), flush=True)
            err = max([err] + list(parts.values()))
            if kind == "wishart":
                pass
            else:
                err = max(err, rel(g["Q_b"], np.broadcast_to(np.asarray(st["Q_b"]).reshape(N, -1), g["Q_b"].shape)),
                          rel(g["R_b"], np.broadcast_to(np.asarray(st["R_b"]).reshape(N, -1), g["R_b"].shape)))
            # more latent than observed dimensions: the posterior precisions are ill conditioned (1e-10 between two correct
            # implementations after a few parameter updates) and the noise residuals cancel
            assert err < (1e-7 if D <= K else 1e-4), "case %d (%s T=%d D=%d K=%d N=%d) after %s of %s: rel err %.3e" % (case, kind, T, D, K, N, op, ops, err)
        b.close()
        print("case %2d %-14s T=%3d D=%2d K=%2d N=%d missing=%d knowns=%d  %s  worst %.2e" % (case, kind, T, D, K, N, missing, knowns, " ".join(ops), err), flush=True)
        worst = max(worst, err)
    print("worst", worst)


if __name__ == "__main__":
    main()
