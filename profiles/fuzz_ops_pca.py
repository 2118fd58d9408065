"""Differential fuzz of the node API on the VB-PCA graph (examples/PCA_missing_data.py): random sequences of Network.learn
calls, single node updates, reads and per-node bounds on a graph left to the recogniser (fused PCA plan) against a twin forced
onto the generic node-by-node path.      python profiles/fuzz_ops_pca.py [cases] [seed]"""
import importlib.util, os, sys
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import numpy as np
import pyvb_amd
from pyvb_amd import generic, _recognise

spec = importlib.util.spec_from_file_location("make_golden", os.path.join(os.path.dirname(HERE), "tests", "golden", "make_golden.py"))
G = importlib.util.module_from_spec(spec); spec.loader.exec_module(G)
_bind = _recognise.bind


class forced_generic(object):
    def __enter__(self):
        _recognise.bind = lambda node: generic.GenericPlan(node)

    def __exit__(self, *a):
        _recognise.bind = _bind


def apply(g, op):
    k, n, i = op
    Ws, Zs, Xs = g["Ws"], g["Zs"], g["Xs"]
    n = n % len(Zs); i = i % len(Ws)
    if k == "learn": g["net"].learn(1 + n % 3, tol=-np.inf, verbose=False); return [np.array(g["net"].llb)]
    if k == "w": Ws[i].update()
    elif k == "Ws": [w.update() for w in Ws]
    elif k == "z": Zs[n].update()
    elif k == "Zs": [z.update() for z in Zs]
    elif k == "x": Xs[n].update()
    elif k == "Xs": [x.update() for x in Xs]
    elif k == "mu": g["Mu"].update()
    elif k == "beta": g["Beta"].update()
    elif k == "set_w": Ws[i].qmu = np.cos(np.arange(Ws[i].shape[0], dtype=float) + n).reshape(-1, 1)
    elif k == "set_z": Zs[n].qmu = np.sin(np.arange(Zs[n].shape[0], dtype=float) + i).reshape(-1, 1)
    elif k == "set_beta": g["Beta"].qb = 0.5 + (n % 7) / 3.0
    elif k == "reobs":
        if Xs[n].observed: Xs[n].observe(np.cos(np.arange(Xs[n].shape[0], dtype=float) * (i + 1)).reshape(-1, 1))
    elif k == "read_z": return [Zs[n].qmu.copy(), Zs[n].qcov.copy()]
    elif k == "read_x": return [Xs[n].qmu.copy(), np.diag(Xs[n].qcov).copy()]
    elif k == "read_w": return [Ws[i].qmu.copy(), Ws[i].qcov.copy(), g["Mu"].qmu.copy(), np.array(g["Beta"].qb)]
    elif k == "llb": return [np.array(Zs[n].log_lower_bound()), np.array(Xs[n].log_lower_bound()), np.array(Ws[i].log_lower_bound()),
                             np.array(g["Mu"].log_lower_bound()), np.array(g["Beta"].log_lower_bound())]
    return None


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
    only = int(sys.argv[3]) if len(sys.argv) > 3 else None      # run this case alone, comparing everything after every operation
    kinds = ["learn", "w", "Ws", "z", "Zs", "x", "Xs", "mu", "beta", "read_z", "read_x", "read_w", "llb", "set_w", "set_z", "set_beta", "reobs"]
    worst = 0.0
    for case in range(cases):
        N = int(rng.integers(4, 40)); d = int(rng.integers(2, 9)); q = int(rng.integers(1, min(d, 4) + 1))
        init, pri = G.pca_problem(N, d, q, int(rng.integers(1 << 30)), p_missing=float(rng.choice([0.0, 0.15, 0.4])))
        seed = int(rng.integers(1 << 30)); explicit = bool(rng.random() < 0.5)
        np.random.seed(seed)                                # the constructors draw from the global generator
        fused = G.pca_build_graph(pyvb_amd, dict(init), pri, explicit_x=explicit)
        with forced_generic():
            np.random.seed(seed)
            slow = G.pca_build_graph(pyvb_amd, dict(init), pri, explicit_x=explicit)
        friendly = rng.random() < 0.6
        w = np.array([6 if friendly else 2, 1, 1, 1, 1, 1, 1, 1, 1, 2, 2, 2, 2, 0.5, 0.5, 0.5, 0.5], float)
        if friendly:
            w[1:9] = 0.15
        ops = [(str(rng.choice(kinds, p=w / w.sum())), int(rng.integers(0, 1000)), int(rng.integers(0, 1000))) for _ in range(14)]
        err = 0.0
        if only is not None and case != only:
            continue
        for n_op, op in enumerate(ops):
            a = apply(fused, op)
            with forced_generic():
                b = apply(slow, op)
            if only is not None and a is not None and op[0] == "llb":
                print("   llb fused %s\n   llb slow  %s" % ([float(x) for x in a], [float(x) for x in b]), flush=True)
                n_, i_ = op[1] % len(fused["Zs"]), op[2] % len(fused["Ws"])
                x_ = fused["Xs"][n_]
                print("   X[%d]: observed %s partially %s obs %s  qcov diag %s" % (n_, x_.observed, x_.partially_observed,
                      getattr(x_, "obs_index", None), np.diag(x_.qcov)), flush=True)
            if only is not None:
                rows = []
                for key in ("Ws", "Zs", "Xs"):
                    for i, (u, v) in enumerate(zip(fused[key], slow[key])):
                        with forced_generic():
                            mv, cv = v.qmu.copy(), v.qcov.copy()
                        rows.append((max(np.abs(u.qmu - mv).max(), np.abs(u.qcov - cv).max()), "%s[%d]" % (key, i)))
                with forced_generic():
                    mm, bb = slow["Mu"].qmu.copy(), float(slow["Beta"].qb)
                rows += [(np.abs(fused["Mu"].qmu - mm).max(), "Mu"), (abs(float(fused["Beta"].qb) - bb), "Beta")]
                top = sorted(rows, key=lambda r: -r[0])[:3]
                print("  after op %d %s (n=%d i=%d): plan %s  %s" % (n_op, op[0], op[1], op[2], type(fused["Zs"][0]._plan).__name__,
                                                                    "  ".join("%s %.2e" % (t, d_) for d_, t in top if d_ > 1e-9)), flush=True)
            if a is not None:
                for u, v in zip(a, b):
                    u, v = np.asarray(u, float), np.asarray(v, float)
                    if not np.all(np.isfinite(v)):      # NaN on the node-by-node path: a term the reference cannot evaluate yet
                        continue                        # (q_ln_det before the node's first update: AttributeError there)
                    assert np.all(np.isfinite(u)), (case, n_op, op, u, v)
                    e = float(np.abs(u - v).max() / max(np.abs(v).max(), 1e-3))    # relative, but not to a posterior that has collapsed to ~0
                    err = max(err, e)
                    assert e < 1e-7, "case %d op %d %s: rel err %.3e; ops %s" % (case, n_op, op[0], e, [o[0] for o in ops[:n_op + 1]])
        p = type(fused["Zs"][0]._plan).__name__
        print("case %2d N=%2d d=%d q=%d plan at the end %-11s ops %s  worst %.2e" % (case, N, d, q, p, " ".join(o[0] for o in ops), err), flush=True)
        worst = max(worst, err)
    print("worst", worst)


if __name__ == "__main__":
    main()
