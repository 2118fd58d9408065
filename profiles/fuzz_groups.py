"""Differential fuzz of graphs that SHARE a device handle (pyvb_amd/_recognise.py: LDSGroup): M LDS graphs of one structure are
built side by side and driven by a random interleaving of node operations -- each operation goes to one graph, to a random
subset or to all of them -- so that their request queues agree, part ways and meet again in every combination; a twin of every
graph, forced onto the node-by-node plan and driven graph by graph, has to give the same reads.

    python profiles/fuzz_groups.py [cases] [seed]
"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pyvb_amd import nodes, generic, _recognise
from fuzz_ops import build, apply, forced_generic


def main(cases=None, seed=None, noises=("gamma", "diagonal_gamma", "wishart"), noise_p=(0.4, 0.45, 0.15), allow_missing=True):
    """(tests/test_groups_cpu.py calls this with the handle replaced by its oracle-backed stand-in: plain noise kinds only)"""
    cases = cases if cases is not None else (int(sys.argv[1]) if len(sys.argv) > 1 else 30)
    rng = np.random.default_rng(seed if seed is not None else (int(sys.argv[2]) if len(sys.argv) > 2 else 0))
    worst, shared_seen, left_seen = 0.0, 0, 0
    kinds = ["fwd", "bwd", "As", "Cs", "Q", "R", "iter", "read_x", "read_a", "read_q", "x", "a", "set_a", "set_x", "llb_a", "learn", "Ys", "read_y"]
    w = np.array([4, 4, 3, 3, 2, 2, 4, 3, 2, 2, 0.6, 1, 0.7, 0.7, 0.7, 1, 1, 1], float)
    for case in range(cases):
        M = int(rng.integers(2, 6))
        T = int(rng.integers(3, 30)); q = int(rng.integers(1, 5)); d = int(rng.integers(2, 6))
        noise = str(rng.choice(list(noises), p=list(noise_p)))
        missing = allow_missing and noise != "wishart" and rng.random() < 0.25
        pattern = rng.random((T, d)) < 0.2
        pattern[0] = False
        fused, slow = [], []
        for m in range(M):
            Y = rng.standard_normal((T, d))
            if missing:
                Y[pattern] = np.nan                     # the same outputs are unobserved in every graph: one signature
            seed = int(rng.integers(1 << 30))
            fused.append(build(seed, T, q, d, noise, Y, False))
            with forced_generic():
                slow.append(build(seed, T, q, d, noise, Y, False))
        n_ops = 30
        log = []
        err = 0.0
        # every graph starts with a forward sweep (queued: nothing is read yet), so that they are bound together
        script = [("fwd", 0, 0, None, list(range(M)))]
        for _ in range(n_ops):
            k = str(rng.choice(kinds, p=w / w.sum()))
            r = rng.random()
            who = list(range(M)) if r < 0.5 else ([int(rng.integers(0, M))] if r < 0.8 else sorted(set(int(v) for v in rng.integers(0, M, size=M))))
            script.append((k, int(rng.integers(0, T)), int(rng.integers(0, q)), rng.standard_normal(d), who))
        print("case %2d M=%d T=%2d q=%d d=%d %-14s missing=%d: %s" % (case, M, T, q, d, noise, int(missing), " ".join("%s@%s" % (o[0], "*" if len(o[4]) == M else ",".join(map(str, o[4]))) for o in script)), flush=True)
        for n, (k, t, i, vec, who) in enumerate(script):
            op = (k, t, i, vec)
            outs = {}
            try:
                for m in who:           # first the requests to all addressed graphs (queued side by side) ...
                    outs[m] = apply(fused[m], op)
                for m in who:
                    with forced_generic():
                        b = apply(slow[m], op)
                    a = outs[m]
                    if a is None:
                        continue
                    for u, v in zip(a, b):
                        u, v = np.asarray(u, float), np.asarray(v, float)
                        if not np.all(np.isfinite(v)):
                            continue
                        assert np.all(np.isfinite(u)), (case, n, k, m)
                        e = float(np.abs(u - v).max() / max(np.abs(v).max(), 1e-3))
                        err = max(err, e)
                        assert e < 1e-7, "case %d op %d %s graph %d: rel err %.3e" % (case, n, k, m, e)
            except Exception:
                print("case %d failed at op %d (%s, t=%d, i=%d, graphs %r)" % (case, n, k, t, i, who), flush=True)
                raise
            groups = set()
            for g in fused:
                p = g["Xs"][0]._plan
                grp = getattr(p, "group", None)
                if grp is not None:
                    groups.add(id(grp))
                    if len(grp.live()) > 1:
                        shared_seen += 1
                    if len(grp.live()) < len(grp.members):
                        left_seen += 1
        # the end state of every graph, whatever plan it ended on
        for m in range(M):
            a = [np.hstack([x.qmu for x in fused[m]["Xs"]]), np.hstack([c.qmu for c in fused[m]["As"]]), np.hstack([c.qmu for c in fused[m]["Cs"]])]
            with forced_generic():
                b = [np.hstack([x.qmu for x in slow[m]["Xs"]]), np.hstack([c.qmu for c in slow[m]["As"]]), np.hstack([c.qmu for c in slow[m]["Cs"]])]
            for u, v in zip(a, b):
                e = float(np.abs(u - v).max() / max(np.abs(v).max(), 1e-3))
                err = max(err, e)
                assert e < 1e-7, "case %d end state of graph %d: rel err %.3e" % (case, m, e)
        plans = [type(g["Xs"][0]._plan).__name__[:3] + (":%d" % len(g["Xs"][0]._plan.group.live()) if getattr(g["Xs"][0]._plan, "group", None) is not None else "") for g in fused]
        print("case %2d plans at the end %s  worst rel err %.2e" % (case, " ".join(plans), err), flush=True)
        worst = max(worst, err)
    print("operations seen with a handle shared by several graphs: %d, with a handle some graph had left: %d" % (shared_seen, left_seen))
    print("worst", worst)
    return worst, shared_seen, left_seen


if __name__ == "__main__":
    main()
