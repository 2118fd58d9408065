"""Random graph compositions (tests/golden/generic_scenarios.py: random_graph) beyond the 36 that have reference fixtures: the HIP
tape interpreter (per-node tapes and Network.learn-style many-node tapes with their side-by-side programs) against the numpy
restatement of the interpreter on the same graph.     python profiles/fuzz_generic.py [n] [first seed]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import numpy as np
import generic_scenarios as GS
from oracle.tape_ref import NumpyExecutor
from pyvb_amd import nodes, generic, _recognise


def run(seed, factory, batched):
    _recognise.bind = lambda node: generic.GenericPlan(node, executor_factory=factory)
    order, named = GS.random_graph(nodes, np.random.default_rng(seed))
    for _ in range(2):
        if batched:
            plans = []
            for n in order:
                p = nodes._plan_of(n)
                if p not in plans:
                    plans.append(p)
            for p in plans:
                p.update_all([n for n in order if n._plan is p])
        else:
            for n in order:
                n.update()
    snap = GS.snapshot(named)
    llb = GS.lower_bounds(named)
    return snap, llb


n = int(sys.argv[1]) if len(sys.argv) > 1 else 100
first = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
worst = 0.0
for seed in range(first, first + n):
    try:
        ref, rl = run(seed, NumpyExecutor, False)
    except (AssertionError, NotImplementedError, np.linalg.LinAlgError) as e:
        print("seed %d: the emitters refuse this graph (%s)" % (seed, str(e)[:60]), flush=True)
        continue
    err = 0.0
    for batched in (False, True):
        got, gl = run(seed, None, batched)
        for k in ref:
            a, b = np.asarray(got[k], float), np.asarray(ref[k], float)
            err = max(err, float(np.abs(a - b).max() / max(np.abs(b).max(), 1.0)))
        for k in rl:
            if np.isfinite(rl[k]):
                err = max(err, abs(gl[k] - rl[k]) / max(abs(rl[k]), 1.0))
    assert err < 1e-9, (seed, err)
    worst = max(worst, err)
print("%d random graphs, device (per node and batched) vs numpy interpreter: worst %.2e" % (n, worst))
