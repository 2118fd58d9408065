"""The reference's example loop (examples/Linear_Dynamic_System.py:69-77 + the lower bound) through the
pyvb-compatible node API, timed per iteration, for the example's own shape and config 1's."""
import importlib.util, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from pyvb_amd import nodes, synth

spec = importlib.util.spec_from_file_location("make_golden", os.path.join(ROOT, "tests", "golden", "make_golden.py"))
G = importlib.util.module_from_spec(spec); spec.loader.exec_module(G)
for (T, D, K) in [(200, 2, 5), (200, 4, 5), (2000, 8, 8)]:
    Y, st0, pri = synth.make_problem(T, D, K, 1, seed=11)
    g = G.build_graph(nodes, Y[0], pri, {k: v for k, v in st0.items()})
    Xs, As, Cs, Q, R = g["Xs"], g["As"], g["Cs"], g["Q"], g["R"]
    allnodes = Xs + g.get("Ys", []) + As + Cs + [Q, R]

    def one():
        [x.update() for x in Xs]
        Xs.reverse()
        [x.update() for x in Xs]
        Xs.reverse()
        [a.update() for a in As]
        [c.update() for c in Cs]
        Q.update(); R.update()
        return float(np.asarray(Q.qb).sum())          # a read: forces the queued updates to run

    for _ in range(3):
        one()
    t0 = time.perf_counter()
    iters = 30
    for _ in range(iters):
        one()
    dt = (time.perf_counter() - t0) / iters
    print("node API, T=%d D=%d K=%d: %.3f ms per iteration" % (T, D, K, dt * 1e3), flush=True)
