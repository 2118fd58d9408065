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

# -- M graphs of the example's shape side by side: Network.learn over all their nodes against LDSBatch(M) driven directly
#    with the same operations (a forward sweep, A, C, Q, R, the lower bound read back: what learn's list order spells)
from pyvb_amd.lds import LDSBatch
from pyvb_amd.network import Network
for M in (8, 64, 256):
    T, D, K = 200, 2, 5
    probs = [synth.make_problem(T, D, K, 1, seed=900 + k) for k in range(M)]
    t0 = time.perf_counter()
    graphs = [G.build_graph(nodes, Y[0], pri, {k: v for k, v in st0.items()}) for Y, st0, pri in probs]
    t_build = time.perf_counter() - t0
    net = Network([n for g in graphs for n in g["Xs"] + g["Ys"] + g["As"] + g["Cs"] + [g["Q"], g["R"]]])
    t0 = time.perf_counter()
    net.learn(2, tol=-np.inf, verbose=False)            # binds the graphs, walks the node list once, uploads one handle
    t_first = time.perf_counter() - t0
    iters = 50
    t0 = time.perf_counter()
    net.learn(iters, tol=-np.inf, verbose=False)
    per_learn = (time.perf_counter() - t0) / iters
    Y = np.concatenate([p[0] for p in probs]); st0 = {k: np.concatenate([p[1][k] for p in probs]) for k in probs[0][1]}
    b = LDSBatch.from_problem(Y, st0, probs[0][2])

    def direct():
        b.sweep("forward"); b.update_columns("A", 0, D); b.update_columns("C", 0, D); b.update_Q(); b.update_R()
        return float(b.elbo().sum())
    direct(); direct()
    t0 = time.perf_counter()
    for _ in range(iters):
        llb = direct()
    per_direct = (time.perf_counter() - t0) / iters
    grp = graphs[0]["Xs"][0]._plan.group
    print("%d graphs (T=%d D=%d K=%d): handle N=%d | build %.2f s, first learn(2) %.2f s | Network.learn %.3f ms per iteration, "
          "LDSBatch(%d) directly %.3f ms: ratio %.2f | llb %.9e vs %.9e" % (M, T, D, K, grp.batch.N, t_build, t_first, per_learn * 1e3, M,
                                                                        per_direct * 1e3, per_learn / per_direct, net.llb, llb), flush=True)
    b.close()
