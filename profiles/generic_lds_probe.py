"""The LDS example forced onto the node-by-node plan, split into host and device time: python profiles/generic_lds_probe.py [T q d iters]
(host: time of the update() calls of an iteration until the last request is queued; total: until the device has finished)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pyvb_amd import nodes, generic, _recognise

_recognise.bind = lambda node: generic.GenericPlan(node)      # no fused plans in this process
T, q, d, iters = [int(x) for x in (sys.argv[1:5] + ["200", "2", "5", "10"][len(sys.argv) - 1:])]
rng = np.random.default_rng(0)
Y = rng.standard_normal((T, d))
As = [nodes.Gaussian(q, np.zeros((q, 1)), np.eye(q) * 1e-3) for _ in range(q)]
Cs = [nodes.Gaussian(d, np.zeros((d, 1)), np.eye(d) * 1e-3) for _ in range(q)]
A, C = nodes.hstack(As), nodes.hstack(Cs)
Q = nodes.DiagonalGamma(q, np.ones(q) * 1e-3, np.ones(q) * 1e-3)
R = nodes.DiagonalGamma(d, np.ones(d) * 1e-3, np.ones(d) * 1e-3)
Xs = [nodes.Gaussian(q, np.zeros((q, 1)), np.eye(q))]
Ys = [nodes.Gaussian(d, C * Xs[0], R)]
for t in range(1, T):
    Xs.append(nodes.Gaussian(q, A * Xs[-1], Q)); Ys.append(nodes.Gaussian(d, C * Xs[-1], R))
for y, row in zip(Ys, Y):
    y.observe(row.reshape(d, 1))


def it():
    [x.update() for x in Xs]; [x.update() for x in reversed(Xs)]
    [a.update() for a in As]; [c.update() for c in Cs]; Q.update(); R.update()


it(); _ = Xs[0].qmu
it(); _ = Xs[0].qmu
host = tot = 0.0
for _i in range(iters):
    t0 = time.perf_counter(); it(); t1 = time.perf_counter(); _ = Xs[0].qmu; t2 = time.perf_counter()
    host += t1 - t0; tot += t2 - t0
plan = generic._plan_of(Xs[0]) if hasattr(generic, "_plan_of") else None
print("generic LDS T=%d q=%d d=%d: %.2f ms per iteration, of which %.2f ms until the last update() returned" % (T, q, d, tot / iters * 1e3, host / iters * 1e3))
try:
    p = nodes._plan_of(Xs[0])
    seqs = [(k, v) for k, v in p._tapes.items() if k[0] == "seq"]
    for k, v in seqs:
        prog = p._programs.get(k)
        print("  queued run of %d nodes: %d records, %s" % (len(k) - 1, len(v[1]), "program with %d launches, %d blocks" % (len(prog[1]), len(prog[0])) if prog else "one block"))
except Exception as e:
    print("  (no tape statistics: %r)" % (e,))
