"""Timing of the generic node-by-node device path (tape interpreter) on graphs the fused plans would otherwise take:
python profiles/generic_probe.py    (the LDS example shape and a small PCA, forced onto GenericPlan)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pyvb_amd import nodes, generic, _recognise, network

_recognise.bind = lambda node: generic.GenericPlan(node)      # no fused plans in this process


def lds(T, q, d, iters):
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
    # an iteration ends with a read (Network.learn evaluates the bound; here the first state's mean): the queued requests of one
    # iteration are one program, built -- and scheduled -- at its first occurrence and reused afterwards
    t0 = time.time(); it(); _ = Xs[0].qmu; t1 = time.time()
    it(); _ = Xs[0].qmu
    t1b = time.time()
    for _i in range(iters):
        it(); _ = Xs[0].qmu
    t2 = time.time()
    print("generic LDS T=%d q=%d d=%d: first iteration (tapes built) %.2f s, then %.1f ms per iteration" % (T, q, d, t1 - t0, (t2 - t1b) / iters * 1e3), flush=True)


def pca(N, d, q, iters):
    rng = np.random.default_rng(1)
    X = rng.standard_normal((N, q)) @ rng.standard_normal((q, d)) + 0.1 * rng.standard_normal((N, d))
    X[rng.random((N, d)) < 0.1] = np.nan
    Ws = [nodes.Gaussian(d, np.zeros((d, 1)), np.eye(d) * 1e-3) for _ in range(q)]
    W = nodes.hstack(Ws)
    Mu = nodes.Gaussian(d, np.zeros((d, 1)), np.eye(d) * 1e-3)
    Beta = nodes.Gamma(d, 1e-3, 1e-3)
    Zs = [nodes.Gaussian(q, np.zeros((q, 1)), np.eye(q)) for _ in range(N)]
    Xs = [nodes.Gaussian(d, W * z + Mu, Beta) for z in Zs]
    for x, row in zip(Xs, X):
        x.observe(row.reshape(d, 1))
    net = network.Network()
    net.addnode(Xs[0]); net.fetch_network(verbose=False)
    t0 = time.time(); net.learn(1, tol=0, verbose=False); t1 = time.time()
    net.learn(iters, tol=0, verbose=False)
    t2 = time.time()
    print("generic PCA N=%d d=%d q=%d through Network.learn: first iteration %.2f s, then %.1f ms per iteration" % (N, d, q, t1 - t0, (t2 - t1) / iters * 1e3), flush=True)

    # the same updates as a hand-written loop (examples/PCA_missing_data.py style, without the lower bound): the requests queue up
    # and are issued as one tape per pass, independent nodes side by side
    def it():
        [w.update() for w in Ws]; [z.update() for z in Zs]; [x.update() for x in Xs]; Mu.update(); Beta.update()
    t0 = time.time(); it(); _ = Mu.qmu; t1 = time.time()
    it(); _ = Mu.qmu
    t1b = time.time()
    for _i in range(iters):
        it(); _ = Mu.qmu
    t2 = time.time()
    print("generic PCA N=%d d=%d q=%d as a hand-written loop: first pass %.2f s, then %.1f ms per pass" % (N, d, q, t1 - t0, (t2 - t1b) / iters * 1e3), flush=True)


lds(200, 2, 5, 5)
lds(1000, 4, 8, 3)
pca(200, 10, 3, 5)
pca(2000, 10, 3, 3)
