#!/usr/bin/env python3
"""A linear dynamical system learnt by variational message passing on an MI355X, written against pyvb's node API
(the flow of the reference's examples/Linear_Dynamic_System.py: simulate, build the graph, sweep, read the posteriors;
without its plots).  Swap the import for `from pyvb import nodes` and the same script drives the reference.

    python examples/lds_example.py [T] [iterations]
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pyvb_amd import nodes, synth

T = int(sys.argv[1]) if len(sys.argv) > 1 else 200
niters = int(sys.argv[2]) if len(sys.argv) > 2 else 50
q, d = 2, 5                                          # latent and observed dimension
sim = synth.simulate_lds(T, q, d, 1, seed=3)
Y = sim["Y"][0]

# parameters: the columns of A and C are Gaussian nodes, the noise precisions Gamma nodes per dimension
As = [nodes.Gaussian(q, np.zeros((q, 1)), np.eye(q) * 1e-3) for _ in range(q)]
Cs = [nodes.Gaussian(d, np.zeros((d, 1)), np.eye(d) * 1e-3) for _ in range(q)]
A, C = nodes.hstack(As), nodes.hstack(Cs)
Q = nodes.DiagonalGamma(q, np.ones(q) * 1e-3, np.ones(q) * 1e-3)
R = nodes.DiagonalGamma(d, np.ones(d) * 1e-3, np.ones(d) * 1e-3)

# the chain: x_0 ~ N(0, I), x_t ~ N(A x_{t-1}, Q), y_t ~ N(C x_t, R)
Xs = [nodes.Gaussian(q, np.zeros((q, 1)), np.eye(q))]
Ys = [nodes.Gaussian(d, C * Xs[0], R)]
for t in range(1, T):
    Xs.append(nodes.Gaussian(q, A * Xs[-1], Q))
    Ys.append(nodes.Gaussian(d, C * Xs[-1], R))
for y, row in zip(Ys, Y):
    y.observe(row.reshape(d, 1))

for it in range(niters):
    [x.update() for x in Xs]                         # one launch: forward sweep over all states
    [x.update() for x in reversed(Xs)]               # one launch: backward sweep
    [a.update() for a in As]
    [c.update() for c in Cs]
    Q.update()
    R.update()

Yhat = np.hstack([y.mean_parent.pass_down_Ex() for y in Ys]).T
print("observation noise precision, learnt :", np.round(np.diag(R.pass_down_Ex()), 1))
print("observation noise precision, true   :", np.round(1.0 / sim["R"][0], 1))
print("rms of y - <C><x> :", float(np.sqrt(np.mean((Y - Yhat) ** 2))), " rms of y :", float(np.sqrt(np.mean(Y ** 2))))
print("|eig <A>| :", np.round(np.sort(np.abs(np.linalg.eigvals(A.pass_down_Ex()))), 3),
      " true :", np.round(np.sort(np.abs(np.linalg.eigvals(sim["A"][0]))), 3))
