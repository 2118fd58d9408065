#!/usr/bin/env python3
"""A linear dynamical system some of whose transition-matrix entries are known, through the partial observation of the
Gaussian columns of A (the flow of the reference's examples/LDS_knowns_in_A.py: a mass-spring-damper whose first row of A
is known -- position integrates velocity -- and whose second row is learnt; without its plots).

    python examples/lds_knowns_in_a.py [T] [iterations]
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pyvb_amd import nodes

T = int(sys.argv[1]) if len(sys.argv) > 1 else 200
niters = int(sys.argv[2]) if len(sys.argv) > 2 else 100
q, d = 2, 5
rng = np.random.default_rng(5)

# simulate: x_{t+1} ~ N(A x_t, Q), y_t ~ N(C x_t, R)
m, k, c, dt = 1.0, 1000.0, 1.0, 1e-4
true_A = np.array([[0.0, 1.0], [-c / m, -k / m]]) * dt + np.eye(2)
true_C = rng.standard_normal((d, q))
true_R, true_Q = rng.random(d) * 0.1, rng.random(q) * 0.1
X = np.zeros((T, q)); Y = np.zeros((T, d))
X[0] = rng.standard_normal(q)
for t in range(T):
    if t:
        X[t] = true_A @ X[t - 1] + np.sqrt(true_Q) * rng.standard_normal(q)
    Y[t] = true_C @ X[t] + np.sqrt(true_R) * rng.standard_normal(d)

As = [nodes.Gaussian(q, np.zeros((q, 1)), np.eye(q) * 1e-3) for _ in range(q)]
Cs = [nodes.Gaussian(d, np.zeros((d, 1)), np.eye(d) * 1e-3) for _ in range(q)]
A, C = nodes.hstack(As), nodes.hstack(Cs)
Q = nodes.DiagonalGamma(q, np.ones(q) * 1e-3, np.ones(q) * 1e-3)
R = nodes.DiagonalGamma(d, np.ones(d) * 1e-3, np.ones(d) * 1e-3)
Xs = [nodes.Gaussian(q, np.zeros((q, 1)), np.eye(q))]
Ys = [nodes.Gaussian(d, C * Xs[0], R)]
for t in range(1, T):
    Xs.append(nodes.Gaussian(q, A * Xs[-1], Q))
    Ys.append(nodes.Gaussian(d, C * Xs[-1], R))
for y, row in zip(Ys, Y):
    y.observe(row.reshape(d, 1))

# the known elements of A: its first row (NaN marks what is to be learnt)
As[0].observe(np.array([[1.0], [np.nan]]))
As[1].observe(np.array([[dt], [np.nan]]))

for it in range(niters):
    [x.update() for x in Xs]
    [x.update() for x in reversed(Xs)]
    [a.update() for a in As]
    [c_.update() for c_ in Cs]
    Q.update()
    R.update()

EA = A.pass_down_Ex()
print("true A :", np.round(true_A, 4).tolist())
print("E[A]   :", np.round(EA, 4).tolist())
print("known row kept exactly:", bool(np.all(EA[0] == true_A[0])))
Yhat = np.hstack([y.mean_parent.pass_down_Ex() for y in Ys]).T
print("rms of y - <C><x> :", float(np.sqrt(np.mean((Y - Yhat) ** 2))), " rms of y :", float(np.sqrt(np.mean(Y ** 2))))
