#!/usr/bin/env python3
"""Variational PCA with missing entries on an MI355X through pyvb's node / network API (the flow of the reference's
examples/PCA_missing_data.py without its plots): X_n ~ N(W z_n + mu, beta), a tenth of the entries NaN.

    python examples/pca_missing_data.py [N] [iterations]
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pyvb_amd import nodes
from pyvb_amd.network import Network

N = int(sys.argv[1]) if len(sys.argv) > 1 else 400
niters = int(sys.argv[2]) if len(sys.argv) > 2 else 40
d, q, prec = 5, 2, 50.0
rng = np.random.default_rng(1)
W_true, Z_true, mean_true = rng.standard_normal((d, q)), rng.standard_normal((N, q)), rng.standard_normal(d)
X = Z_true @ W_true.T + mean_true + rng.standard_normal((N, d)) / np.sqrt(prec)
X_missing = np.where(rng.random((N, d)) < 0.1, np.nan, X)

Ws = [nodes.Gaussian(d, np.zeros((d, 1)), np.eye(d) * 1e-3) for _ in range(q)]
W = nodes.hstack(Ws)
Mu = nodes.Gaussian(d, np.zeros((d, 1)), np.eye(d) * 1e-3)
Beta = nodes.Gamma(d, 1e-3, 1e-3)
Zs = [nodes.Gaussian(q, np.zeros((q, 1)), np.eye(q)) for _ in range(N)]
for z in Zs:                                         # one shared initial covariance (what the fused kernels hold)
    z.qcov = np.eye(q)
Xs = [nodes.Gaussian(d, W * z + Mu, Beta) for z in Zs]
for x, row in zip(Xs, X_missing):
    x.observe(row.reshape(d, 1))

net = Network()
net.addnode(W)
net.fetch_network(verbose=False)
# tol = -inf: run all iterations.  (network.py:53 stops as soon as the bound improves by less than tol -- also when it
# DEcreases, which the reference's bound, with its reciprocal q_ln_det term, does now and then: SURVEY.md Q1, Q9.)
net.learn(niters, tol=-np.inf, verbose=False)

filled = np.hstack([x.qmu for x in Xs]).T
miss = np.isnan(X_missing)
print("lower bound            :", net.llb)
print("noise precision, learnt:", float(Beta.pass_down_Ex()[0, 0]), " true:", prec)
print("rms error of the imputed entries:", float(np.sqrt(np.mean((filled[miss] - X[miss]) ** 2))),
      " (spread of the data:", float(X.std()), ")")
