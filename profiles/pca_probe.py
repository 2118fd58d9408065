"""Timing probe of the VB-PCA path at BASELINE configs[4] scale on one GPU: python profiles/pca_probe.py [N d q iters]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pyvb_amd.pca import PCABatch

N, d, q, iters = [int(x) for x in (sys.argv[1:5] + ["1000000", "256", "16", "10"][len(sys.argv) - 1:])]
rng = np.random.default_rng(0)
t0 = time.time()
W = rng.standard_normal((d, q)); Z = rng.standard_normal((N, q)).astype(np.float64)
X = Z @ W.T + rng.standard_normal(d) + 0.2 * rng.standard_normal((N, d))
obs = rng.random((N, d)) > 0.1                      # Bernoulli(0.1) missing
init = {"obs": obs, "X": np.where(obs, X, 0.0), "W_mean": rng.standard_normal((d, q)), "Z": rng.standard_normal((N, q)),
        "Z_cov": np.eye(q), "Mu_mean": np.zeros(d), "beta_b": 1.0}
pri = {"W_prior_mean": np.zeros((d, q)), "W_prior_prec": np.full((q, d), 1e-3), "Mu_prior_mean": np.zeros(d),
       "Mu_prior_prec": np.full(d, 1e-3), "beta_a0": 1e-3, "beta_b0": 1e-3}
print("generated in %.1fs" % (time.time() - t0), flush=True)
t0 = time.time()
b = PCABatch.from_problem(init, pri)
print("uploaded in %.1fs" % (time.time() - t0), flush=True)
b.iterate(2); b.sync()
t0 = time.time()
b.iterate(iters); b.sync()
dt = (time.time() - t0) / iters
e = b.elbo()
gb = N * d * 8 / 1e9
print("N=%d d=%d q=%d: %.3f ms/iteration; X is %.2f GB -> %.0f GB/s per X-pass-equivalent; elbo %.6e" % (N, d, q, dt * 1e3, gb, gb / dt, e.sum()))
