"""Wishart noise at the headline shape (T = 10^4, D = K = 64, N replicates): time per iteration and per kernel class."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pyvb_amd import synth
from pyvb_amd.lds import LDSBatch

N = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
T, D, K = 10000, 64, 64
Y, st0, pri = synth.make_problem(T, D, K, 8, seed=1)
Y = np.concatenate([Y] * (N // 8)); st0 = {k: np.concatenate([v] * (N // 8)) for k, v in st0.items()}
pri["noise"] = "wishart"
pri["Q_a0"], pri["Q_b0"] = np.float64(0.5 * D + 1.0), np.eye(D) * 0.05
pri["R_a0"], pri["R_b0"] = np.float64(0.5 * K + 1.0), np.eye(K) * 0.05
b = LDSBatch.from_problem(Y, st0, pri)
b.iterate(2); b.sync(); b.timing(True)
t0 = time.perf_counter()
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 10
b.iterate(iters); b.sync()
dt = (time.perf_counter() - t0) / iters
print("Wishart noise, N=%d T=%d D=%d K=%d: %.3f ms per iteration" % (N, T, D, K, dt * 1e3),
      {k: round(v[0] / iters, 3) for k, v in b.kernel_times().items() if v[1]}, "elbo finite:", bool(np.isfinite(b.elbo()).all()))
