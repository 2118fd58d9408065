"""One chain (and a few) at the second shape class of the fused kernels, T = 10^4, D = K = 128: ms per iteration with the time axis
split over W workgroups (the library's choice) and with W = 1 (round 3: a single workgroup)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pyvb_amd import synth
from pyvb_amd.lds import LDSBatch

T, D, K = 10000, 128, 128
for N in (1, 8, 64):
    Y, st0, pri = synth.make_problem(T, D, K, N, seed=5)
    pri["A_prior_prec"] = np.full_like(pri["A_prior_prec"], 1e-2); pri["C_prior_prec"] = np.full_like(pri["C_prior_prec"], 1e-2)
    res = {}
    for force in (None, 1):
        b = LDSBatch.from_problem(Y, st0, pri)
        if force:
            b.set_time_split(force)
        W = b.get_time_split()
        b.iterate(2); b.sync()
        t0 = time.perf_counter(); b.iterate(10); b.sync()
        res[W] = ((time.perf_counter() - t0) / 10 * 1e3, float(b.elbo().sum()))
        b.close()
    print("N=%d T=%d D=K=%d: " % (N, T, D) + "  ".join("W=%d: %.2f ms/iteration (elbo %.9e)" % (w, v[0], v[1]) for w, v in res.items()), flush=True)
