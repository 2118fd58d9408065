"""Time k_stats alone at the headline shape (backward sweep + statistics, four times)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pyvb_amd import synth
from pyvb_amd.lds import LDSBatch
N, T, D, K = 1024, 10000, 64, 64
Y, st0, pri = synth.make_problem(T, D, K, 8, seed=1)
Y = np.concatenate([Y] * 128); st0 = {k: np.concatenate([v] * 128) for k, v in st0.items()}
b = LDSBatch.from_problem(Y, st0, pri)
b.sweep("forward"); b.sweep("backward"); b.sync(); b.timing(True)
for i in range(4):
    b.sweep("backward")
    try:
        b.update_A(); b.sync()
    except Exception as e:
        pass
kt = b.kernel_times()
print("stats %.3f ms" % (kt["stats"][0] / max(kt["stats"][1], 1)))
