import sys, time
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pyvb_amd import synth
from pyvb_amd.lds import LDSBatch
for (T, D, K, N) in [(10000, 16, 16, 1), (10000, 64, 64, 1), (200, 4, 5, 1)]:
    Y, st0, pri = synth.make_problem(T, D, K, N, seed=3)
    b = LDSBatch.from_problem(Y, st0, pri)
    b.iterate(3); b.sync(); b.timing(True)
    t0 = time.perf_counter(); b.iterate(20); b.sync(); dt = (time.perf_counter() - t0) / 20
    print("T=%d D=%d K=%d N=%d: %.3f ms/iteration" % (T, D, K, N, dt * 1e3), {k: round(v[0] / 20, 3) for k, v in b.kernel_times().items() if v[1]})
    b.close()
