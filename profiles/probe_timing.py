"""Quick per-kernel timing probe (HIP events inside the library): python profiles/probe_timing.py N T D K iters"""
import sys, time
import numpy as np
sys.path.insert(0, ".")
from pyvb_amd import synth
from pyvb_amd.lds import LDSBatch

N, T, D, K, iters = [int(x) for x in (sys.argv[1:6] + ["64", "10000", "64", "64", "3"][len(sys.argv) - 1:])]
t0 = time.time()
base = min(N, 16)
Y, st0, pri = synth.make_problem(T, D, K, base, seed=1)
rep = (N + base - 1) // base
Y = np.concatenate([Y] * rep)[:N]
st0 = {k: np.concatenate([v] * rep)[:N] for k, v in st0.items()}
print("generated in %.1fs" % (time.time() - t0), flush=True)
b = LDSBatch.from_problem(Y, st0, pri)
b.iterate(1); b.sync()
b.timing(True)
t0 = time.time()
b.iterate(iters); b.sync()
wall = time.time() - t0
kt = b.kernel_times()
print("N=%d T=%d D=%d K=%d: %.2f ms/iteration (timed mode, serialised)" % (N, T, D, K, wall / iters * 1e3))
for k, (ms, cnt) in kt.items():
    if cnt:
        print("  %-7s %8.3f ms total  %4d launches  %8.3f ms/launch" % (k, ms, cnt, ms / cnt))
print("warm-up lengths (first replicates):", b.get_warmup()[:4].tolist())
b.timing(False)
t0 = time.time()
b.iterate(iters); b.sync()
print("untimed: %.2f ms/iteration" % ((time.time() - t0) / iters * 1e3))
print("elbo total", b.elbo_total().sum())
