import sys, os, time
sys.path.insert(0, ".")
import numpy as np
from pyvb_amd import synth
from pyvb_amd.lds import LDSBatch
N,T,D,K=1024,10000,64,64
Y, st0, pri = synth.make_problem(T, D, K, 8, seed=1)
Y=np.concatenate([Y]*128); st0={k:np.concatenate([v]*128) for k,v in st0.items()}
b = LDSBatch.from_problem(Y, st0, pri)
b.sweep("forward"); b.sweep("backward"); b.sync(); b.timing(True)
for _ in range(3):
    b.sweep("forward"); b.sync(); f=b.kernel_times()["sweep"][0]
    b.sweep("backward"); b.sync(); t=b.kernel_times()["sweep"][0]
    print("fwd %.3f ms  bwd %.3f ms (cumulative %.3f)" % (f - getattr(b,'_last',0.0), t - f, t)); b._last = t
