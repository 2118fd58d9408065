import sys, os, time
sys.path.insert(0, ".")
import numpy as np
from pyvb_amd import synth
from pyvb_amd.lds import LDSBatch
N,T,D,K=1024,64,64,64
Y, st0, pri = synth.make_problem(T, D, K, 16, seed=1)
Y=np.concatenate([Y]*64); st0={k:np.concatenate([v]*64) for k,v in st0.items()}
b = LDSBatch.from_problem(Y, st0, pri)
b.iterate(2); b.sync(); b.timing(True)
b.iterate(5); b.sync()
print(os.environ.get("PYVB_PREP_SKIP","0"), {k:(round(v[0]/max(v[1],1),3)) for k,v in b.kernel_times().items()})
