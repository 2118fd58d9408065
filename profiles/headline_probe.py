"""Headline shape only (no CPU baseline, no small problems): a clean target for rocprofv3 --kernel-trace --stats."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pyvb_amd import synth
from pyvb_amd.lds import LDSBatch

N, T, D, K = 1024, 10000, 64, 64
Y, st0, pri = synth.make_problem(T, D, K, 8, seed=1)
Y = np.concatenate([Y] * 128)
st0 = {k: np.concatenate([v] * 128) for k, v in st0.items()}
b = LDSBatch.from_problem(Y, st0, pri)
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 8
b.iterate(2); b.sync(); b.timing(True)
b.iterate(iters); b.sync()
w = b.get_warmup()
print({k: round(v[0] / iters, 3) for k, v in b.kernel_times().items()},
      "warm-up forward", dict(zip(*np.unique(w[:, 0], return_counts=True))), "backward", dict(zip(*np.unique(w[:, 1], return_counts=True))))
