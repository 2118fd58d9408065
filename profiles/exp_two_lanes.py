"""Would two half-batches on two streams hide the small kernels (k_prep, k_cols, k_elbo: low power, ~1 ms of 11.5) behind
the other half's sweeps?  One LDSBatch of 1024 replicates against two of 512 whose iterations are enqueued alternately
(each handle has its own stream; nothing synchronises inside the timed loop).  See profiles/r02/limits.txt."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pyvb_amd import synth
from pyvb_amd.lds import LDSBatch

T, D, K = 10000, 64, 64
Y, st0, pri = synth.make_problem(T, D, K, 8, seed=1)


def batch(N):
    rep = N // 8
    b = LDSBatch.from_problem(np.concatenate([Y] * rep), {k: np.concatenate([v] * rep) for k, v in st0.items()}, pri)
    b.set_time_split(1)
    return b


def timed(bs, iters, stagger):
    for b in bs:
        b.iterate(2)
    for b in bs:
        b.sync()
    t0 = time.perf_counter()
    if stagger and len(bs) == 2:
        bs[0].sweep("forward"); bs[0].sweep("backward")     # puts lane 0 half an iteration ahead (extra work, not counted in its favour)
    for _ in range(iters):
        for b in bs:
            b.iterate(1)
    for b in bs:
        b.sync()
    return (time.perf_counter() - t0) / iters * 1e3


one = batch(1024)
print("one handle, 1024 replicates:            %.3f ms per iteration" % timed([one], 20, False), flush=True)
one.close()
two = [batch(512), batch(512)]
print("two handles of 512, alternating:        %.3f ms per iteration of all 1024" % timed(two, 20, False), flush=True)
print("two handles of 512, staggered start:    %.3f ms per iteration of all 1024" % timed(two, 20, True), flush=True)
for b in two:
    b.close()
four = [batch(256) for _ in range(4)]
print("four handles of 256, alternating:       %.3f ms per iteration of all 1024" % timed(four, 20, False), flush=True)
