"""What a record of the tape interpreter costs, by kind, on operands of node size: python profiles/tape_record_cost.py
One workgroup runs a tape of R copies of one record (R = 4000); the time per record is (run time - empty launch) / R."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pyvb_amd import generic as G

R = 4000
ex = G.DeviceExecutor(1 << 16)
ex.write(0, np.linspace(1.0, 2.0, 4096))


def timed(ops, reps=5):
    t = ex.tape(np.asarray(ops, dtype=np.int32))
    ex.run(t); ex.sync()
    t0 = time.perf_counter()
    for _ in range(reps):
        ex.run(t)
    ex.sync()
    dt = (time.perf_counter() - t0) / reps
    ex.drop(t)
    return dt


base = timed([[G.T_FILL, 100, 0, 2, 2, 2, 0, 0]] * 3)
cases = {
    "fill 2x2": [G.T_FILL, 100, 0, 2, 2, 2, 0, 0],
    "copy2d 2x2": [G.T_COPY2D, 100, 0, 2, 2, 2, 2, 0],
    "axpby 2x1": [G.T_AXPBY, 100, 0, 8, 2, 1, 16, 17],
    "gemm 2x2x2": [G.T_GEMM, 100, 0, 8, 2, 2, 2, 0],
    "gemm 5x5x5": [G.T_GEMM, 100, 0, 32, 5, 5, 5, 0],
    "trace 5x5": [G.T_TRACE, 100, 0, 0, 5, 5, 0, 0],
    "mul 5x1": [G.T_MUL, 100, 0, 8, 5, 1, 0, 0],
    "unary recip 5": [G.T_UNARY, 100, 0, 0, 5, 1, 0, 3],
    "cholinv 2x2": None,
}
ex.write(200, np.array([2.0, 0.3, 0.3, 1.5]))
cases["cholinv 2x2"] = [G.T_CHOLINV, 100, 200, 110, 2, 2, 120, 0]
print("empty launch (3 records): %.1f us" % (base * 1e6))
for name, rec in cases.items():
    dt = timed([rec] * R)
    print("%-16s %7.1f ns per record" % (name, (dt - base) / R * 1e9))
# the same records far apart in the arena (a window of its own each time is not formed: all in one window here), and a long tape
# whose working set does not fit one window: 4000 gemms over 40000 doubles
ops = [[G.T_GEMM, 20000 + 8 * (i % 4000), 8 * (i % 4000), 8, 2, 2, 2, 0] for i in range(R)]
print("%-16s %7.1f ns per record" % ("gemm 2x2 spread", (timed(ops) - base) / R * 1e9))
