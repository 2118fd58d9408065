"""What limits the sweeps: matrix-pipe issue, HBM, or the chip's power/clock management?

    python3 profiles/exp_limits.py N [smi]

Times the forward / backward sweep launches and the statistics kernel at the headline shape (T = 10^4,
D = K = 64) for N replicates (N wavefronts on the chip's 1024 SIMDs).  With `smi`, samples
`rocm-smi --showclocks --showpower` while a long run of sweeps is in flight.  Run it once per library
build (PYVB_HIP_LIB) and per N; see profiles/r02/limits.txt for the table.
"""
import os
import subprocess
import sys
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pyvb_amd import synth
from pyvb_amd.lds import LDSBatch

N = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
smi = "smi" in sys.argv[2:]
zero = "zero" in sys.argv[2:]        # all-zero data and states: the same instruction stream with (almost) no operand toggling
T, D, K = 10000, 64, 64
base = 8
Y, st0, pri = synth.make_problem(T, D, K, base, seed=1)
rep = (N + base - 1) // base
Y = np.concatenate([Y] * rep)[:N]
st0 = {k: np.concatenate([v] * rep)[:N] for k, v in st0.items()}
b = LDSBatch.from_problem(Y, st0, pri)
b.set_time_split(1)
b.iterate(2)
b.sync()
if zero:
    b.set_observations(np.zeros_like(Y))
    b.set_state(X=np.zeros((N, T, D)))


def pairs(n):
    for _ in range(n):
        b.sweep("forward")
        b.sweep("backward")


pairs(2)
b.sync()
b.timing(True)
pairs(10)
b.sync()
kt = b.kernel_times()
print("N=%d%s lib=%s  fwd %.3f ms  bwd(MODE 2) %.3f ms" % (
    N, " ZERO DATA" if zero else "", os.path.basename(os.environ.get("PYVB_HIP_LIB", "default")), kt["sweep_fwd"][0] / 10, kt["sweep_bwd"][0] / 10), flush=True)
b.timing(False)

if smi:
    samples = []
    stop = threading.Event()

    def sampler():
        while not stop.is_set():
            try:
                out = subprocess.run(["rocm-smi", "--showclocks", "--showpower", "--csv"], capture_output=True, text=True, timeout=10).stdout
                samples.append((time.time(), out))
            except Exception as e:
                samples.append((time.time(), "ERR %s" % e))
            time.sleep(0.05)

    th = threading.Thread(target=sampler)
    th.start()
    t0 = time.time()
    for _ in range(40):
        pairs(10)
        b.sync()
    t1 = time.time()
    time.sleep(1.0)      # a second of idle at the end for contrast
    stop.set()
    th.join()
    print("busy window %.2f s (%.3f ms per forward+backward pair)" % (t1 - t0, (t1 - t0) / 400 * 1e3))
    for ts, out in samples:
        tag = "busy" if t0 <= ts <= t1 else "idle"
        lines = [l for l in out.strip().split("\n") if l]
        print(tag, "%.2f" % (ts - t0), " | ".join(lines[-2:]) if len(lines) >= 2 else out.strip())
b.close()
