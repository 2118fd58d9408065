"""Where the pair-owning VB-PCA sweep (k_pca_pairs) spends its time, and at what clock: per-phase s_memtime stamps of the first pair
of every workgroup plus the kernel's length on both of the chip's clocks (s_memtime: shader clock; s_memrealtime: 100 MHz), from a
library built with -DP12_STAMP (bash profiles/build_pca_variant.sh stamp "-DP12_STAMP").

    PYVB_PCA_SWEEP=pairs PYVB_HIP_LIB=build/variants/libpyvb_hip_stamp.so python profiles/pca_pair_stamps.py [rows d q]
"""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pyvb_amd import synth, _capi
from pyvb_amd.pca import PCABatch

N, d, q = [int(x) for x in (sys.argv[1:4] + ["1000000", "256", "16"][len(sys.argv) - 1:])]
init, pri = synth.pca_problem(N, d, q, 33)
b = PCABatch.from_problem(init, pri)
b.iterate(3); b.sync()
t0 = time.perf_counter(); b.iterate(10); b.sync(); dt = (time.perf_counter() - t0) / 10
lib = ctypes.CDLL(_capi.LIB_PATH)
nchunk = 256
out = np.zeros((nchunk, 2, 12), dtype=np.uint64)
assert lib.pyvb_pca_debug_stamps(out.ctypes.data_as(ctypes.c_void_p), ctypes.c_int(nchunk)) == 0
o = out.astype(float)
print("N=%d d=%d q=%d: %.3f ms per iteration (stamped build)" % (N, d, q, dt * 1e3))
for w in (0, 1):
    tiles = o[:, w, 11].mean()
    real = o[:, w, 9].mean() / 100.0           # microseconds
    tick = o[:, w, 10].mean()
    print("wavefront %d of a workgroup: kernel %.1f us long (min %.1f max %.1f over workgroups), %.0f shader ticks -> %.2f GHz; %.1f tiles" %
          (w, real, o[:, w, 9].min() / 100, o[:, w, 9].max() / 100, tick, tick / real / 1e3, tiles))
    for i, nm in enumerate(["phase A (recompute, partial Z)", "wait for the partner", "phase B (z)", "phase C (predict, impute, sums)"]):
        print("   %-34s %7.0f ticks per tile  (%4.1f %%)" % (nm, o[:, w, i].mean() / tiles, 100 * o[:, w, i].mean() / tick))
b.close()
