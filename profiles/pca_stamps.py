"""Where a step of the VB-PCA sweep (k_pca_pass12) spends its time: per-stage s_memtime stamps of wavefronts 0 and 4 of every
workgroup, from a library built with -DP12_STAMP (bash profiles/build_pca_variant.sh stamp "-DP12_STAMP").

    PYVB_HIP_LIB=build/variants/libpyvb_hip_stamp.so python profiles/pca_stamps.py [rows d q]
    PYVB_PCA_WRITEBACK=1 PYVB_HIP_LIB=... python profiles/pca_stamps.py        (the sweep that stores the imputed entries)
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
nchunk = min(4096, (N + 15) // 16, max(1, 16384 // ((d + 15) // 16)))
rows = (((N + nchunk - 1) // nchunk) + 15) & ~15
nchunk = (N + rows - 1) // rows
out = np.zeros((nchunk, 2, 12), dtype=np.uint64)
rc = lib.pyvb_pca_debug_stamps(out.ctypes.data_as(ctypes.c_void_p), ctypes.c_int(nchunk))
assert rc == 0
names = ["s0 recompute", "s1 partial Z", "wait A", "s2 sum+store Z", "stage 3 predict/impute", "stage 4 stats+fetch", "wait B", "-", "-", "prologue"]
print("N=%d d=%d q=%d: %.3f ms per iteration (stamped build, %s); %d workgroups of %d rows; s_memtime ticks at the shader clock here (~2.3 GHz): the columns are in HUNDREDS OF CYCLES" %
      (N, d, q, dt * 1e3, "write-back" if os.environ.get("PYVB_PCA_WRITEBACK") == "1" else "lazy", nchunk, rows))
o = out[:nchunk - 1].astype(float)                      # the last chunk is short
steps = o[:, :, 11].mean()
for w, tag in ((0, "wavefront 0 (also sums Z)"), (1, "wavefront 4")):
    print(tag)
    tot = o[:, w, 10].mean()
    acc = 0.0
    for i, nm in enumerate(names):
        if nm == "-":
            continue
        v = o[:, w, i].mean()
        acc += v
        print("   %-26s %8.2f x100 cycles per workgroup   %6.3f x100 cycles per step" % (nm, v / 100.0, v / 100.0 / (steps if i != 9 else 1)))
    print("   %-26s %8.2f x100 cycles per workgroup (stages above: %.2f); %.1f steps -> %.2f x100 cycles per step; workgroup to workgroup: min %.1f max %.1f"
          % ("whole kernel", tot / 100.0, acc / 100.0, steps, tot / 100.0 / steps, o[:, w, 10].min() / 100, o[:, w, 10].max() / 100))
b.close()
