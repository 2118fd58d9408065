"""Soak: repeated create/iterate/destroy (device memory must return), long runs twice (bitwise equal)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pyvb_amd import synth
from pyvb_amd.lds import LDSBatch

hip = ctypes.CDLL("libamdhip64.so")
def free_bytes():
    f, t = ctypes.c_size_t(), ctypes.c_size_t()
    assert hip.hipMemGetInfo(ctypes.byref(f), ctypes.byref(t)) == 0
    return f.value

Y, st0, pri = synth.make_problem(700, 24, 20, 37, seed=5)
b = LDSBatch.from_problem(Y, st0, pri); b.iterate(2); b.close()
base = free_bytes()
for i in range(int(os.environ.get("SOAK_CYCLES", "40"))):
    b = LDSBatch.from_problem(Y, st0, pri)
    b.iterate(3)
    e = b.elbo().sum()
    b.close()
print("free memory drift after the create/destroy cycles: %d bytes" % (base - free_bytes()))
runs = []
for r in range(2):
    b = LDSBatch.from_problem(Y, st0, pri)
    tr = []
    for it in range(300):
        b.iterate(1)
        tr.append(b.elbo().sum())
    runs.append((np.array(tr), b.get_state(("X",))["X"]))
    b.close()
assert np.array_equal(runs[0][0], runs[1][0]) and np.array_equal(runs[0][1], runs[1][1]), "runs differ"
tr = runs[0][0]
print("300 iterations twice: bitwise equal; bound %.6e -> %.6e, all finite: %s" % (tr[0], tr[-1], bool(np.all(np.isfinite(tr)))))

# the generic plan: graphs created and dropped (arena, tapes, stream must go with them)
from pyvb_amd import nodes
base = free_bytes()
for i in range(int(os.environ.get("SOAK_GRAPHS", "60"))):
    mu = nodes.Gaussian(3, np.zeros((3, 1)), np.eye(3) * 1e-2)
    prec = nodes.DiagonalGamma(3, np.full(3, 1e-3), np.full(3, 1e-3))
    xs = [nodes.Gaussian(3, mu, prec) for _ in range(20)]
    [x.observe(np.random.randn(3, 1)) for x in xs]
    for it in range(3):
        mu.update(); prec.update()
    v = mu.qmu
    mu._plan.release()
print("generic plan: free memory drift after %s graphs: %d bytes" % (os.environ.get("SOAK_GRAPHS", "60"), base - free_bytes()))

# the VB-PCA handle, with the extra buffers of constructor-drawn rows
import importlib.util
from pyvb_amd.pca import PCABatch
spec = importlib.util.spec_from_file_location("make_golden", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "make_golden.py"))
G = importlib.util.module_from_spec(spec); spec.loader.exec_module(G)
init, ppri = G.pca_problem(3000, 40, 6, seed=3)
init["X_full"] = np.where(init["obs"].all(1)[:, None], init["X"], np.random.default_rng(0).standard_normal(init["X"].shape))
init["X_var0"] = np.where(init["obs"].all(1), 0.0, 1.3)
b = PCABatch.from_problem(init, ppri); b.iterate(2); b.close()
base = free_bytes()
for i in range(int(os.environ.get("SOAK_CYCLES", "40"))):
    b = PCABatch.from_problem(init, ppri)
    b.iterate(3)
    e = b.elbo().sum()
    b.close()
print("VB-PCA: free memory drift after the create/destroy cycles: %d bytes" % (base - free_bytes()))
