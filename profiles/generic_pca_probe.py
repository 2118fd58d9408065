import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
from pyvb_amd import nodes, generic, _recognise, network
_recognise.bind = lambda node: generic.GenericPlan(node)
N, d, q = 2000, 10, 3
rng = np.random.default_rng(1)
X = rng.standard_normal((N, q)) @ rng.standard_normal((q, d)) + 0.1 * rng.standard_normal((N, d))
X[rng.random((N, d)) < 0.1] = np.nan
Ws = [nodes.Gaussian(d, np.zeros((d, 1)), np.eye(d) * 1e-3) for _ in range(q)]
W = nodes.hstack(Ws)
Mu = nodes.Gaussian(d, np.zeros((d, 1)), np.eye(d) * 1e-3)
Beta = nodes.Gamma(d, 1e-3, 1e-3)
Zs = [nodes.Gaussian(q, np.zeros((q, 1)), np.eye(q)) for _ in range(N)]
Xs = [nodes.Gaussian(d, W * z + Mu, Beta) for z in Zs]
for x, row in zip(Xs, X):
    x.observe(row.reshape(d, 1))
def it():
    [w.update() for w in Ws]; [z.update() for z in Zs]; [x.update() for x in Xs]; Mu.update(); Beta.update()
it(); _ = Mu.qmu
it(); _ = Mu.qmu
t0 = time.time()
for _i in range(5):
    it(); _ = Mu.qmu
print("per pass %.2f ms" % ((time.time() - t0) / 5 * 1e3))
