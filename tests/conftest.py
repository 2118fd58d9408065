import glob
import os
import sys

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

GOLDEN_DIR = os.path.join(REPO, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def golden_files():
    return sorted(glob.glob(os.path.join(GOLDEN_DIR, "lds_*.npz")))


def load_golden(path):
    """Fixture -> (meta, Y[1,T,K], compact initial state, priors, raw npz dict)."""
    z = dict(np.load(path, allow_pickle=False))
    T, D, K = int(z["T"]), int(z["D"]), int(z["K"])
    kind = str(z["noise"])
    st0 = {k[5:]: z[k][None].copy() for k in z if k.startswith("init_")}
    pri = {k[6:]: z[k].copy() for k in z if k.startswith("prior_")}
    pri["noise"] = kind
    meta = {"T": T, "D": D, "K": K, "noise": kind, "iters": [int(i) for i in z["iters"]],
            "name": os.path.basename(path)[4:-4]}
    return meta, z["Y"][None].copy(), st0, pri, z


@pytest.fixture(params=golden_files(), ids=lambda p: os.path.basename(p)[4:-4])
def golden(request):
    return load_golden(request.param)
