"""One rank of the multi-rank rehearsal on ONE GPU (tests/test_multirank_gpu.py): the library's collectives travel through
the host transport (pyvb_*_comm_init_host over pyvb_amd.dist.SocketComm) because RCCL refuses several ranks on one device.

    python tests/multirank_worker.py {pca|lds} RANK WORLD OUT_PREFIX
"""
import importlib.util
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from pyvb_amd import dist, synth                              # noqa: E402


def _golden():
    spec = importlib.util.spec_from_file_location("make_golden", os.path.join(HERE, "golden", "make_golden.py"))
    G = importlib.util.module_from_spec(spec); spec.loader.exec_module(G)
    return G


PCA_SHAPE = (1003, 40, 5, 21)           # N (not divisible by the world sizes tried), d, q, seed
LDS_SHAPE = (300, 8, 6, 7, 5)           # T, D, K, N, seed
ITERS = 4


def pca_rank(rank, world, comm):
    from pyvb_amd.pca import PCABatch
    N, d, q, seed = PCA_SHAPE
    init, pri = _golden().pca_problem(N, d, q, seed)
    lo, hi = dist.shard_range(N, rank, world)
    b = PCABatch(hi - lo, d, q, device=0, N_total=N, row_offset=lo)
    if world > 1:
        b.comm_init_host(comm, rank, world)
    b.set_priors(pri)
    b.set_data(np.where(init["obs"], init["X"], np.nan)[lo:hi])
    b.set_state(X_missing=init["X"][lo:hi], W_mean=init["W_mean"], Z=init["Z"][lo:hi], Z_cov=init["Z_cov"],
                Mu_mean=init["Mu_mean"], beta_b=float(init["beta_b"]))
    b.iterate(ITERS)
    st = b.get_state()
    st["elbo"] = b.elbo()
    st["rows"] = np.array([lo, hi])
    b.close()
    return st


def lds_rank(rank, world, comm):
    from pyvb_amd.lds import LDSBatch
    T, D, K, N, seed = LDS_SHAPE
    Y, st0, pri = synth.make_problem(T, D, K, N, seed=seed)
    lo, hi = dist.shard_range(N, rank, world)
    sl = slice(lo, hi)
    b = LDSBatch.from_problem(Y[sl], {k: v[sl] for k, v in st0.items()}, pri, device=0)
    if world > 1:
        b.comm_init_host(comm, rank, world)
    b.iterate(ITERS)
    out = {"elbo_total": b.elbo_total(), "elbo_local": b.elbo().sum(0), "history": b.elbo_history(), "rows": np.array([lo, hi]),
           "X": b.get_state(("X",))["X"]}
    b.close()
    return out


if __name__ == "__main__":
    what, rank, world, prefix = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
    comm = dist.SocketComm(world, rank) if world > 1 else dist.LocalComm()
    res = (pca_rank if what == "pca" else lds_rank)(rank, world, comm)
    np.savez(prefix + "_%d.npz" % rank, **res)
    comm.barrier()
    comm.close()
