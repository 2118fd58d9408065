"""One rank of the multi-rank rehearsal on ONE GPU (tests/test_multirank_gpu.py): the library's collectives travel through
the host transport (pyvb_*_comm_init_host over pyvb_amd.dist.SocketComm) because RCCL refuses several ranks on one device.

    python tests/multirank_worker.py {pca|lds|pcabig|pcafuzz} RANK WORLD OUT_PREFIX [N d q | seed]
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


def pca_ops(seed, N):
    """A random sequence of update calls in GLOBAL terms; every rank issues the same calls, row ranges cut to its shard."""
    rng = np.random.default_rng(seed)
    ops = []
    for _ in range(16):
        k = str(rng.choice(["W", "Z", "X", "Xall", "X0", "Mu", "Beta", "elbo"], p=[.15, .15, .2, .1, .1, .12, .12, .06]))
        lo = int(rng.integers(0, N)); hi = int(rng.integers(lo, N + 1))
        ops.append((k, lo, hi))
    return ops


def pca_fuzz_rank(rank, world, comm, seed):
    from pyvb_amd.pca import PCABatch
    N, d, q, _ = PCA_SHAPE
    init, pri = _golden().pca_problem(N, d, q, seed, p_missing=0.2)
    lo, hi = dist.shard_range(N, rank, world)
    b = PCABatch(hi - lo, d, q, device=0, N_total=N, row_offset=lo)
    if world > 1:
        b.comm_init_host(comm, rank, world)
    b.set_priors(pri)
    b.set_data(np.where(init["obs"], init["X"], np.nan)[lo:hi])
    b.set_state(X_missing=init["X"][lo:hi], W_mean=init["W_mean"], Z=init["Z"][lo:hi], Z_cov=init["Z_cov"],
                Mu_mean=init["Mu_mean"], beta_b=float(init["beta_b"]))
    elbos = []
    b.iterate(1)                                # every node has been updated once: the bound is defined from here on
    for k, glo, ghi in pca_ops(seed, N):
        if k == "W": b.update_W()
        elif k == "Z": b.update_Z()
        elif k == "X0": b.update_X0()
        elif k in ("X", "Xall"):
            if k == "Xall":
                glo, ghi = 0, N
            a, e = min(max(glo, lo), hi) - lo, min(max(ghi, lo), hi) - lo
            if (glo, ghi) == (0, 1):
                b.update_X0()
            else:
                b.update_X(a, max(a, e))
        elif k == "Mu": b.update_Mu()
        elif k == "Beta": b.update_Beta()
        else: elbos.append(b.elbo())
    st = b.get_state()
    st["elbo"] = b.elbo()
    st["elbos"] = np.array(elbos).reshape(-1, 5)
    st["rows"] = np.array([lo, hi])
    b.close()
    return st


def big_rows(lo, hi, d, q, seed, block=10000):
    """Rows lo..hi of a seeded N x d problem, generated block by block so that every rank makes only its own rows."""
    g = np.random.default_rng(seed)
    W = g.standard_normal((d, q)); mean = g.standard_normal(d)
    W0 = g.standard_normal((d, q))
    X, obs, Z0 = [], [], []
    for b0 in range(lo // block * block, hi, block):
        r = np.random.default_rng([seed, b0 // block])
        Z = r.standard_normal((block, q))
        x = Z @ W.T + mean + 0.2 * r.standard_normal((block, d))
        o = r.random((block, d)) > 0.1
        z0 = r.standard_normal((block, q))
        a, e = max(lo, b0) - b0, min(hi, b0 + block) - b0
        X.append(x[a:e]); obs.append(o[a:e]); Z0.append(z0[a:e])
    return np.concatenate(X), np.concatenate(obs), np.concatenate(Z0), W0


def pca_big_rank(rank, world, comm, N, d, q):
    """BASELINE configs[4]'s shape, rows sharded over the ranks."""
    from pyvb_amd.pca import PCABatch
    lo, hi = dist.shard_range(N, rank, world)
    X, obs, Z0, W0 = big_rows(lo, hi, d, q, 33)
    pri = {"W_prior_mean": np.zeros((d, q)), "W_prior_prec": np.full((q, d), 1e-3), "Mu_prior_mean": np.zeros(d),
           "Mu_prior_prec": np.full(d, 1e-3), "beta_a0": 1e-3, "beta_b0": 1e-3}
    b = PCABatch(hi - lo, d, q, device=0, N_total=N, row_offset=lo)
    if world > 1:
        b.comm_init_host(comm, rank, world)
    b.set_priors(pri)
    b.set_data(np.where(obs, X, np.nan))
    b.set_state(X_missing=np.where(obs, X, 0.0), W_mean=W0, Z=Z0, Z_cov=np.eye(q), Mu_mean=np.zeros(d), beta_b=1.0)
    del X, obs
    b.iterate(3)
    st = b.get_state()
    out = {k: st[k] for k in ("W_mean", "W_var", "Mu_mean", "Mu_var", "Z_cov", "beta_ab")}
    out["elbo"] = b.elbo()
    out["Z_head"] = st["Z"][:64]; out["X_head"] = st["X"][:64]       # of this rank's shard
    out["Z_sum"] = st["Z"].sum(0); out["X_sum"] = st["X"].sum(0)
    out["rows"] = np.array([lo, hi])
    b.close()
    return out


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
    if what == "pcafuzz":
        res = pca_fuzz_rank(rank, world, comm, int(sys.argv[5]))
    elif what == "pcabig":
        res = pca_big_rank(rank, world, comm, *[int(x) for x in sys.argv[5:8]])
    else:
        res = (pca_rank if what == "pca" else lds_rank)(rank, world, comm)
    np.savez(prefix + "_%d.npz" % rank, **res)
    comm.barrier()
    comm.close()
