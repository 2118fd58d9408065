"""Synthetic linear-dynamical-system data and initial variational state.

Follows the simulator of the reference's LDS example
(examples/Linear_Dynamic_System.py:20-44) with the two changes SURVEY.md §8(d)
states: seeded generators instead of the global numpy stream, and a rescaled
transition matrix instead of the rejection loop at :20-23 (which does not
terminate for latent dimension >~6).

Everything here is host-side numpy; it only produces inputs.
"""
import numpy as np

__all__ = ["simulate_lds", "initial_state", "default_priors", "make_problem"]


def simulate_lds(T, D, K, N=1, seed=20240):
    """Draw N independent LDS data sets.

    x_t = A x_{t-1} + Q^{1/2} eps,   y_t = C x_t + R^{1/2} eta
    (Linear_Dynamic_System.py:40-44), A rescaled to spectral radius 0.9,
    C = 10*randn (:25), Q, R = 0.1*diag(U(0,1)) (:29, :34), x_0 ~ N(0, I) (:40).

    Returns dict with Y [N,T,K] and the true parameters.
    """
    rng = np.random.default_rng(seed)
    A = rng.standard_normal((N, D, D))
    rad = np.max(np.abs(np.linalg.eigvals(A)), axis=-1)
    A *= (0.9 / rad)[:, None, None]
    C = 10.0 * rng.standard_normal((N, K, D))
    Rd = 0.1 * rng.random((N, K))
    Qd = 0.1 * rng.random((N, D))
    sq, sr = np.sqrt(Qd), np.sqrt(Rd)
    X = np.empty((N, T, D))
    Y = np.empty((N, T, K))
    x = rng.standard_normal((N, D))
    for t in range(T):
        if t > 0:
            x = np.einsum("nij,nj->ni", A, x) + sq * rng.standard_normal((N, D))
        X[:, t] = x
        Y[:, t] = np.einsum("nkj,nj->nk", C, x) + sr * rng.standard_normal((N, K))
    return {"Y": Y, "X": X, "A": A, "C": C, "Q": Qd, "R": Rd}


def initial_state(T, D, K, N=1, seed=1):
    """Initial variational state drawn the way the reference's constructors do
    (gaussian.py:70-72: qmu ~ N(0,1), qprec = I*u; nodes_todo.py:177: qb = u),
    but from an explicit generator.

    Column covariances are isotropic (I/u), so only their diagonals are kept:
    A_colvar[n, i, k] is the variance of entry k of column i of A.
    """
    rng = np.random.default_rng(seed)
    st = {
        "X": rng.standard_normal((N, T, D)),
        "A_mean": rng.standard_normal((N, D, D)),       # [row, col]
        "C_mean": rng.standard_normal((N, K, D)),       # [row, col]
        "A_colvar": np.repeat(1.0 / rng.random((N, D, 1)), D, axis=2),  # [col, entry]
        "C_colvar": np.repeat(1.0 / rng.random((N, D, 1)), K, axis=2),  # [col, entry]
        "Q_b": np.repeat(rng.random((N, 1)), D, axis=1),
        "R_b": np.repeat(rng.random((N, 1)), K, axis=1),
    }
    return st


def default_priors(D, K):
    """Priors of the example (Linear_Dynamic_System.py:47-58)."""
    return {
        "x0_mean": np.zeros(D),
        "x0_prec": np.eye(D),
        "A_prior_mean": np.zeros((D, D)),        # [row, col]
        "A_prior_prec": np.full((D, D), 1e-3),   # [col, entry] diagonal of each column's prior precision
        "C_prior_mean": np.zeros((K, D)),
        "C_prior_prec": np.full((D, K), 1e-3),
        "Q_a0": np.full(D, 1e-3), "Q_b0": np.full(D, 1e-3),
        "R_a0": np.full(K, 1e-3), "R_b0": np.full(K, 1e-3),
        "noise": "diagonal_gamma",
    }


def make_problem(T, D, K, N=1, seed=20240):
    """Data + initial state + priors for N replicates (one generator per call)."""
    sim = simulate_lds(T, D, K, N, seed)
    st = initial_state(T, D, K, N, seed + 7919)
    return sim["Y"], st, default_priors(D, K)


def pca_rows(lo, hi, d, q, seed, block=10000):
    """Rows [lo, hi) of the synthetic VB-PCA problem (BASELINE configs[4]: low-rank rows + noise, Bernoulli(0.1) missing
    entries), generated block by block from (seed, block index) so that any row range of any total size is reproducible --
    ranks generate only their own shard.  Returns X, the observation mask, initial latent means, the initial W mean."""
    g = np.random.default_rng(seed)
    W = g.standard_normal((d, q)); mean = g.standard_normal(d); W0 = g.standard_normal((d, q))
    X, obs, Z0 = [], [], []
    for b0 in range(lo // block * block, hi, block):
        r = np.random.default_rng([seed, b0 // block])
        x = r.standard_normal((block, q)) @ W.T + mean + 0.2 * r.standard_normal((block, d))
        o = r.random((block, d)) > 0.1
        z0 = r.standard_normal((block, q))
        a, e = max(lo, b0) - b0, min(hi, b0 + block) - b0
        X.append(x[a:e]); obs.append(o[a:e]); Z0.append(z0[a:e])
    return np.concatenate(X), np.concatenate(obs), np.concatenate(Z0), W0


def pca_problem(N, d, q, seed):
    """(init, pri) for pyvb_amd.pca.PCABatch.from_problem / oracle.pca_closed_form.make_state."""
    X, obs, Z0, W0 = pca_rows(0, N, d, q, seed)
    init = {"obs": obs, "X": np.where(obs, X, 0.0), "W_mean": W0, "Z": Z0, "Z_cov": np.eye(q), "Mu_mean": np.zeros(d), "beta_b": 1.0}
    pri = {"W_prior_mean": np.zeros((d, q)), "W_prior_prec": np.full((q, d), 1e-3), "Mu_prior_mean": np.zeros(d),
           "Mu_prior_prec": np.full(d, 1e-3), "beta_a0": 1e-3, "beta_b0": 1e-3}
    return init, pri
