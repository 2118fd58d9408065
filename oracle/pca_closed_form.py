"""ORACLE (test infrastructure, not product code): VB-PCA with missing data.

numpy restatement of what the reference computes for the graph of
examples/PCA_missing_data.py:31-45 --

    W = hstack(q Gaussian columns of dim d, Constant parents)      Mu ~ N(Constant, Constant)
    Beta = Gamma(d, a0, b0)      Z_n ~ N(0, I)      X_n ~ N(W * Z_n + Mu, Beta),  X_n.observe(row with NaNs)

-- when `Network.fetch_network(); Network.learn()` drives it: every iteration updates the
iterable nodes in crawl order  W columns, Z_0..Z_{N-1}, X_0, Mu, X_1..X_{N-1}, Beta  and then sums
log_lower_bound() (network.py:46-49).  Only tests/ and bench code may import this file.

Parity status: PINNED by tests/golden/pca_*.npz (tests/golden/make_golden.py runs the reference).

All Z_n share one posterior covariance (their precision I + <W^T beta W> does not depend on n), the
rows' missing entries are imputed by X_n.update() (gaussian.py:125-134 on a diagonal covariance: known
entries pinned, missing ones <W><z_n> + <Mu> with variance 1/beta) and then message as if observed.  Before
its first update a partially observed row messages with whatever mean it holds at ALL its entries -- the
constructor's random draw unless the caller assigned one (fixture pca_default_init_*: init X_full, X_var0).
Paths are relative to /root/reference/src/pyvb/.

State (float64):  W_mean [d,q], W_var [q,d] (diagonals of the column covariances), Z [N,q], Z_cov [q,q],
X [N,d] (qmu: data where observed, imputed elsewhere), X_var [N,d], Mu_mean [d], Mu_var [d], beta_a, beta_b,
qld_W [q], qld_Z, qld_X (of never-observed rows), qld_Mu.  obs [N,d] bool.
"""
import numpy as np
from scipy.special import digamma, gammaln

LN2PI = np.log(2.0 * np.pi)


def _moments(st):
    """sums over n that the W, Mu and Beta updates read (hstack.pass_up_m1_m2 nodes_todo.py:43-62 through
    Multiplication.pass_up_m1_m2 node.py:193-202 and Addition.pass_up_m1_m2 node.py:95-110)"""
    Z, X = st["Z"], st["X"]
    N = Z.shape[0]
    return {"Szz": Z.T @ Z + N * st["Z_cov"], "Sxz": X.T @ Z, "sx": X.sum(0), "sz": Z.sum(0)}


def update_W(st, pri):
    """[w.update() for w in Ws]: Gaussian.update gaussian.py:102-123; the child chain is
    hstack -> Mult(W, z_n) -> Addition(., Mu) -> X_n, which sends (beta I, beta (x_n - <Mu>))."""
    m = _moments(st)
    beta = st["beta_a"] / st["beta_b"]
    W, q = st["W_mean"], st["W_mean"].shape[1]
    H = m["Sxz"] - np.outer(st["Mu_mean"], m["sz"])                 # sum_n (x_n - mu) z_n^T
    for i in range(q):
        prec = pri["W_prior_prec"][i] + beta * m["Szz"][i, i]
        g = m["Szz"][i].copy(); g[i] = 0.0
        num = pri["W_prior_prec"][i] * pri["W_prior_mean"][:, i] + beta * (H[:, i] - W @ g)
        W[:, i] = num / prec
        st["W_var"][i] = 1.0 / prec
        st["qld_W"][i] = 0.5 / np.sum(0.5 * np.log(prec))           # gaussian.py:120 (quirk Q1)


def _WtW(st):
    """<W^T W> for independent Gaussian columns (node.py:213-227 with an isotropic child precision)"""
    out = st["W_mean"].T @ st["W_mean"]
    out[np.diag_indices_from(out)] += st["W_var"].sum(1)
    return out


def update_Z(st, pri):
    """[z.update() for z in Zs]: parents Constant(0), Constant(I); one child Mult(W, z_n)."""
    beta = st["beta_a"] / st["beta_b"]
    q = st["Z"].shape[1]
    prec = np.eye(q) + beta * _WtW(st)
    st["Z_cov"] = np.linalg.inv(prec)
    st["qld_Z"] = 0.5 / np.sum(np.log(np.diag(np.linalg.cholesky(prec))))
    st["Z"] = (beta * (st["X"] - st["Mu_mean"]) @ st["W_mean"]) @ st["Z_cov"]


def update_X(st, pri, lo, hi):
    """X_n.update() for n in [lo, hi): fully observed rows return (gaussian.py:109-110); the others take
    mean <W><z_n> + <Mu> and covariance I/beta, then the known entries are pinned (:125-134)."""
    beta = st["beta_a"] / st["beta_b"]
    obs = st["obs"][lo:hi]
    pred = st["Z"][lo:hi] @ st["W_mean"].T + st["Mu_mean"]
    upd = ~obs.all(1)
    X, V = st["X"][lo:hi], st["X_var"][lo:hi]
    X[upd] = np.where(obs[upd], st["Xdata"][lo:hi][upd], pred[upd])     # until its first update a row may carry other means there
    V[upd] = np.where(obs[upd], 0.0, 1.0 / beta)
    d = X.shape[1]
    if upd.any():
        st["qld_X"] = 0.5 / (0.5 * d * np.log(beta))


def update_Mu(st, pri):
    """Mu.update(): its N children are the Addition nodes; each sends (beta I, beta (x_n - <W><z_n>))."""
    m = _moments(st)
    beta = st["beta_a"] / st["beta_b"]
    N = st["Z"].shape[0]
    prec = pri["Mu_prior_prec"] + N * beta
    num = pri["Mu_prior_prec"] * pri["Mu_prior_mean"] + beta * (m["sx"] - st["W_mean"] @ m["sz"])
    st["Mu_mean"] = num / prec
    st["Mu_var"] = 1.0 / prec
    st["qld_Mu"] = 0.5 / np.sum(0.5 * np.log(prec))


def _residual(st):
    """sum_n tr[ <x x^T> + <m m^T> - 2 <x><m>^T ],  m = W z_n + Mu  (Addition.pass_down_ExxT node.py:121-129,
    Multiplication.pass_down_ExxT hstack branch node.py:260-271)"""
    m = _moments(st)
    W, mu = st["W_mean"], st["Mu_mean"]
    N = st["Z"].shape[0]
    own = np.sum(st["X"] ** 2) + np.sum(st["X_var"])
    WzWz = np.sum((W.T @ W) * m["Szz"]) + np.sum(st["W_var"].sum(1) * np.diag(m["Szz"]))
    mm = WzWz + N * (mu @ mu + st["Mu_var"].sum()) + 2.0 * (W @ m["sz"]) @ mu
    cross = np.sum(m["Sxz"] * W) + m["sx"] @ mu
    return own + mm - 2.0 * cross


def update_Beta(st, pri):
    """Beta.update(): Gamma nodes_todo.py:130-138 (traces); qa = a0 + d N / 2 (:125-128)."""
    N, d = st["X"].shape
    st["beta_b"] = pri["beta_b0"] + 0.5 * _residual(st)
    st["beta_a"] = pri["beta_a0"] + 0.5 * d * N


def elbo_parts(st, pri):
    """[L_W, L_Z, L_X, L_Mu, L_Beta]: sums of log_lower_bound() per node class
    (gaussian.py:136-151; nodes_todo.py:149-157)."""
    N, d = st["X"].shape
    q = st["Z"].shape[1]
    a, b = st["beta_a"], st["beta_b"]
    beta = a / b
    lnd_beta = d * (np.log(a) - np.log(b))                            # Gamma.pass_down_lndet (quirk Q2)
    # X_n
    LX = N * (-0.5 * d * LN2PI + 0.5 * lnd_beta) - 0.5 * beta * _residual(st)
    nmiss = (~st["obs"]).sum(1)
    part = (nmiss > 0) & (nmiss < d)
    none = nmiss == d
    Vm = np.where(st["obs"], 1.0, st["X_var"])
    LX -= np.sum(0.5 * nmiss[part] * LN2PI - 0.5 * np.log(Vm[part]).sum(1) - 0.5 * nmiss[part])    # gaussian.py:148-150
    if none.any():      # :145-147; each latent row keeps the q_ln_det of ITS last update: qprec = <beta> I then, qcov = I / <beta>
        with np.errstate(divide="ignore"):
            qld_rows = 0.5 / (0.5 * d * np.log(1.0 / st["X_var"][none, 0]))
        LX += none.sum() * (0.5 * d * LN2PI + 0.5 * d) + 0.5 * qld_rows.sum()
    # Z_n against Constant(0), Constant(I)
    m = _moments(st)
    LZ = N * (-0.5 * q * LN2PI) - 0.5 * np.trace(m["Szz"]) + N * (0.5 * q * LN2PI + 0.5 * st["qld_Z"] + 0.5 * q)
    # W columns and Mu against Constant parents
    W, Wv = st["W_mean"], st["W_var"]
    LW = 0.0
    for i in range(q):
        pp, pm = pri["W_prior_prec"][i], pri["W_prior_mean"][:, i]
        LW += -0.5 * d * LN2PI + 0.5 * np.sum(np.log(pp)) - 0.5 * np.sum(pp * (W[:, i] ** 2 + Wv[i] + pm ** 2 - 2 * W[:, i] * pm))
        LW += 0.5 * d * LN2PI + 0.5 * st["qld_W"][i] + 0.5 * d
    pp, pm = pri["Mu_prior_prec"], pri["Mu_prior_mean"]
    LM = -0.5 * d * LN2PI + 0.5 * np.sum(np.log(pp)) - 0.5 * np.sum(pp * (st["Mu_mean"] ** 2 + st["Mu_var"] + pm ** 2 - 2 * st["Mu_mean"] * pm))
    LM += 0.5 * d * LN2PI + 0.5 * st["qld_Mu"] + 0.5 * d
    # Beta
    a0, b0 = pri["beta_a0"], pri["beta_b0"]
    Elnx = digamma(a) - np.log(b)
    LB = (a0 - 1) * Elnx - gammaln(a0) + a0 * np.log(b0) - b0 * beta
    LB -= (a - 1) * Elnx - gammaln(a) + a * np.log(b) - b * beta
    return np.array([LW, LZ, LX, LM, LB])


def iterate(st, pri):
    """One pass of Network.learn over the fetched PCA network, in the reference's crawl order."""
    N = st["X"].shape[0]
    update_W(st, pri)
    update_Z(st, pri)
    update_X(st, pri, 0, 1)
    update_Mu(st, pri)
    update_X(st, pri, 1, N)
    update_Beta(st, pri)
    return elbo_parts(st, pri)


def make_state(init, pri, N, d, q):
    """Copy an explicit initial state (tests/golden/make_golden.py: pca_initial_state) into the layout above."""
    st = {k: np.array(v, dtype=float, copy=True) for k, v in init.items() if k not in ("obs", "X_full", "X_var0")}
    st["obs"] = np.array(init["obs"], dtype=bool)
    st["X_var"] = np.zeros((N, d))
    st["Xdata"] = st["X"].copy()            # the observations (where obs)
    if "X_full" in init:
        # the X_n as their constructors drew them (gaussian.py:70-72): a row that is not fully observed carries a mean at
        # ALL its entries and the covariance c_n I until its first update conditions it on the observed ones (:90-96, :125-134)
        free = ~st["obs"].all(1)
        st["X"][free] = np.asarray(init["X_full"], dtype=float)[free]
        st["X_var"][free] = np.asarray(init["X_var0"], dtype=float)[free, None]
    st["W_var"] = np.zeros((q, d))
    st["Mu_var"] = np.zeros(d)
    st["qld_W"] = np.full(q, np.nan)
    st["qld_Z"] = st["qld_X"] = st["qld_Mu"] = np.nan
    st["beta_a"] = pri["beta_a0"] + 0.5 * d * N
    st["beta_b"] = float(init["beta_b"])
    return st
