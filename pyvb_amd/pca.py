"""PCABatch: the VB-PCA-with-missing-data graph of the reference's examples/PCA_missing_data.py:31-42
(N rows of one model) resident on one MI355X; thin Python over the C ABI (include/pyvb_hip.h)."""
import numpy as np

from . import _capi as C

__all__ = ["PCABatch"]


def _f64(a, shape, name):
    a = np.ascontiguousarray(a, dtype=np.float64)
    if a.shape != tuple(shape):
        raise AssertionError("%s has shape %s, expected %s" % (name, a.shape, tuple(shape)))
    return a


class PCABatch(object):
    ELBO_PARTS = ("W", "Z", "X", "Mu", "Beta")

    def __init__(self, N, d, q, device=0, N_total=None, row_offset=0):
        self.N, self.d, self.q = int(N), int(d), int(q)
        h = C.ctypes.c_void_p()
        C.check(C.lib.pyvb_pca_create(C.ctypes.byref(h), int(device), self.N, self.d, self.q,
                                      int(self.N if N_total is None else N_total), int(row_offset)))
        self._h = h

    def close(self):
        if getattr(self, "_h", None):
            C.lib.pyvb_pca_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def comm_init(self, uid, rank, world):
        """Attach the RCCL communicator (before set_data, so that the row counts become global)."""
        C.check(C.lib.pyvb_pca_comm_init(self._h, uid, int(rank), int(world)))

    def comm_init_host(self, comm, rank, world):
        """The same collectives through a host process group (pyvb_amd.dist: SocketComm / GlooComm) instead of RCCL:
        rehearsal of the sharded path with several ranks on one GPU (pyvb_pca_comm_init_host)."""
        from .dist import host_allreduce_callback
        self._host_cb = host_allreduce_callback(comm)          # kept alive with the handle
        C.check(C.lib.pyvb_pca_comm_init_host(self._h, self._host_cb, None, int(rank), int(world)))

    def set_priors(self, pri):
        d, q = self.d, self.q
        a = [_f64(pri["W_prior_mean"], (d, q), "W_prior_mean"), _f64(pri["W_prior_prec"], (q, d), "W_prior_prec"),
             _f64(pri["Mu_prior_mean"], (d,), "Mu_prior_mean"), _f64(pri["Mu_prior_prec"], (d,), "Mu_prior_prec")]
        C.check(C.lib.pyvb_pca_set_priors(self._h, *[C.dptr(x) for x in a], float(pri["beta_a0"]), float(pri["beta_b0"])))

    def set_data(self, X):
        """X [N, d] with NaN where an entry is missing (Gaussian.observe, gaussian.py:74-100)."""
        C.check(C.lib.pyvb_pca_set_data(self._h, C.dptr(_f64(X, (self.N, self.d), "X"))))

    def set_state(self, X_missing=None, W_mean=None, Z=None, Z_cov=None, Mu_mean=None, beta_b=None):
        N, d, q = self.N, self.d, self.q
        arrs = [None if X_missing is None else _f64(X_missing, (N, d), "X_missing"),
                None if W_mean is None else _f64(W_mean, (d, q), "W_mean"),
                None if Z is None else _f64(Z, (N, q), "Z"),
                None if Z_cov is None else _f64(Z_cov, (q, q), "Z_cov"),
                None if Mu_mean is None else _f64(Mu_mean, (d,), "Mu_mean"),
                None if beta_b is None else np.array([float(beta_b)])]
        C.check(C.lib.pyvb_pca_set_state(self._h, *[C.dptr(x) for x in arrs]))

    def set_unpinned_rows(self, X_full, row_var):
        """Rows that are not fully observed start as Gaussian.__init__ leaves them: mean X_full[n] at ALL entries, covariance
        row_var[n] * I, until their first update conditions them on the observed entries (pyvb_pca_set_unpinned_rows)."""
        C.check(C.lib.pyvb_pca_set_unpinned_rows(self._h, C.dptr(_f64(X_full, (self.N, self.d), "X_full")),
                                                 C.dptr(_f64(row_var, (self.N,), "row_var"))))

    def set_initial_variances(self, W_var=None, Mu_var=None):
        """Diagonals of the initial covariances of the W columns [q, d] and of Mu [d] (pyvb_pca_set_initial_variances)."""
        a = None if W_var is None else _f64(W_var, (self.q, self.d), "W_var")
        b = None if Mu_var is None else _f64(Mu_var, (self.d,), "Mu_var")
        C.check(C.lib.pyvb_pca_set_initial_variances(self._h, C.dptr(a), C.dptr(b)))

    def get_state(self):
        N, d, q = self.N, self.d, self.q
        out = {"X": np.empty((N, d)), "X_rowvar": np.empty(N), "W_mean": np.empty((d, q)), "W_var": np.empty((q, d)),
               "Z": np.empty((N, q)), "Z_cov": np.empty((q, q)), "Mu_mean": np.empty(d), "Mu_var": np.empty(d), "beta_ab": np.empty(2)}
        order = ["X", "X_rowvar", "W_mean", "W_var", "Z", "Z_cov", "Mu_mean", "Mu_var", "beta_ab"]
        C.check(C.lib.pyvb_pca_get_state(self._h, *[C.dptr(out[k]) for k in order]))
        out["beta_a"], out["beta_b"] = out["beta_ab"]
        return out

    def get_qld(self, rows=True):
        """q_ln_det values as the updates on this handle left them (NaN: not updated here yet): W columns [q], the Z_n's
        shared one, Mu's, and per row for the X_n without any observed entry (NaN for the others)."""
        w, z, m = np.empty(self.q), np.empty(1), np.empty(1)
        x = np.empty(self.N) if rows else None
        C.check(C.lib.pyvb_pca_get_qld(self._h, C.dptr(w), C.dptr(z), C.dptr(m), C.dptr(x)))
        return {"W": w, "Z": float(z[0]), "Mu": float(m[0]), "X": x}

    def update_W(self):
        C.check(C.lib.pyvb_pca_update_W(self._h))

    def update_Z(self):
        C.check(C.lib.pyvb_pca_update_Z(self._h))

    def update_X(self, lo=0, hi=None):
        C.check(C.lib.pyvb_pca_update_X(self._h, int(lo), int(self.N if hi is None else hi)))

    def update_X0(self):
        """Xs[0].update() of the global row 0; with a communicator the collective single-row step of every rank."""
        C.check(C.lib.pyvb_pca_update_X0(self._h))

    def update_Mu(self):
        C.check(C.lib.pyvb_pca_update_Mu(self._h))

    def update_Beta(self):
        C.check(C.lib.pyvb_pca_update_Beta(self._h))

    def elbo(self):
        out = np.empty(5)
        C.check(C.lib.pyvb_pca_elbo(self._h, C.dptr(out)))
        return out

    def iterate(self, niters=1):
        C.check(C.lib.pyvb_pca_iterate(self._h, int(niters)))

    def sync(self):
        C.check(C.lib.pyvb_pca_sync(self._h))

    @classmethod
    def from_problem(cls, init, pri, device=0):
        """init: dict with obs [N,d] bool, X [N,d] (data / initial means of the missing entries), W_mean, Z, Z_cov,
        Mu_mean, beta_b (tests/golden/make_golden.py: pca_problem)."""
        N, d = init["X"].shape
        q = init["Z"].shape[1]
        b = cls(N, d, q, device)
        b.set_priors(pri)
        b.set_data(np.where(init["obs"], init["X"], np.nan))
        b.set_state(X_missing=init["X"], W_mean=init["W_mean"], Z=init["Z"], Z_cov=init["Z_cov"],
                    Mu_mean=init["Mu_mean"], beta_b=float(init["beta_b"]))
        if "W_var" in init or "Mu_var" in init:
            b.set_initial_variances(init.get("W_var"), init.get("Mu_var"))
        if "X_full" in init:                # the X_n as their constructors drew them (fixture pca_default_init_*)
            b.set_unpinned_rows(init["X_full"], np.where(np.asarray(init["X_var0"]) > 0, init["X_var0"], 1.0))
        return b
