"""LDSBatch: N independent replicates of the linear-dynamical-system graph of the
reference's examples/Linear_Dynamic_System.py:46-66, resident on one MI355X.

Thin Python over the C ABI (include/pyvb_hip.h); every method is one or a few
kernel launches.  The node classes in pyvb_amd.nodes bind to an LDSBatch with
N = 1; bench.py and the parity tests drive it directly.
"""
import numpy as np

from . import _capi as C

__all__ = ["LDSBatch"]

_NOISE = {"diagonal_gamma": C.NOISE_DIAGONAL_GAMMA, "gamma": C.NOISE_GAMMA, "wishart": C.NOISE_WISHART}


def _f64(a, shape, name):
    a = np.ascontiguousarray(a, dtype=np.float64)
    if a.shape != tuple(shape):
        raise AssertionError("%s has shape %s, expected %s" % (name, a.shape, tuple(shape)))
    return a


class LDSBatch(object):
    ELBO_PARTS = ("X", "Y", "A", "C", "Q", "R")

    def __init__(self, N, T, D, K, noise="diagonal_gamma", device=0):
        if noise not in _NOISE:
            raise NotImplementedError("noise precision %r has no HIP path (DiagonalGamma, Gamma and Wishart do)" % (noise,))
        self.N, self.T, self.D, self.K, self.noise, self.device = int(N), int(T), int(D), int(K), noise, int(device)
        h = C.ctypes.c_void_p()
        C.check(C.lib.pyvb_lds_create(C.ctypes.byref(h), self.device, self.N, self.T, self.D, self.K, _NOISE[noise]))
        self._h = h

    # -- lifetime ---------------------------------------------------------------------------
    def close(self):
        if getattr(self, "_h", None):
            C.lib.pyvb_lds_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- inputs -----------------------------------------------------------------------------
    def set_priors(self, pri):
        """pri: dict as pyvb_amd.synth.default_priors (Constant parents of X_0 and of the
        columns, Gamma-family hyper-parameters).  Scalars are broadcast for the Gamma kind."""
        D, K = self.D, self.K
        bc = lambda v, n: np.ascontiguousarray(np.broadcast_to(np.asarray(v, dtype=np.float64), (n,)))
        arrs = [
            _f64(pri["x0_mean"], (D,), "x0_mean"), _f64(pri["x0_prec"], (D, D), "x0_prec"),
            _f64(pri["A_prior_mean"], (D, D), "A_prior_mean"), _f64(pri["A_prior_prec"], (D, D), "A_prior_prec"),
            _f64(pri["C_prior_mean"], (K, D), "C_prior_mean"), _f64(pri["C_prior_prec"], (D, K), "C_prior_prec"),
        ]
        if self.noise == "wishart":     # Wishart(dim, v0, w0): Q_a0 / R_a0 hold v0, Q_b0 / R_b0 the matrices w0
            C.check(C.lib.pyvb_lds_set_priors(self._h, *([C.dptr(a) for a in arrs] + [None] * 4)))
            qw0, rw0 = _f64(pri["Q_b0"], (D, D), "Q_w0"), _f64(pri["R_b0"], (K, K), "R_w0")
            C.check(C.lib.pyvb_lds_set_wishart_priors(self._h, float(pri["Q_a0"]), C.dptr(qw0), float(pri["R_a0"]), C.dptr(rw0)))
            return
        arrs += [bc(pri["Q_a0"], D), bc(pri["Q_b0"], D), bc(pri["R_a0"], K), bc(pri["R_b0"], K)]
        C.check(C.lib.pyvb_lds_set_priors(self._h, *[C.dptr(a) for a in arrs]))

    def set_observations(self, Y):
        """Y[N,T,K]; NaN = missing entry (the rows concerned become variational nodes: set_output_state, update_Y)."""
        Y = _f64(Y, (self.N, self.T, self.K), "Y")
        C.check(C.lib.pyvb_lds_set_observations(self._h, C.dptr(Y)))

    def set_output_state(self, Yq, Yrowvar):
        """Initial posterior of the outputs that are not fully observed: means [N,T,K], isotropic variances [N,T]."""
        q, v = _f64(Yq, (self.N, self.T, self.K), "Yq"), _f64(Yrowvar, (self.N, self.T), "Yrowvar")
        C.check(C.lib.pyvb_lds_set_output_state(self._h, C.dptr(q), C.dptr(v)))

    def update_Y(self):
        """[y.update() for y in Ys if not y.observed]"""
        C.check(C.lib.pyvb_lds_update_Y(self._h))

    def get_outputs(self, with_qld=False):
        """(posterior means [N,T,K], variances [N,T,K]) of the outputs; fully observed rows: (value, 0).
        with_qld: also q_ln_det [N,T] of the rows updated so far (NaN otherwise)."""
        q, v = np.empty((self.N, self.T, self.K)), np.empty((self.N, self.T, self.K))
        ld = np.empty((self.N, self.T)) if with_qld else None
        C.check(C.lib.pyvb_lds_get_outputs(self._h, C.dptr(q), C.dptr(v), C.dptr(ld)))
        return (q, v, ld) if with_qld else (q, v)

    def set_state(self, X=None, A_mean=None, A_colvar=None, C_mean=None, C_colvar=None, Q_b=None, R_b=None):
        N, T, D, K = self.N, self.T, self.D, self.K
        shapes = [("X", X, (N, T, D)), ("A_mean", A_mean, (N, D, D)), ("A_colvar", A_colvar, (N, D, D)),
                  ("C_mean", C_mean, (N, K, D)), ("C_colvar", C_colvar, (N, D, K)), ("Q_b", Q_b, (N, D)), ("R_b", R_b, (N, K))]
        arrs = [None if a is None else _f64(a, s, nm) for nm, a, s in shapes]
        C.check(C.lib.pyvb_lds_set_state(self._h, *[C.dptr(a) for a in arrs]))

    def set_wishart_state(self, Q_w=None, R_w=None):
        """Posterior qw of the Wishart nodes, [N,D,D] and [N,K,K] (nodes_todo.py:216-217 draws a random rank-one one)."""
        q = None if Q_w is None else _f64(Q_w, (self.N, self.D, self.D), "Q_w")
        r = None if R_w is None else _f64(R_w, (self.N, self.K, self.K), "R_w")
        C.check(C.lib.pyvb_lds_set_wishart_state(self._h, C.dptr(q), C.dptr(r)))

    def get_wishart_state(self):
        N, D, K = self.N, self.D, self.K
        out = {"Q_v": np.empty(N), "Q_w": np.empty((N, D, D)), "R_v": np.empty(N), "R_w": np.empty((N, K, K))}
        C.check(C.lib.pyvb_lds_get_wishart_state(self._h, C.dptr(out["Q_v"]), C.dptr(out["Q_w"]), C.dptr(out["R_v"]), C.dptr(out["R_w"])))
        return out

    def set_column_cov(self, A_cov=None, C_cov=None):
        a = None if A_cov is None else _f64(A_cov, (self.N, self.D, self.D, self.D), "A_cov")
        c = None if C_cov is None else _f64(C_cov, (self.N, self.D, self.K, self.K), "C_cov")
        C.check(C.lib.pyvb_lds_set_column_cov(self._h, C.dptr(a), C.dptr(c)))

    def get_column_cov(self):
        """Dense posterior covariances of the columns of A ([N,D,D,D]) and C ([N,D,K,K]); Wishart noise only."""
        A, Cc = np.empty((self.N, self.D, self.D, self.D)), np.empty((self.N, self.D, self.K, self.K))
        C.check(C.lib.pyvb_lds_get_column_cov(self._h, C.dptr(A), C.dptr(Cc)))
        return A, Cc

    def set_column_observations(self, A_obs=None, C_obs=None):
        """Known entries of A ([D,D]) and C ([K,D]) as (row, col) arrays with NaN where unknown
        (As[i].observe(...), examples/LDS_knowns_in_A.py:73-74).  Call after set_state."""
        a = None if A_obs is None else _f64(A_obs, (self.D, self.D), "A_obs")
        c = None if C_obs is None else _f64(C_obs, (self.K, self.D), "C_obs")
        C.check(C.lib.pyvb_lds_set_column_observations(self._h, C.dptr(a), C.dptr(c)))

    # -- outputs ----------------------------------------------------------------------------
    def get_state(self, what=("X", "A_mean", "A_colvar", "C_mean", "C_colvar", "Q_a", "Q_b", "R_a", "R_b")):
        N, T, D, K = self.N, self.T, self.D, self.K
        shapes = {"X": (N, T, D), "A_mean": (N, D, D), "A_colvar": (N, D, D), "C_mean": (N, K, D), "C_colvar": (N, D, K),
                  "Q_a": (N, D), "Q_b": (N, D), "R_a": (N, K), "R_b": (N, K)}
        order = ["X", "A_mean", "A_colvar", "C_mean", "C_colvar", "Q_a", "Q_b", "R_a", "R_b"]
        out = {k: np.empty(shapes[k]) for k in order if k in what}
        C.check(C.lib.pyvb_lds_get_state(self._h, *[C.dptr(out.get(k)) for k in order]))
        return out

    def get_posterior_classes(self):
        """(Sigma[N,3,D,D], q_ln_det[N,3]) of X_0, the interior X_t and X_{T-1} as of their last update."""
        S = np.empty((self.N, 3, self.D, self.D))
        q = np.empty((self.N, 3))
        C.check(C.lib.pyvb_lds_get_posterior_classes(self._h, C.dptr(S), C.dptr(q)))
        return S, q

    def set_posterior_classes(self, Sigma, qld_x=None):
        """Initial covariances of X_0, the interior X_t and X_{T-1} ([N,3,D,D]); see include/pyvb_hip.h."""
        S = _f64(Sigma, (self.N, 3, self.D, self.D), "Sigma")
        q = None if qld_x is None else _f64(qld_x, (self.N, 3), "qld_x")
        C.check(C.lib.pyvb_lds_set_posterior_classes(self._h, C.dptr(S), C.dptr(q)))

    def get_column_qld(self):
        qa, qc = np.empty((self.N, self.D)), np.empty((self.N, self.D))
        C.check(C.lib.pyvb_lds_get_column_qld(self._h, C.dptr(qa), C.dptr(qc)))
        return qa, qc

    def get_warmup(self):
        w = np.empty((self.N, 2), dtype=np.int32)
        C.check(C.lib.pyvb_lds_get_warmup(self._h, w.ctypes.data_as(C._ip)))
        return w

    def get_time_split(self):
        w = C.ctypes.c_int()
        C.check(C.lib.pyvb_lds_get_time_split(self._h, C.ctypes.byref(w)))
        return w.value

    def set_time_split(self, W):
        """Wavefronts per replicate in the sweeps (chosen by the library; tests force W = 1, the headline code path)."""
        C.check(C.lib.pyvb_lds_set_time_split(self._h, int(W)))

    # -- updates ----------------------------------------------------------------------------
    def sweep(self, direction="forward"):
        C.check(C.lib.pyvb_lds_sweep(self._h, C.FORWARD if direction == "forward" else C.BACKWARD))

    def update_x(self, t):
        C.check(C.lib.pyvb_lds_update_x(self._h, int(t)))

    def update_A(self):
        C.check(C.lib.pyvb_lds_update_A(self._h))

    def update_C(self):
        C.check(C.lib.pyvb_lds_update_C(self._h))

    def update_columns(self, which, col_begin, col_end):
        """As[i].update() (which = "A") or Cs[i].update() ("C") for i in [col_begin, col_end), in order."""
        C.check(C.lib.pyvb_lds_update_columns(self._h, 0 if which == "A" else 1, int(col_begin), int(col_end)))

    def update_Q(self):
        C.check(C.lib.pyvb_lds_update_Q(self._h))

    def update_R(self):
        C.check(C.lib.pyvb_lds_update_R(self._h))

    def elbo(self):
        """Per-replicate lower-bound parts [N,6] (X, Y, A, C, Q, R), reference mode."""
        C.check(C.lib.pyvb_lds_elbo(self._h))
        out = np.empty((self.N, 6))
        C.check(C.lib.pyvb_lds_get_elbo(self._h, C.dptr(out)))
        return out

    def elbo_total(self):
        """Parts summed over replicates (and over ranks when a communicator is attached)."""
        out = np.empty(6)
        C.check(C.lib.pyvb_lds_elbo_total(self._h, C.dptr(out)))
        return out

    def iterate(self, niters=1):
        """niters x (forward sweep, backward sweep, A, C, Q, R, lower bound); asynchronous."""
        C.check(C.lib.pyvb_lds_iterate(self._h, int(niters)))

    def elbo_history(self, last=4096):
        """Lower-bound parts of the most recent iterate() iterations, [count, 6], summed over the replicates (and over
        the ranks when a communicator is attached); oldest first."""
        out = np.empty((int(last), 6))
        cnt = C.ctypes.c_int()
        C.check(C.lib.pyvb_lds_get_elbo_history(self._h, C.dptr(out), int(last), C.ctypes.byref(cnt)))
        return out[:cnt.value].copy()

    def reset_elbo_history(self):
        C.check(C.lib.pyvb_lds_reset_elbo_history(self._h))

    def sync(self):
        C.check(C.lib.pyvb_lds_sync(self._h))

    # -- measurement ------------------------------------------------------------------------
    def timing(self, on=True):
        C.check(C.lib.pyvb_lds_timing_enable(self._h, 1 if on else 0))
        C.check(C.lib.pyvb_lds_timing_reset(self._h))

    def kernel_times(self):
        names = {"prep": C.K_PREP, "sweep_fwd": C.K_SWEEP_FWD, "sweep_bwd": C.K_SWEEP_BWD, "stats": C.K_STATS,
                 "params": C.K_PARAMS, "elbo": C.K_ELBO, "step": C.K_STEP, "gy": C.K_GY}
        out = {}
        for nm, k in names.items():
            ms, cnt = C.ctypes.c_double(), C.ctypes.c_int()
            C.check(C.lib.pyvb_lds_timing_get(self._h, k, C.ctypes.byref(ms), C.ctypes.byref(cnt)))
            out[nm] = (ms.value, cnt.value)
        return out

    # -- multi-GPU --------------------------------------------------------------------------
    @staticmethod
    def comm_unique_id():
        buf = C.ctypes.create_string_buffer(128)
        C.check(C.lib.pyvb_comm_unique_id(buf))
        return buf.raw

    def comm_init(self, uid, rank, world):
        C.check(C.lib.pyvb_lds_comm_init(self._h, uid, int(rank), int(world)))

    def comm_init_host(self, comm, rank, world):
        """The ELBO all-reduce through a host process group (pyvb_amd.dist) instead of RCCL: rehearsal of the sharded
        path with several ranks on one GPU (pyvb_lds_comm_init_host)."""
        from .dist import host_allreduce_callback
        self._host_cb = host_allreduce_callback(comm)          # kept alive with the handle
        C.check(C.lib.pyvb_lds_comm_init_host(self._h, self._host_cb, None, int(rank), int(world)))

    # -- convenience ------------------------------------------------------------------------
    @classmethod
    def from_problem(cls, Y, st0, pri, device=0):
        N, T, K = Y.shape
        D = st0["A_mean"].shape[1]
        b = cls(N, T, D, K, pri.get("noise", "diagonal_gamma"), device)
        b.set_priors(pri)
        b.set_observations(Y)
        if "Yq" in st0 and np.isnan(Y).any():
            b.set_output_state(st0["Yq"], st0["Yrowvar"])
        if b.noise == "wishart":        # the compact initial state carries the diagonal of qw in Q_b / R_b
            b.set_state(**{k: st0[k] for k in ("X", "A_mean", "A_colvar", "C_mean", "C_colvar")})
            dense = lambda v: np.einsum("nd,de->nde", v, np.eye(v.shape[1])) if v.ndim == 2 else v
            b.set_wishart_state(dense(st0["Q_b"]), dense(st0["R_b"]))
        else:
            b.set_state(**{k: st0[k] for k in ("X", "A_mean", "A_colvar", "C_mean", "C_colvar", "Q_b", "R_b")})
        if pri.get("A_obs") is not None or pri.get("C_obs") is not None:
            b.set_column_observations(pri.get("A_obs"), pri.get("C_obs"))
        return b
