"""TEST INFRASTRUCTURE: a stand-in for pyvb_amd.lds.LDSBatch that computes with the oracle (oracle/lds_closed_form.py)
instead of the HIP library, so that the HOST logic above the C ABI -- the recogniser, the queues, graphs of one structure
sharing a handle (pyvb_amd/_recognise.py: LDSGroup), Network.learn's schedule -- can be exercised without a GPU
(`-m "not gpu"`).  Plain graphs with DiagonalGamma / Gamma noise only.  Nothing under pyvb_amd/ imports this file; the
`-m gpu` tests run the same scenarios on the real handle.
"""
import numpy as np

from oracle import lds_closed_form as O


class OracleBatch(object):
    instances = []

    def __init__(self, N, T, D, K, noise="diagonal_gamma", device=0):
        assert noise in ("diagonal_gamma", "gamma")
        self.N, self.T, self.D, self.K, self.noise = N, T, D, K, noise
        self.st0, self.st, self.closed = {}, None, False
        self.log = []               # the calls that stand for launches, in order
        OracleBatch.instances.append(self)

    def set_priors(self, pri):
        self.pri = dict(pri)

    def set_observations(self, Y):
        assert Y.shape == (self.N, self.T, self.K) and not np.isnan(Y).any()
        self.Y = np.array(Y, dtype=float)

    def set_column_observations(self, A_obs=None, C_obs=None):
        raise NotImplementedError

    def set_state(self, **kw):
        if self.st is None:
            self.st0.update({k: np.array(v, dtype=float) for k, v in kw.items() if v is not None})
            if len(self.st0) == 7:
                self.st = O.expand_state(self.st0, self.pri, self.T)
            return
        st = self.st
        for k, v in kw.items():
            if v is None:
                continue
            v = np.array(v, dtype=float)
            if k in ("X", "A_mean", "C_mean"):
                st[k] = v
            elif k in ("A_colvar", "C_colvar"):
                st[k[0] + "_cov"] = np.einsum("nik,kl->nikl", v, np.eye(v.shape[2]))
            else:
                st[k] = v[:, 0].copy() if self.noise == "gamma" else v

    def set_posterior_classes(self, Sigma, qld_x=None):
        self.st["Sigma"] = np.array(Sigma, dtype=float)
        self.st["qld_x"] = np.array(qld_x, dtype=float)

    def sweep(self, direction="forward"):
        self.log.append(direction)
        O.sweep(self.st, self.pri, self.Y, direction)

    def _stats(self):
        assert "Sigma" in self.st, "PYVB_E_STALE"
        return O.statistics(self.st, self.Y)

    def update_columns(self, which, lo, hi):
        self.log.append((which, lo, hi))
        (O.update_A if which == "A" else O.update_C)(self.st, self.pri, self._stats(), cols=(lo, hi))

    def update_Q(self):
        self.log.append("Q")
        O.update_Q(self.st, self.pri, self._stats(), self.T)

    def update_R(self):
        self.log.append("R")
        O.update_R(self.st, self.pri, self._stats(), self.T)

    def elbo(self):
        self.log.append("elbo")
        return O.elbo_parts(self.st, self.pri, self._stats(), self.T)

    def get_state(self, what=None):
        st, N = self.st, self.N
        bc = lambda v, dim: np.broadcast_to(np.asarray(v, dtype=float).reshape(N, -1), (N, dim)).copy()
        return {"X": st["X"].copy(), "A_mean": st["A_mean"].copy(), "C_mean": st["C_mean"].copy(),
                "A_colvar": np.einsum("nikk->nik", st["A_cov"]).copy(), "C_colvar": np.einsum("nikk->nik", st["C_cov"]).copy(),
                "Q_a": bc(st["Q_a"], self.D), "Q_b": bc(st["Q_b"], self.D), "R_a": bc(st["R_a"], self.K), "R_b": bc(st["R_b"], self.K)}

    def get_posterior_classes(self):
        if "Sigma" not in self.st:
            return np.zeros((self.N, 3, self.D, self.D)), np.full((self.N, 3), np.nan)
        return self.st["Sigma"].copy(), self.st["qld_x"].copy()

    def get_column_qld(self):
        return self.st["qld_A"].copy(), self.st["qld_C"].copy()

    def close(self):
        self.closed = True
