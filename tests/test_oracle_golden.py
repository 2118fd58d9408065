"""Pins oracle/lds_closed_form.py against the reference's own outputs
(tests/golden/*.npz, produced by tests/golden/make_golden.py from the reference's
node classes).  CPU only.

Tolerances: 1e-10 relative on states / parameters / lower bound;
observed agreement is ~1e-13.
"""
import numpy as np

from oracle import lds_closed_form as O

RTOL = 1e-10


def _close(a, b, what, rtol=RTOL):
    a, b = np.asarray(a, dtype=float), np.asarray(b, dtype=float)
    scale = max(np.abs(b).max(), 1e-300)
    err = np.abs(a - b).max() / scale
    assert err <= rtol, "%s: rel err %.3e" % (what, err)


def _close_qld(a, b, what):
    """q_ln_det = 0.5 / s with s = sum(log diag chol) (quirk Q1) is ill-conditioned when
    s is near 0: compare s itself, whose rounding error is ~1e-16 * sum|log diag|."""
    sa, sb = 0.5 / np.asarray(a, dtype=float), 0.5 / np.asarray(b, dtype=float)
    ok = np.isfinite(sb)            # never-updated (fully observed) columns have no q_ln_det
    assert np.all(np.abs(sa - sb)[ok] <= 1e-10 * np.maximum(1.0, np.abs(sb[ok]))), what


def _prepare(st0, pri, T, Y=None):
    st = O.expand_state(st0, pri, T, Y)
    if pri["noise"] == "wishart":   # the fixture stores qw as the diagonal matrix diag(init Q_b)
        st["Q_b"] = np.einsum("nd,de->nde", st0["Q_b"], np.eye(st0["Q_b"].shape[1]))
        st["R_b"] = np.einsum("nd,de->nde", st0["R_b"], np.eye(st0["R_b"].shape[1]))
        O.init_noise_a(st, pri, T)
    return st


def _check_snapshot(st, parts, z, tag, kind):
    _close(st["X"][0], z[tag + "X"], tag + "X")
    cls = [0, 1, 2] if st["X"].shape[1] > 2 else [0, 2]      # no interior node when T == 2
    _close(st["Sigma"][0][cls], z[tag + "Sigma"][cls], tag + "Sigma")
    _close_qld(st["qld_x"][0][cls], z[tag + "qld_x"][cls], tag + "qld_x")
    _close(st["A_mean"][0], z[tag + "A_mean"], tag + "A_mean")
    _close(st["C_mean"][0], z[tag + "C_mean"], tag + "C_mean")
    for nm in ("A", "C"):
        cov = st[nm + "_cov"][0]
        if tag + nm + "_cov" in z:
            _close(cov, z[tag + nm + "_cov"], tag + nm + "_cov")
        else:
            _close(np.einsum("ikk->ik", cov), z[tag + nm + "_colvar"], tag + nm + "_colvar")
            assert z[tag + nm + "_cov_offdiag_max"] == 0.0
        _close_qld(st["qld_" + nm][0], z[tag + "qld_" + nm], tag + "qld_" + nm)
    for nm in ("Q_a", "Q_b", "R_a", "R_b"):
        _close(st[nm][0], z[tag + nm], tag + nm)
    if parts is not None and tag + "elbo_parts" in z:
        _close(parts[0], z[tag + "elbo_parts"], tag + "elbo_parts")
        tot, ref = parts[0].sum(), z[tag + "elbo_parts"].sum()
        assert abs(tot - ref) <= RTOL * abs(ref)


def test_interior_classes_coincide(golden):
    """The reference recomputes a Cholesky at every t; its interior covariances are
    identical, which is what the three-class closed form relies on."""
    meta, Y, st0, pri, z = golden
    for it in meta["iters"]:
        key = "it%d_interior_cov_spread" % it
        if key in z:
            assert z[key] <= 1e-13 * np.abs(z["it%d_Sigma" % it]).max()
            assert z["it%d_interior_qld_spread" % it] <= 1e-12


def test_forward_sweep_order(golden):
    """State after the first forward sweep only: pins the Gauss-Seidel order
    (new mu_{t-1}, old mu_{t+1})."""
    meta, Y, st0, pri, z = golden
    st = _prepare(st0, pri, meta["T"], Y)
    O.sweep(st, pri, Y, "forward")
    _close(st["X"][0], z["it1_fwd_X"], "forward sweep")


def test_iterations_match_reference(golden):
    meta, Y, st0, pri, z = golden
    st = _prepare(st0, pri, meta["T"], Y)
    missing = bool(np.isnan(Y).any())
    for it in range(1, max(meta["iters"]) + 1):
        parts = O.iterate(st, pri, Y, with_elbo=(meta["noise"] != "wishart"), update_outputs=missing)
        if it in meta["iters"]:
            _check_snapshot(st, parts, z, "it%d_" % it, meta["noise"])
            if missing:     # the outputs that are not fully observed: posterior means and (diagonal) covariances
                _close(st["Yq"][0], z["it%d_Yq" % it], "Yq")
                _close(st["Yvar"][0], z["it%d_Yvar" % it], "Yvar")
                assert z["it%d_Ycov_offdiag_max" % it] == 0.0


def test_single_updates_equal_sweep(golden):
    """update_x(t) for t = 0..T-1 is the forward sweep."""
    meta, Y, st0, pri, z = golden
    st = _prepare(st0, pri, meta["T"], Y)
    for t in range(meta["T"]):
        O.update_x(st, pri, Y, t)
    _close(st["X"][0], z["it1_fwd_X"], "stepwise forward sweep")
