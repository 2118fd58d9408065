"""GPU parity of the VB-PCA-with-missing-data path: against the fixtures produced by the reference
(tests/golden/pca_*.npz) and against the oracle, stage by stage.  Tolerance 1e-8 relative."""
import glob
import os

import numpy as np
import pytest

from oracle import pca_closed_form as P

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
FILES = sorted(glob.glob(os.path.join(HERE, "golden", "pca_*.npz")))
RTOL = 1e-8


def _load(path):
    from test_pca_oracle_golden import load_pca
    return load_pca(path)


def _close(a, b, what, rtol=RTOL):
    a, b = np.asarray(a, dtype=float), np.asarray(b, dtype=float)
    assert np.all(np.isfinite(a)), what + ": non-finite"
    err = np.abs(a - b).max() / max(np.abs(b).max(), 1e-300)
    assert err <= rtol, "%s: rel err %.3e" % (what, err)


def _compare(b, st, tag):
    g = b.get_state()
    for k in ("W_mean", "W_var", "Z", "Z_cov", "X", "Mu_mean", "Mu_var"):
        _close(g[k], st[k], tag + k)
    _close(g["beta_a"], st["beta_a"], tag + "beta_a")
    _close(g["beta_b"], st["beta_b"], tag + "beta_b")
    nm = (~st["obs"]).sum(1)
    ref_var = np.where(nm > 0, st["X_var"].max(1), 0.0)
    _close(g["X_rowvar"][nm > 0], ref_var[nm > 0], tag + "X_rowvar")


@pytest.mark.parametrize("path", FILES, ids=lambda p: os.path.basename(p)[4:-4])
def test_golden_fixtures(path):
    from pyvb_amd.pca import PCABatch
    N, d, q, init, pri, z = _load(path)
    b = PCABatch.from_problem(init, pri)
    for it in range(1, int(max(z["iters"])) + 1):
        b.iterate(1)
        if it in z["iters"]:
            tag = "it%d_" % it
            g = b.get_state()
            for k in ("W_mean", "W_var", "Z", "Z_cov", "X", "Mu_mean", "Mu_var", "beta_a", "beta_b"):
                _close(g[k], z[tag + k], tag + k)
            parts = b.elbo()
            ref = z[tag + "elbo_parts"]
            assert np.all(np.abs(parts - ref) <= RTOL * np.abs(ref).sum()), (parts, ref)
    b.close()


@pytest.mark.parametrize("sweep", ["rows", "pairs"])
@pytest.mark.parametrize("path", [f for f in FILES if "default_init" not in f], ids=lambda p: os.path.basename(p)[4:-4])
def test_golden_fixtures_through_the_row_owning_sweeps(path, sweep, monkeypatch):
    """k_pca_rows (a wavefront owns whole rows) and k_pca_pairs (a pair of wavefronts does, half the columns each), chosen with
    PYVB_PCA_SWEEP at handle creation (pyvb_amd/csrc/k_pca.hip): they have to give the reference's numbers all the same."""
    from pyvb_amd.pca import PCABatch
    monkeypatch.setenv("PYVB_PCA_SWEEP", sweep)
    N, d, q, init, pri, z = _load(path)
    b = PCABatch.from_problem(init, pri)
    for it in range(1, int(max(z["iters"])) + 1):
        b.iterate(1)
        if it in z["iters"]:
            tag = "it%d_" % it
            g = b.get_state()
            for k in ("W_mean", "W_var", "Z", "Z_cov", "X", "Mu_mean", "Mu_var", "beta_a", "beta_b"):
                _close(g[k], z[tag + k], tag + k)
            parts = b.elbo()
            ref = z[tag + "elbo_parts"]
            assert np.all(np.abs(parts - ref) <= RTOL * np.abs(ref).sum()), (parts, ref)
    b.close()


@pytest.mark.parametrize("sweep", ["columns", "rows", "pairs"])
def test_imputed_entries_are_recomputed_not_stored(sweep, monkeypatch):
    """Round 4: the sweep of an iteration leaves the imputed entries of X unstored (they are <W> z_n + <Mu> of what IS stored; the
    next sweep recomputes them, pyvb_pca_get_state and every other reader has them put into X first: pca_materialize_x).  A run
    that is interrupted by reads, by a partial row update and by a Z update of its own ends bitwise where an uninterrupted one
    ends, and both agree with a handle that stores the entries (PYVB_PCA_WRITEBACK=1) to rounding."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("make_golden", os.path.join(HERE, "golden", "make_golden.py"))
    G = importlib.util.module_from_spec(spec); spec.loader.exec_module(G)
    from pyvb_amd.pca import PCABatch
    N, d, q = 3000, 250, 16
    init, pri = G.pca_problem(N, d, q, seed=77)
    monkeypatch.setenv("PYVB_PCA_SWEEP", sweep)
    a = PCABatch.from_problem(init, pri); a.iterate(5); sa = a.get_state(); ea = a.elbo(); a.close()
    b = PCABatch.from_problem(init, pri)
    b.iterate(2)
    mid = b.get_state()                                 # materialises X
    assert np.isfinite(mid["X"]).all()
    b.iterate(1)
    b.elbo()
    b.iterate(2)
    sb = b.get_state(); eb = b.elbo(); b.close()
    for k in ("X", "Z", "W_mean", "Mu_mean", "beta_b"):
        assert np.array_equal(sa[k], sb[k]), k
    assert np.array_equal(ea, eb)
    monkeypatch.setenv("PYVB_PCA_WRITEBACK", "1")
    c = PCABatch.from_problem(init, pri); c.iterate(5); sc = c.get_state(); ec = c.elbo(); c.close()
    for k in ("X", "Z", "W_mean", "Mu_mean"):
        _close(sa[k], sc[k], "lazy vs stored: " + k)
    assert np.all(np.abs(ea - ec) <= 1e-10 * np.abs(ec).sum())
    # the oracle, too
    st = P.make_state(init, pri, N, d, q)
    for _ in range(5):
        ref = P.iterate(st, pri)
    _close(sa["X"], st["X"], "X after five iterations")
    assert np.all(np.abs(ea - ref) <= RTOL * np.abs(ref).sum())


def test_the_sweep_a_long_problem_gets_by_default_with_fewer_than_sixteen_latents(monkeypatch):
    """From 512 rows per CU on (and d > 192) a handle takes the pair-owning sweep without being asked (api_pca.hip); here with q = 7
    and d = 250 -- padded latent indices and a padded column tile in the kernel that otherwise only meets q = 16 at that size --
    against the oracle and against the column-owning sweep on the same problem."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("make_golden", os.path.join(HERE, "golden", "make_golden.py"))
    G = importlib.util.module_from_spec(spec); spec.loader.exec_module(G)
    from pyvb_amd.pca import PCABatch
    N, d, q = 140000, 250, 7
    init, pri = G.pca_problem(N, d, q, seed=5)
    monkeypatch.delenv("PYVB_PCA_SWEEP", raising=False)
    a = PCABatch.from_problem(init, pri); a.iterate(3); sa = a.get_state(); ea = a.elbo(); a.close()
    monkeypatch.setenv("PYVB_PCA_SWEEP", "columns")
    c = PCABatch.from_problem(init, pri); c.iterate(3); sc = c.get_state(); ec = c.elbo(); c.close()
    st = P.make_state(init, pri, N, d, q)
    for _ in range(3):
        ref = P.iterate(st, pri)
    for k in ("W_mean", "W_var", "Z", "X", "Mu_mean", "beta_b"):
        _close(sa[k], st[k], "default sweep vs oracle: " + k)
        _close(sa[k], sc[k], "default sweep vs column-owning sweep: " + k)
    assert np.all(np.abs(ea - ref) <= RTOL * np.abs(ref).sum()) and np.all(np.abs(ea - ec) <= 1e-10 * np.abs(ec).sum())


@pytest.mark.parametrize("N,d,q", [(300, 20, 4), (1000, 64, 16), (77, 33, 17), (5000, 256, 16), (16, 3, 1), (17, 250, 31)])
def test_stagewise_vs_oracle(N, d, q):
    import importlib.util
    spec = importlib.util.spec_from_file_location("make_golden", os.path.join(HERE, "golden", "make_golden.py"))
    G = importlib.util.module_from_spec(spec); spec.loader.exec_module(G)
    from pyvb_amd.pca import PCABatch
    init, pri = G.pca_problem(N, d, q, seed=1000 + N + d)
    st = P.make_state(init, pri, N, d, q)
    b = PCABatch.from_problem(init, pri)
    for it in range(2):
        tag = "it%d " % it
        P.update_W(st, pri); b.update_W()
        _close(b.get_state()["W_mean"], st["W_mean"], tag + "W after update_W")
        P.update_Z(st, pri); b.update_Z()
        _close(b.get_state()["Z"], st["Z"], tag + "Z after update_Z")
        P.update_X(st, pri, 0, 1); b.update_X(0, 1)
        P.update_Mu(st, pri); b.update_Mu()
        _close(b.get_state()["Mu_mean"], st["Mu_mean"], tag + "Mu after update_Mu")
        P.update_X(st, pri, 1, N); b.update_X(1, N)
        _close(b.get_state()["X"], st["X"], tag + "X after imputation")
        P.update_Beta(st, pri); b.update_Beta()
        _compare(b, st, tag)
        ref = P.elbo_parts(st, pri)
        got = b.elbo()
        assert np.all(np.abs(got - ref) <= RTOL * np.abs(ref).sum()), (got, ref)
    b.close()


@pytest.mark.parametrize("order", ["Z r", "Z X0 r", "Z Z r", "Z X0 Mu Xpart r X r", "Z W r X r", "Z X0 Z X r", "Z Beta r", "Z X0 Mu X Beta W Z elbo X0 r X r"])
def test_a_deferred_z_update_is_carried_out_by_whatever_comes_next(order):
    """pyvb_pca_update_Z only prepares the update (posterior covariance, gains, the sum of z from the sum of x); the rows of Z are
    written by the next sweep over X, or by the first call that reads or replaces them.  Every continuation must give what the
    reference's order of node updates gives: checked against the oracle after each read ("r")."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("make_golden", os.path.join(HERE, "golden", "make_golden.py"))
    G = importlib.util.module_from_spec(spec); spec.loader.exec_module(G)
    from pyvb_amd.pca import PCABatch
    N, d, q = 150, 40, 5
    init, pri = G.pca_problem(N, d, q, seed=99)
    st = P.make_state(init, pri, N, d, q)
    b = PCABatch.from_problem(init, pri)
    P.update_W(st, pri); b.update_W()
    for op in order.split():
        if op == "Z": P.update_Z(st, pri); b.update_Z()
        elif op == "X0": P.update_X(st, pri, 0, 1); b.update_X(0, 1)
        elif op == "Mu": P.update_Mu(st, pri); b.update_Mu()
        elif op == "Xpart": P.update_X(st, pri, 20, 97); b.update_X(20, 97)
        elif op == "X": P.update_X(st, pri, 1, N); b.update_X(1, N)
        elif op == "W": P.update_W(st, pri); b.update_W()
        elif op == "Beta": P.update_Beta(st, pri); b.update_Beta()
        elif op == "elbo":
            ref, got = P.elbo_parts(st, pri), b.elbo()
            assert np.all(np.abs(got - ref) <= RTOL * np.abs(ref).sum()), (got, ref)
        else:
            _compare(b, st, order + ": ")
    _compare(b, st, order + " (end): ")
    b.close()


def test_iterate_equals_individual_calls():
    import importlib.util
    spec = importlib.util.spec_from_file_location("make_golden", os.path.join(HERE, "golden", "make_golden.py"))
    G = importlib.util.module_from_spec(spec); spec.loader.exec_module(G)
    from pyvb_amd.pca import PCABatch
    N, d, q = 400, 24, 5
    init, pri = G.pca_problem(N, d, q, seed=5)
    a, b = PCABatch.from_problem(init, pri), PCABatch.from_problem(init, pri)
    a.iterate(3)
    for _ in range(3):
        b.update_W(); b.update_Z(); b.update_X(0, 1); b.update_Mu(); b.update_X(1, N); b.update_Beta()
    ga, gb = a.get_state(), b.get_state()
    for k in ga:
        assert np.array_equal(ga[k], gb[k]), k
    assert np.array_equal(a.elbo(), b.elbo())
    a.close(); b.close()


def test_single_rank_communicator_is_transparent():
    """With a one-rank RCCL communicator attached every statistics exchange goes through ncclAllReduce
    (sum over one rank = identity): the results must not change."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("make_golden", os.path.join(HERE, "golden", "make_golden.py"))
    G = importlib.util.module_from_spec(spec); spec.loader.exec_module(G)
    from pyvb_amd.pca import PCABatch
    from pyvb_amd.lds import LDSBatch
    N, d, q = 500, 40, 6
    init, pri = G.pca_problem(N, d, q, seed=77)
    a = PCABatch.from_problem(init, pri)
    b = PCABatch(N, d, q)
    b.comm_init(LDSBatch.comm_unique_id(), 0, 1)
    b.set_priors(pri)
    b.set_data(np.where(init["obs"], init["X"], np.nan))
    b.set_state(X_missing=init["X"], W_mean=init["W_mean"], Z=init["Z"], Z_cov=init["Z_cov"], Mu_mean=init["Mu_mean"], beta_b=float(init["beta_b"]))
    a.iterate(3); b.iterate(3)
    ga, gb = a.get_state(), b.get_state()
    for k in ga:
        assert np.array_equal(ga[k], gb[k]), k
    assert np.array_equal(a.elbo(), b.elbo())
    a.close(); b.close()


def test_baseline_config5_full_size_on_one_gpu():
    """BASELINE configs[4] at its full size on one GPU: N = 10^6 rows x d = 256, q = 16, Bernoulli(0.1) mask -- one
    Network.learn iteration against the oracle (about a minute of numpy), then a size-independent property: the rows
    after the first can be permuted without changing the posteriors of W, Mu and Beta (sums over n in another order)."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("make_golden", os.path.join(HERE, "golden", "make_golden.py"))
    G = importlib.util.module_from_spec(spec); spec.loader.exec_module(G)
    from pyvb_amd.pca import PCABatch
    N, d, q = 1000000, 256, 16
    init, pri = G.pca_problem(N, d, q, seed=2024, p_missing=0.1)
    st = P.make_state(init, pri, N, d, q)
    ref = P.iterate(st, pri)
    b = PCABatch.from_problem(init, pri)
    b.iterate(1)
    got = b.elbo()
    g = b.get_state()
    for k in ("W_mean", "W_var", "Z_cov", "Mu_mean", "Mu_var", "beta_a", "beta_b"):
        _close(g[k], st[k], "N = 10^6: " + k)
    _close(g["Z"][:2000], st["Z"][:2000], "N = 10^6: Z (first rows)")
    _close(g["X"][-2000:], st["X"][-2000:], "N = 10^6: X (last rows)")
    assert np.all(np.abs(got - ref) <= RTOL * np.abs(ref).sum()), (got, ref)
    b.close()
    del st
    perm = np.concatenate([[0], 1 + np.random.default_rng(0).permutation(N - 1)])
    init2 = {k: (v[perm] if isinstance(v, np.ndarray) and v.shape[:1] == (N,) else v) for k, v in init.items()}
    b2 = PCABatch.from_problem(init2, pri)
    b2.iterate(1)
    g2 = b2.get_state()
    for k in ("W_mean", "W_var", "Z_cov", "Mu_mean", "Mu_var", "beta_b"):
        _close(g2[k], g[k], "row permutation: " + k, 1e-10)
    _close(g2["Z"][:100], g["Z"][perm[:100]], "row permutation: Z", 1e-10)
    b2.close()
