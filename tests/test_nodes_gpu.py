"""GPU: the reference's example script, verbatim in structure, on pyvb_amd.nodes -- checked against
the fixtures the reference produced (tests/golden/*.npz) for the same inputs."""
import importlib.util
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
RTOL = 1e-8


def _golden_module():
    spec = importlib.util.spec_from_file_location("make_golden", os.path.join(HERE, "golden", "make_golden.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def _rel(a, b):
    return np.abs(np.asarray(a) - np.asarray(b)).max() / max(np.abs(b).max(), 1e-300)


def test_example_loop_through_the_node_api(golden):
    from pyvb_amd import nodes
    meta, Y, st0, pri, z = golden
    g = _golden_module().build_graph(nodes, Y[0], pri, {k: v for k, v in st0.items()})
    Xs, As, Cs, Q, R = g["Xs"], g["As"], g["Cs"], g["Q"], g["R"]
    missing = bool(np.isnan(Y).any())
    if missing:         # explicit initial posterior of the outputs that are not fully observed (as make_golden.run_case does)
        for t, y in enumerate(g["Ys"]):
            if not y.observed:
                y.qmu = st0["Yq"][0, t].reshape(-1, 1).copy()
                y.qcov = np.eye(meta["K"]) * st0["Yrowvar"][0, t]
    plan = None
    for it in range(1, max(meta["iters"]) + 1):
        # examples/Linear_Dynamic_System.py:69-77
        [x.update() for x in Xs]
        if it == 1:
            assert _rel(np.hstack([x.qmu for x in Xs]).T, z["it1_fwd_X"]) <= RTOL
        Xs.reverse()
        [x.update() for x in Xs]
        Xs.reverse()
        if missing:
            [y.update() for y in g["Ys"] if not y.observed]
        [a.update() for a in As]
        [c.update() for c in Cs]
        Q.update()
        R.update()
        if it in meta["iters"]:
            tag = "it%d_" % it
            T = meta["T"]
            assert _rel(np.hstack([x.qmu for x in Xs]).T, z[tag + "X"]) <= RTOL
            assert _rel(np.hstack([a.qmu for a in As]), z[tag + "A_mean"]) <= RTOL
            assert _rel(np.hstack([c.qmu for c in Cs]), z[tag + "C_mean"]) <= RTOL
            assert _rel(Xs[0].qcov, z[tag + "Sigma"][0]) <= RTOL
            assert _rel(Xs[T - 1].qcov, z[tag + "Sigma"][2]) <= RTOL
            if T > 2:
                assert _rel(Xs[1].qcov, z[tag + "Sigma"][1]) <= RTOL
            plan = Xs[0]._plan
            if meta["noise"] == "wishart":      # first update only; no reference lower bound (Q8)
                assert _rel(Q.qw, z[tag + "Q_b"]) <= RTOL and _rel(R.qw, z[tag + "R_b"]) <= RTOL
                assert abs(Q.qv - float(z[tag + "Q_a"])) <= 1e-12 and abs(R.qv - float(z[tag + "R_a"])) <= 1e-12
                assert _rel(np.stack([a.qcov for a in As]), z[tag + "A_cov"]) <= RTOL
                qw = z[tag + "Q_b"]            # the expectation uses the symmetric part of the reference's qw (k_wishart.hip)
                assert _rel(Q.pass_down_Ex(), float(z[tag + "Q_a"]) * np.linalg.inv(0.5 * (qw + qw.T))) <= 1e-7
                assert np.isfinite(Q.log_lower_bound()) and np.isfinite(plan.elbo_parts()).all()
                continue
            assert _rel(np.asarray(Q.qb, dtype=float), z[tag + "Q_b"]) <= RTOL
            assert _rel(np.asarray(R.qb, dtype=float), z[tag + "R_b"]) <= RTOL
            assert _rel(np.asarray(Q.qa, dtype=float), z[tag + "Q_a"]) <= RTOL
            if missing:
                from pyvb_amd._recognise import LDSPlan
                assert isinstance(plan, LDSPlan)        # still the fused kernels
                assert _rel(np.hstack([y.qmu for y in g["Ys"]]).T, z[tag + "Yq"]) <= RTOL
                assert _rel(np.stack([np.diag(y.qcov) for y in g["Ys"]]), z[tag + "Yvar"]) <= RTOL
            parts = plan.elbo_parts()
            ref = z[tag + "elbo_parts"]
            assert abs(parts.sum() - ref.sum()) <= RTOL * abs(ref.sum())
            assert abs(Q.log_lower_bound() - ref[4]) <= RTOL * np.abs(ref).sum()
            # accessors the example's plotting code uses (:122-128)
            ymean = np.hstack([y.mean_parent.pass_down_Ex() for y in g["Ys"]]).T
            assert _rel(ymean, z[tag + "X"] @ z[tag + "C_mean"].T) <= 1e-7
    assert plan is not None and plan.pending == []


def test_network_learn_matches_reference_bound():
    """Network.learn over the fetched network: updates in crawl order, lower bound from the device."""
    from pyvb_amd import nodes, synth
    from pyvb_amd.network import Network
    from oracle import lds_closed_form as O
    T, D, K = 40, 3, 4
    Y, st0, pri = synth.make_problem(T, D, K, 1, 11)
    g = _golden_module().build_graph(nodes, Y[0], pri, st0)
    net = Network(g["Xs"] + g["Ys"] + g["As"] + g["Cs"] + [g["Q"], g["R"]])
    net.learn(3, tol=-np.inf, verbose=False)
    # the same order in the oracle: forward sweep, A, C, Q, R, then the bound
    st = O.expand_state(st0, pri, T)
    for _ in range(3):
        O.sweep(st, pri, Y, "forward")
        S = O.statistics(st, Y)
        O.update_A(st, pri, S); O.update_C(st, pri, S); O.update_Q(st, pri, S, T); O.update_R(st, pri, S, T)
        ref = O.elbo_parts(st, pri, S, T)[0].sum()
    assert abs(net.llb - ref) <= RTOL * abs(ref)
    assert _rel(np.hstack([x.qmu for x in g["Xs"]]).T, st["X"][0]) <= RTOL
    # the convergence predicate of network.py:53 also fires on a decreasing bound
    net2 = Network(g["Xs"] + g["Ys"] + g["As"] + g["Cs"] + [g["Q"], g["R"]])
    net2.learn(50, tol=np.inf, verbose=False)           # "llb - old < tol" is immediately true
    assert np.isfinite(net2.llb)


def test_partial_order_and_attribute_writes():
    """Out-of-pattern requests (single nodes, a subset of columns) run as individual launches and
    give what the oracle gives; assigning a posterior attribute reaches the device."""
    from pyvb_amd import nodes, synth
    from oracle import lds_closed_form as O
    T, D, K = 12, 4, 3
    Y, st0, pri = synth.make_problem(T, D, K, 1, 23)
    g = _golden_module().build_graph(nodes, Y[0], pri, st0)
    Xs, As = g["Xs"], g["As"]
    st = O.expand_state(st0, pri, T)
    order = [5, 2, 7, 0, 11, 3]
    for t in order:
        Xs[t].update()
        O.update_x(st, pri, Y, t)
    assert _rel(np.hstack([x.qmu for x in Xs]).T, st["X"][0]) <= RTOL
    [x.update() for x in Xs]; O.sweep(st, pri, Y, "forward")
    As[0].update(); As[1].update()                       # two of four columns
    S = O.statistics(st, Y)
    full = {k: (v.copy() if isinstance(v, np.ndarray) else v) for k, v in st.items()}
    O.update_A(full, pri, S)
    got = np.hstack([a.qmu for a in As])
    assert _rel(got[:, :2], full["A_mean"][0][:, :2]) <= RTOL
    assert _rel(got[:, 2:], st["A_mean"][0][:, 2:]) <= 1e-15       # untouched columns
    new = np.arange(D, dtype=float).reshape(D, 1)
    Xs[4].qmu = new
    assert np.array_equal(Xs[4].qmu, new)


def test_pca_example_through_network_learn():
    """examples/PCA_missing_data.py: net.addnode(W); net.fetch_network(); net.learn(n) on pyvb_amd, against the
    reference's recorded outputs."""
    import glob
    import pyvb_amd
    from test_pca_oracle_golden import load_pca
    G = _golden_module()
    for path in [p for p in sorted(glob.glob(os.path.join(HERE, "golden", "pca_*.npz"))) if "default_init" not in p][:2]:
        N, d, q, init, pri, z = load_pca(path)
        g = G.pca_build_graph(pyvb_amd, init, pri)
        net = g["net"]
        done = 0
        for it in [int(i) for i in z["iters"]]:
            net.learn(it - done, tol=-np.inf, verbose=False)
            done = it
            tag = "it%d_" % it
            assert _rel(np.hstack([w.qmu for w in g["Ws"]]), z[tag + "W_mean"]) <= RTOL
            assert _rel(np.hstack([zz.qmu for zz in g["Zs"]]).T, z[tag + "Z"]) <= RTOL
            assert _rel(np.hstack([x.qmu for x in g["Xs"]]).T, z[tag + "X"]) <= RTOL
            assert _rel(g["Mu"].qmu.reshape(-1), z[tag + "Mu_mean"]) <= RTOL
            assert _rel(g["Zs"][3].qcov, z[tag + "Z_cov"]) <= RTOL
            assert _rel(np.stack([np.diag(x.qcov) for x in g["Xs"]]), z[tag + "X_var"]) <= RTOL
            assert abs(g["Beta"].qb - float(z[tag + "beta_b"])) <= RTOL * abs(float(z[tag + "beta_b"]))
            ref = z[tag + "elbo_parts"].sum()
            assert abs(net.llb - ref) <= RTOL * abs(ref)
            # the accessor the example prints (PCA_missing_data.py:92)
            assert abs(g["Beta"].pass_down_Ex()[0, 0] - float(z[tag + "beta_a"]) / float(z[tag + "beta_b"])) <= 1e-8 * g["Beta"].pass_down_Ex()[0, 0]


def test_integration_stub_runs():
    """The ctypes stub of INTEGRATION.md section 2 (what a maintainer of the reference would add), executed as printed:
    node objects in, posteriors and the lower bound out, compared with the oracle."""
    import os
    import re
    from pyvb_amd import nodes, synth, _capi
    from oracle import lds_closed_form as O
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    text = open(os.path.join(repo, "INTEGRATION.md")).read()
    blocks = re.findall(r"```python\n(.*?)```", text, flags=re.S)
    stub = [b for b in blocks if "def lds_iterate" in b]
    assert len(stub) == 1
    ns = {}
    exec(stub[0].replace('"libpyvb_hip.so"', repr(_capi.LIB_PATH)), ns)
    T, D, K = 60, 3, 4
    Y, st0, pri = synth.make_problem(T, D, K, 1, 77)
    g = _golden_module().build_graph(nodes, Y[0], pri, st0)     # stands for a graph built from the reference's classes
    total = ns["lds_iterate"](g["Xs"], g["Ys"], g["As"], g["Cs"], g["Q"], g["R"], 3)
    st = O.expand_state(st0, pri, T)
    for _ in range(3):
        parts = O.iterate(st, pri, Y)
    assert abs(total - parts.sum()) <= RTOL * abs(parts.sum())
    assert _rel(np.hstack([x.__dict__["_h_qmu"] for x in g["Xs"]]).T, st["X"][0]) <= RTOL


def test_pca_demo_of_the_reference_tests_file():
    """src/tests.py:289-344 `PCA()`: complete data, the constructors' individual random covariances for the Z_n, update
    order Ws, Mu, Zs, Beta.  Runs on the fused PCA kernels (the mean of the initial Z covariances is all that is read
    before their first update) and agrees with the same script node by node on the generic plan."""
    from pyvb_amd import nodes, generic, _recognise
    rng = np.random.default_rng(8)
    q, d, N = 2, 3, 40
    X = rng.standard_normal((N, q)) @ (rng.standard_normal((d, q)) * 3).T + rng.standard_normal(d) + rng.standard_normal((N, d)) * 0.1

    def script(force_generic):
        np.random.seed(5)               # the constructors draw their initial posteriors from the global stream (Q11)
        Ws = [nodes.Gaussian(d, np.zeros((d, 1)), np.eye(d) * 1e-3) for _ in range(q)]
        W = nodes.hstack(Ws)
        Mu = nodes.Gaussian(d, np.zeros((d, 1)), np.eye(d) * 1e-3)
        Beta = nodes.Gamma(d, 1e-3, 1e-3)
        Zs = [nodes.Gaussian(q, np.zeros((q, 1)), np.eye(q)) for _ in range(N)]
        Xs = [nodes.Gaussian(d, W * z + Mu, Beta) for z in Zs]
        [x.observe(v.reshape(d, 1)) for x, v in zip(Xs, X)]
        if force_generic:
            generic.GenericPlan(W)
        for it in range(4):
            [w.update() for w in Ws]
            Mu.update()
            [z.update() for z in Zs]
            Beta.update()
        return np.hstack([w.qmu for w in Ws]), Mu.qmu, np.hstack([z.qmu for z in Zs]), Zs[3].qcov, float(Beta.qb), W._plan

    ref = script(True)
    got = script(False)
    assert isinstance(got[5], _recognise.PCAPlan) and isinstance(ref[5], generic.GenericPlan)
    for a, b, what in zip(got[:5], ref[:5], ("W", "Mu", "Z", "cov of a Z_n", "Beta.qb")):
        assert _rel(a, b) <= 1e-8, what


def _small_lds(seed, T=12, q=3, d=4):
    from pyvb_amd import nodes
    np.random.seed(seed)                    # the nodes draw their initial posteriors from the global RNG
    Y = np.random.randn(T, d)
    As = [nodes.Gaussian(q, np.zeros((q, 1)), np.eye(q) * 1e-3) for _ in range(q)]
    Cs = [nodes.Gaussian(d, np.zeros((d, 1)), np.eye(d) * 1e-3) for _ in range(q)]
    A, C = nodes.hstack(As), nodes.hstack(Cs)
    Q = nodes.DiagonalGamma(q, np.ones(q) * 1e-3, np.ones(q) * 1e-3)
    R = nodes.DiagonalGamma(d, np.ones(d) * 1e-3, np.ones(d) * 1e-3)
    Xs = [nodes.Gaussian(q, np.zeros((q, 1)), np.eye(q))]
    Ys = [nodes.Gaussian(d, C * Xs[0], R)]
    for t in range(1, T):
        Xs.append(nodes.Gaussian(q, A * Xs[-1], Q)); Ys.append(nodes.Gaussian(d, C * Xs[-1], R))
    for y, row in zip(Ys, Y):
        y.observe(row.reshape(d, 1))

    def loop():
        [x.update() for x in Xs]; [x.update() for x in reversed(Xs)]
        [a.update() for a in As]; [c.update() for c in Cs]; Q.update(); R.update()
    return Xs, As, Cs, Q, R, loop


def test_graph_returns_to_the_fused_plan_when_the_loop_resumes(monkeypatch):
    """A single X_t.update() hands the graph to the node-by-node plan (the fused kernels serve whole sweeps); when a forward
    and a backward sweep follow each other again it goes back.  Checked against a twin kept node by node throughout."""
    from pyvb_amd import generic, _recognise
    from pyvb_amd._recognise import LDSPlan
    Xs, As, Cs, Q, R, loop = _small_lds(5)
    bind = _recognise.bind
    monkeypatch.setattr(_recognise, "bind", lambda node: generic.GenericPlan(node))
    Xs2, As2, Cs2, Q2, R2, loop2 = _small_lds(5)
    loop2(); Xs2[3].update(); Xs2[7].update(); As2[1].update(); loop2(); loop2()
    ref = [x.qmu.copy() for x in Xs2] + [a.qmu.copy() for a in As2] + [np.asarray(Q2.qb).copy(), Xs2[4].qcov.copy()]
    monkeypatch.setattr(_recognise, "bind", bind)

    loop()
    assert isinstance(Xs[0]._plan, LDSPlan) and Xs[0]._plan.resume_left == 3
    Xs[3].update(); Xs[7].update(); As[1].update()
    Xs[0]._plan.flush()                      # requests are queued until something is read
    assert isinstance(Xs[0]._plan, generic.GenericPlan) and Xs[0]._plan.resume is not None
    loop()                                   # forward + backward sweep in a row: back to the fused plan, mid-loop
    assert isinstance(Xs[0]._plan, LDSPlan) and Xs[0]._plan.resume_left == 2
    loop()
    got = [x.qmu.copy() for x in Xs] + [a.qmu.copy() for a in As] + [np.asarray(Q.qb).copy(), Xs[4].qcov.copy()]
    for g, r in zip(got, ref):
        assert _rel(g, r) < 1e-9
    # a broken pattern runs the waiting requests node by node, in order
    Xs[0].update(); Xs[1].update(); Xs[5].update()
    Xs[0]._plan.flush()
    assert isinstance(Xs[0]._plan, generic.GenericPlan)
    Xs2_plan = Xs2[0]._plan
    monkeypatch.setattr(_recognise, "bind", lambda node: generic.GenericPlan(node))
    Xs2[0].update(); Xs2[1].update(); Xs2[5].update()
    assert Xs2[0]._plan is Xs2_plan
    for t in (0, 1, 5, 6):
        assert _rel(Xs[t].qmu, Xs2[t].qmu) < 1e-9


def test_pca_rows_as_their_constructors_drew_them():
    """The reference's example never assigns the X_n: a partially observed row messages with the constructor's random mean
    at ALL its entries (and covariance I / rand) until its first update pins the observed ones (gaussian.py:70-72, :90-96,
    :125-134).  Fixture pca_default_init_* records such a run of the reference; here the same graph through Network.learn."""
    import pyvb_amd
    from pyvb_amd._recognise import PCAPlan
    from test_pca_oracle_golden import load_pca
    G = _golden_module()
    N, d, q, init, pri, z = load_pca(os.path.join(HERE, "golden", "pca_default_init_n50_d6_q2.npz"))
    np.random.seed(30103)                                   # the generator's seed: the constructors draw from the global RNG
    g = G.pca_build_graph(pyvb_amd, dict(init), pri, explicit_x=False)
    drawn = np.hstack([x.qmu for x in g["Xs"]]).T
    free = ~init["obs"].all(1)
    assert _rel(drawn[free], init["X_full"][free]) < 1e-15     # same draws as the reference's classes made
    assert _rel(np.array([x.qcov[0, 0] for x in g["Xs"]])[free], init["X_var0"][free]) < 1e-15
    net = g["net"]
    part = free & init["obs"].any(1)
    n0 = int(np.nonzero(part)[0][0])
    assert _rel(g["Xs"][n0].qcov, init["X_var0"][n0] * np.eye(d)) < 1e-15      # before any update: the draw, on all entries
    done = 0
    for it in [int(i) for i in z["iters"]]:
        net.learn(it - done, tol=-np.inf, verbose=False)
        done = it
        tag = "it%d_" % it
        assert isinstance(g["Zs"][0]._plan, PCAPlan)
        assert _rel(np.hstack([w.qmu for w in g["Ws"]]), z[tag + "W_mean"]) <= RTOL
        assert _rel(np.hstack([zz.qmu for zz in g["Zs"]]).T, z[tag + "Z"]) <= RTOL
        assert _rel(np.hstack([x.qmu for x in g["Xs"]]).T, z[tag + "X"]) <= RTOL
        assert _rel(np.stack([np.diag(x.qcov) for x in g["Xs"]]), z[tag + "X_var"]) <= RTOL
        assert _rel(g["Mu"].qmu.reshape(-1), z[tag + "Mu_mean"]) <= RTOL
        assert abs(g["Beta"].qb - float(z[tag + "beta_b"])) <= RTOL * abs(float(z[tag + "beta_b"]))
        ref = z[tag + "elbo_parts"].sum()
        assert abs(net.llb - ref) <= RTOL * abs(ref)


def test_assignment_wins_whatever_plan_the_queue_leaves_the_graph_with(monkeypatch):
    """A re-observation while requests are still queued: flushing them hands the graph to the node-by-node plan (a lone
    X_t.update()), the sweeps among them bring the fused plan back -- bound from the old observation -- and further lone updates
    hand it over again.  The new observation must be what every later result is computed from (found by profiles/fuzz_ops.py)."""
    from pyvb_amd import generic, _recognise

    def run(Xs, As, Cs, Q, R, loop, Ys):
        Xs[2].update(); [x.update() for x in reversed(Xs)]; loop(); Xs[3].update(); R.update()
        Ys[5].observe(np.full((4, 1), 0.25))            # everything above is still queued on a fused plan
        loop()
        Ys[6].observe(np.full((4, 1), -0.5)); Xs[1].qmu = np.ones((3, 1))
        loop()
        return [x.qmu.copy() for x in Xs] + [np.asarray(R.qb).copy(), Ys[5].qmu.copy(), Ys[6].qmu.copy()]

    def graph():
        Xs, As, Cs, Q, R, loop = _small_lds(11)
        Ys = [x.children[-1].children[0] if t == len(Xs) - 1 else [c for c in x.children if c.A.shape[0] == 4][0].children[0] for t, x in enumerate(Xs)]
        return Xs, As, Cs, Q, R, loop, Ys
    got = run(*graph())
    monkeypatch.setattr(_recognise, "bind", lambda node: generic.GenericPlan(node))
    ref = run(*graph())
    for g, r in zip(got, ref):
        assert _rel(g, r) < 1e-9
    assert _rel(got[-2], np.full((4, 1), 0.25)) == 0.0 and _rel(got[-1], np.full((4, 1), -0.5)) == 0.0


def test_assigning_a_column_mean_keeps_the_dense_covariances_with_wishart_noise(monkeypatch):
    """With Wishart noise the columns of A and C have dense covariances; assigning one column's mean must not touch them
    (the write-back used to go through the diagonal form; found by profiles/fuzz_ops.py)."""
    from pyvb_amd import nodes, generic, _recognise

    def run():
        np.random.seed(3)
        T, q, d = 10, 3, 2
        Y = np.random.randn(T, d)
        As = [nodes.Gaussian(q, np.zeros((q, 1)), np.eye(q) * 1e-3) for _ in range(q)]
        Cs = [nodes.Gaussian(d, np.zeros((d, 1)), np.eye(d) * 1e-3) for _ in range(q)]
        A, C = nodes.hstack(As), nodes.hstack(Cs)
        Q, R = nodes.Wishart(q, q + 1e-3, np.eye(q) * 1e-3), nodes.Wishart(d, d + 1e-3, np.eye(d) * 1e-3)
        Q.qw, R.qw = np.eye(q) * 0.7, np.eye(d) * 0.9
        Xs = [nodes.Gaussian(q, np.zeros((q, 1)), np.eye(q))]
        Ys = [nodes.Gaussian(d, C * Xs[0], R)]
        for t in range(1, T):
            Xs.append(nodes.Gaussian(q, A * Xs[-1], Q)); Ys.append(nodes.Gaussian(d, C * Xs[-1], R))
        for y, row in zip(Ys, Y):
            y.observe(row.reshape(d, 1))
        for _ in range(2):
            [x.update() for x in Xs]; [x.update() for x in reversed(Xs)]; [a.update() for a in As]; [c.update() for c in Cs]; Q.update(); R.update()
        As[0].qmu = np.array([[0.3], [-0.2], [0.1]])
        As[1].update(); Q.update()
        return [As[2].qcov.copy(), As[1].qmu.copy(), np.asarray(Q.qw).copy()], Xs[0]._plan
    got, plan = run()
    assert isinstance(plan, _recognise.LDSPlan)
    assert np.abs(got[0] - np.diag(np.diag(got[0]))).max() > 0.0           # still dense
    monkeypatch.setattr(_recognise, "bind", lambda node: generic.GenericPlan(node))
    ref, _ = run()
    for g, r in zip(got, ref):
        assert _rel(g, r) < 1e-9


@pytest.mark.parametrize("name", ["example_script_q2d5_t40", "knowns_script_q2d5_t40"])
def test_the_example_exactly_as_the_reference_writes_it(name):
    """examples/Linear_Dynamic_System.py:46-77 with nothing assigned: every initial posterior is what the constructors draw
    from numpy's global generator.  Under the same seed pyvb_amd's classes make the same draws, and the fused plan
    reproduces the reference's run (fixture script_example_script_*: recorded from the reference's classes)."""
    import pyvb_amd
    from pyvb_amd._recognise import LDSPlan
    G = _golden_module()
    z = dict(np.load(os.path.join(HERE, "golden", "script_%s.npz" % name), allow_pickle=False))
    np.random.seed(int(z["seed"]))
    g = G.example_script_graph(pyvb_amd.nodes, z["Y"], int(z["q"]), bool(z["knowns"]))
    assert _rel(np.hstack([x.qmu for x in g["Xs"]]).T, z["init_X"]) < 1e-15
    assert _rel(np.hstack([a.qmu for a in g["As"]]), z["init_A"]) < 1e-15 and _rel(g["Q"].qb, z["init_Qb"]) < 1e-15
    done = 0
    for it in [int(i) for i in z["iters"]]:
        for _ in range(it - done):
            G.example_script_loop(g)
        done = it
        tag = "it%d_" % it
        assert isinstance(g["Xs"][0]._plan, LDSPlan)
        assert _rel(np.hstack([x.qmu for x in g["Xs"]]).T, z[tag + "X"]) <= RTOL
        assert _rel(g["A"].pass_down_Ex(), z[tag + "A"]) <= RTOL and _rel(g["C"].pass_down_Ex(), z[tag + "C"]) <= RTOL
        assert _rel(g["Q"].qb, z[tag + "Qb"]) <= RTOL and _rel(g["R"].qb, z[tag + "Rb"]) <= RTOL
        assert _rel(g["Xs"][1].qcov, z[tag + "Sigma1"]) <= RTOL
        llb = sum(float(n.log_lower_bound()) for n in g["Xs"] + g["Ys"] + g["As"] + g["Cs"] + [g["Q"], g["R"]])
        assert abs(llb - float(z[tag + "llb"])) <= 1e-7 * abs(float(z[tag + "llb"]))


@pytest.mark.parametrize("D,K,fused", [(66, 3, True), (130, 2, False)])
def test_shapes_beyond_64(D, K, fused):
    """D = 66: the recogniser binds the graph to the fused plan's second shape class (k_big.hip, D, K <= 128); D = 130: no fused
    kernel takes it, the graph runs node by node on the generic plan (no shape limit there).  Both against the closed-form
    oracle, which has none either."""
    from pyvb_amd import synth, generic, _recognise
    from oracle import lds_closed_form as O
    G = _golden_module()
    T = 3
    Y, st0, pri = synth.make_problem(T, D, K, 1, seed=9)
    g = G.build_graph(__import__("pyvb_amd").nodes, Y[0], pri, st0)
    st = O.expand_state(st0, pri, T)
    for _ in range(1 if D > 128 else 2):        # (at D = 130 an iteration is 10^9 flops of node-by-node records: one of them)
        [x.update() for x in g["Xs"]]; [x.update() for x in reversed(g["Xs"])]
        [a.update() for a in g["As"]]; [c.update() for c in g["Cs"]]; g["Q"].update(); g["R"].update()
        O.iterate(st, pri, Y, with_elbo=False)
    assert isinstance(g["Xs"][0]._plan, _recognise.LDSPlan if fused else generic.GenericPlan)
    assert _rel(np.hstack([x.qmu for x in g["Xs"]]).T, st["X"][0]) < 1e-7
    assert _rel(np.hstack([a.qmu for a in g["As"]]), st["A_mean"][0]) < 1e-7
    assert _rel(g["R"].qb, st["R_b"][0]) < 1e-7


def test_the_pca_example_exactly_as_the_reference_writes_it():
    """examples/PCA_missing_data.py:31-45 with nothing assigned: W, Mu, every Z_n and X_n as their constructors draw them (each Z_n
    its own covariance I / rand -- the fused plan keeps their mean, which serves exactly because everything that reads them
    before their first update is linear in them), Network.fetch_network + learn.  Fixture recorded from the reference's classes."""
    import pyvb_amd
    from pyvb_amd._recognise import PCAPlan
    G = _golden_module()
    z = dict(np.load(os.path.join(HERE, "golden", "script_pca_script_n40_d5_q2.npz"), allow_pickle=False))
    np.random.seed(int(z["seed"]))
    g = G.pca_script_graph(pyvb_amd, z["X"], int(z["q"]))
    assert _rel(np.hstack([zz.qmu for zz in g["Zs"]]).T, z["init_Z"]) < 1e-15
    assert _rel(np.array([zz.qcov[0, 0] for zz in g["Zs"]]), z["init_Zc"]) < 1e-15
    net, done = g["net"], 0
    for it in [int(i) for i in z["iters"]]:
        net.learn(it - done, tol=-np.inf, verbose=False)
        done = it
        tag = "it%d_" % it
        assert isinstance(g["Zs"][0]._plan, PCAPlan)
        assert _rel(np.hstack([w.qmu for w in g["Ws"]]), z[tag + "W"]) <= RTOL
        assert _rel(np.hstack([zz.qmu for zz in g["Zs"]]).T, z[tag + "Z"]) <= RTOL
        assert _rel(np.hstack([x.qmu for x in g["Xs"]]).T, z[tag + "Xm"]) <= RTOL
        assert _rel(g["Mu"].qmu.reshape(-1), z[tag + "Mu"]) <= RTOL
        assert abs(g["Beta"].qb - float(z[tag + "beta_b"])) <= RTOL * abs(float(z[tag + "beta_b"]))
        assert abs(net.llb - float(z[tag + "llb"])) <= 1e-7 * abs(float(z[tag + "llb"]))


def test_network_learn_over_several_unconnected_graphs():
    """Network.learn (network.py:40-56) iterates any node list.  One list holding an LDS graph (fused kernels), a VB-PCA
    graph (fused kernels) and a small graph without a fused plan (node by node): every posterior and the summed bound equal
    what three separate networks give, and the plans stay what the recogniser chose."""
    import glob
    import pyvb_amd
    from pyvb_amd import nodes, synth, _recognise, generic
    from pyvb_amd.network import Network
    from test_pca_oracle_golden import load_pca
    G = _golden_module()
    T, D, K = 40, 3, 4
    Y, st0, pri = synth.make_problem(T, D, K, 1, 91)
    path = sorted(p for p in glob.glob(os.path.join(HERE, "golden", "pca_*.npz")) if "default_init" not in p)[0]
    _, _, _, init, ppri, _ = load_pca(path)

    def small(rng):
        y = rng.standard_normal(12) * 0.3 + 2.0
        mu = nodes.Gaussian(1, np.array([[0.0]]), np.array([[1e-3]]))
        lam = nodes.Gamma(1, 1e-3, 1e-3)
        ys = [nodes.Gaussian(1, mu, lam) for _ in y]
        for n, v in zip(ys, y):
            n.observe(np.array([[v]]))
        mu.qmu, mu.qcov, lam.qb = np.array([[0.5]]), np.array([[2.0]]), 0.7
        return [mu, lam] + ys

    def build():
        g = G.build_graph(nodes, Y[0], pri, st0)
        # states in chain order, then the parameters: one forward sweep per learn() iteration, which the fused kernels serve
        # (the crawl order of fetch_network is not a sweep: scenario lds_network_crawl covers that, node by node)
        lds_list = g["Xs"] + g["As"] + g["Cs"] + [g["Q"], g["R"]] + g["Ys"]
        pca = G.pca_build_graph(pyvb_amd, init, ppri)
        sm = small(np.random.default_rng(5))
        return g, lds_list, pca, pca["net"].nodes, sm

    iters = 3
    g, lds_nodes, pca, pca_nodes, sm = build()
    parts = []
    for lst in (lds_nodes, pca_nodes, sm):
        net = Network(lst)
        net.learn(iters, tol=-np.inf, verbose=False)
        parts.append(net.llb)
    want = (np.hstack([x.qmu for x in g["Xs"]]), np.hstack([w.qmu for w in pca["Ws"]]), float(sm[0].qmu[0, 0]), float(sm[1].qb))
    g2, lds_nodes, pca2, pca_nodes, sm2 = build()
    mixed = Network(sm2[:1] + lds_nodes + pca_nodes + sm2[1:])          # the small graph's nodes on both sides of the others
    mixed.learn(iters, tol=-np.inf, verbose=False)
    assert isinstance(g2["Xs"][0]._plan, _recognise.LDSPlan) and isinstance(pca2["Ws"][0]._plan, _recognise.PCAPlan)
    assert isinstance(sm2[0]._plan, generic.GenericPlan)
    assert abs(mixed.llb - sum(parts)) <= 1e-9 * abs(sum(parts))
    assert _rel(np.hstack([x.qmu for x in g2["Xs"]]), want[0]) <= 1e-12 and _rel(np.hstack([w.qmu for w in pca2["Ws"]]), want[1]) <= 1e-12
    assert abs(float(sm2[0].qmu[0, 0]) - want[2]) <= 1e-12 * abs(want[2]) and abs(float(sm2[1].qb) - want[3]) <= 1e-12 * want[3]
    # a list that holds only PART of a fused graph's random nodes: the reference sums the listed nodes' terms
    g3, lds_nodes, _, _, _ = build()
    part = Network([n for n in lds_nodes if n is not g3["Q"]])
    part.learn(2, tol=-np.inf, verbose=False)
    iterable = [n for n in lds_nodes if isinstance(n, (nodes.Gaussian, nodes.DiagonalGamma)) and n is not g3["Q"]]
    assert abs(part.llb - sum(float(n.log_lower_bound()) for n in iterable)) <= 1e-9 * abs(part.llb)


def test_pca_q_ln_det_comes_from_the_device():
    """pyvb_pca_get_qld: the q_ln_det of W's columns, the Z_n, Mu and the rows without observations as the device's updates
    left them (quirk Q1 form, gaussian.py:120) -- equal to the formula evaluated on the fetched covariances, NaN before the
    first update -- and PCAPlan._sync_host uses them instead of factorising on the host."""
    import inspect
    from pyvb_amd import _recognise
    from pyvb_amd.pca import PCABatch
    rng = np.random.default_rng(3)
    N, d, q = 60, 7, 3
    X = rng.standard_normal((N, q)) @ rng.standard_normal((q, d)) + 0.1 * rng.standard_normal((N, d))
    obs = rng.random((N, d)) > 0.2
    obs[5] = False
    obs[17] = False
    init = {"obs": obs, "X": np.where(obs, X, 0.0), "W_mean": rng.standard_normal((d, q)), "W_var": np.ones((q, d)),
            "Z": rng.standard_normal((N, q)), "Z_cov": np.eye(q), "Mu_mean": np.zeros(d), "Mu_var": np.ones(d), "beta_b": 1.0}
    pri = {"W_prior_mean": np.zeros((d, q)), "W_prior_prec": np.full((q, d), 1e-3), "Mu_prior_mean": np.zeros(d),
           "Mu_prior_prec": np.full(d, 1e-3), "beta_a0": 1e-3, "beta_b0": 1e-3}
    b = PCABatch.from_problem(init, pri)
    q0 = b.get_qld()
    assert np.all(np.isnan(q0["W"])) and np.isnan(q0["Z"]) and np.isnan(q0["Mu"])
    b.iterate(2)
    st, ql = b.get_state(), b.get_qld()
    b.close()
    qld = lambda cov: 0.5 / np.sum(np.log(np.diag(np.linalg.cholesky(np.linalg.inv(cov)))))
    for i in range(q):
        assert abs(ql["W"][i] - qld(np.diag(st["W_var"][i]))) <= 1e-10 * abs(ql["W"][i])
    assert abs(ql["Z"] - qld(st["Z_cov"])) <= 1e-10 * abs(ql["Z"]) and abs(ql["Mu"] - qld(np.diag(st["Mu_var"]))) <= 1e-10 * abs(ql["Mu"])
    none = ~obs.any(1)
    assert np.array_equal(np.isfinite(ql["X"]), none)
    assert np.allclose(ql["X"][none], 0.5 / (0.5 * d * np.log(1.0 / st["X_rowvar"][none])), rtol=1e-12)
    src = inspect.getsource(_recognise.PCAPlan._sync_host)
    assert "cholesky" not in src and "linalg" not in src
