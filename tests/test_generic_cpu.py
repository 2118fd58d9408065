"""Host logic of the generic (node-by-node) path, without a GPU: the emitters of pyvb_amd/generic.py build tapes for
the graphs of tests/golden/generic_scenarios.py, the numpy restatement of the tape interpreter (oracle/tape_ref.py) runs
them, and the results are compared with what the REFERENCE's classes produced on the same graphs
(tests/golden/generic_*.npz, from tests/golden/make_golden.py).  This pins both the emitters and the oracle interpreter;
tests/test_generic_gpu.py then runs the same tapes through the HIP interpreter.
"""
import os
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))
import generic_scenarios as GS  # noqa: E402

RTOL = 1e-9


@pytest.fixture
def numpy_executor(monkeypatch):
    """The numpy restatement of the tape interpreter behind GenericPlan, and every graph on the generic plan (an LDS with
    missing outputs would otherwise bind to the fused kernels, which need the GPU)."""
    from oracle.tape_ref import NumpyExecutor
    from pyvb_amd import generic, _recognise
    monkeypatch.setattr(_recognise, "bind", lambda node: generic.GenericPlan(node, executor_factory=NumpyExecutor))


def _close(a, b, what, rtol=RTOL):
    a, b = np.asarray(a, dtype=float), np.asarray(b, dtype=float)
    assert a.shape == b.shape or a.size == b.size, "%s: shapes %r vs %r" % (what, a.shape, b.shape)
    err = np.abs(a.reshape(-1) - b.reshape(-1)).max() / max(np.abs(b).max(), 1e-300)
    assert err <= rtol, "%s: rel err %.3e" % (what, err)


def check_scenario(name, rtol=RTOL, llb_rtol=1e-8):
    """Shared with the GPU test: run scenario `name` through pyvb_amd.nodes and compare with the reference fixture."""
    from pyvb_amd import nodes
    build, seed, checkpoints, messages = GS.SCENARIOS[name]
    z = dict(np.load(os.path.join(HERE, "golden", "generic_%s.npz" % name), allow_pickle=False))
    order, named = build(nodes, np.random.default_rng(seed))
    for it in range(1, max(checkpoints) + 1):
        for n in order:
            n.update()
        if it not in checkpoints:
            continue
        tag = "it%d." % it
        for k, v in GS.snapshot(named).items():
            if k.endswith(".qcov") and np.abs(z[tag + k]).max() == 0.0:
                assert np.abs(v).max() == 0.0, k
                continue
            _close(v, z[tag + k], "%s %s%s" % (name, tag, k), rtol)
        got = GS.lower_bounds(named)
        for k, v in got.items():
            ref = z[tag + k]
            if np.isnan(ref):           # the reference raises there: Wishart parents (ours is derived, finite),
                assert np.isfinite(v) or name != "wishart_precision", k      # or a node that was never updated (no q_ln_det)
                continue
            # quirk Q1 makes single terms ill-conditioned (0.5 / a sum of logs): compare on the scale of the node's terms
            scale = max(abs(ref), 1.0)
            assert abs(v - ref) <= llb_rtol * scale * 10, "%s %s%s: %r vs %r" % (name, tag, k, v, ref)
        for a, b in messages:
            m = named[a].pass_up_m1_m2(named[b])
            _close(m[0], z[tag + "msg.%s.%s.m1" % (a, b)], "message m1 %s->%s" % (a, b), rtol)
            _close(m[1], z[tag + "msg.%s.%s.m2" % (a, b)], "message m2 %s->%s" % (a, b), rtol)
    return named


def check_expectations(rtol=RTOL):
    """Shared with the GPU test: pass_down_Ex / pass_down_ExxT of the operation nodes of GS.multiplication_expectations
    against tests/golden/expect_multiplication.npz (generated from the reference's classes)."""
    from pyvb_amd import nodes
    z = dict(np.load(os.path.join(HERE, "golden", "expect_multiplication.npz"), allow_pickle=False))
    order, named, ops = GS.multiplication_expectations(nodes, np.random.default_rng(GS.EXPECTATION_SEED))
    seen = 0
    for it in range(max(GS.EXPECTATION_ITERS) + 1):
        if it:
            for n in order:
                n.update()
        if it not in GS.EXPECTATION_ITERS:
            continue
        for k, v in GS.expectations(ops).items():
            _close(v, z["it%d.%s" % (it, k)], "iteration %d %s" % (it, k), rtol)
            seen += 1
        for k, v in GS.snapshot(named).items():
            if np.abs(z["it%d.%s" % (it, k)]).max() > 0:
                _close(v, z["it%d.%s" % (it, k)], "iteration %d %s" % (it, k), rtol)
    assert seen == 3 * 2 * len(ops)
    assert isinstance(ops["row_vec"].pass_down_ExxT(), float)           # np.trace in the reference: a scalar, not a 1 x 1 array


def test_multiplication_expectations_against_the_reference(numpy_executor):
    check_expectations()


@pytest.mark.parametrize("name", sorted(n for n in GS.SCENARIOS if n not in GS.NEEDS_DEVICE))
def test_generic_scenarios_against_reference(name, numpy_executor):
    check_scenario(name)


def test_network_learn_on_a_generic_graph(numpy_executor):
    """Network.learn (network.py:40-56) on a graph without a fused plan: fetch_network, updates in list order, the sum
    of all log_lower_bound() terms, the convergence predicate."""
    from pyvb_amd import nodes
    from pyvb_amd.network import Network
    build, seed, _, _ = GS.SCENARIOS["mean_and_variance_inference"]
    order, named = build(nodes, np.random.default_rng(seed))
    net = Network([named["mu"]])
    net.fetch_network(verbose=False)
    assert len([n for n in net.nodes if isinstance(n, nodes.Gaussian)]) == 31
    net.learn(6, tol=-np.inf, verbose=False)
    z = dict(np.load(os.path.join(HERE, "golden", "generic_mean_and_variance_inference.npz"), allow_pickle=False))
    # the crawl puts mu first, then its children, then prec: same update order as the scenario (observed nodes do not update)
    _close(named["mu"].qmu, z["it6.mu.qmu"], "mu after Network.learn")
    _close(named["prec"].qb, z["it6.prec.qb"], "prec after Network.learn")
    ref_llb = sum(float(z[k]) for k in z if k.startswith("it6.") and k.endswith(".llb"))
    assert abs(net.llb - ref_llb) <= 1e-8 * abs(ref_llb)


def test_graph_growing_after_the_first_update(numpy_executor):
    """src/tests.py:255-263 updates nodes while the chain is still being built: a generic plan bound to the old graph is
    stale once a node gains a child and is rebuilt (state carried over) at the next use."""
    from pyvb_amd import nodes
    rng = np.random.default_rng(3)
    mu = nodes.Gaussian(2, np.zeros((2, 1)), np.eye(2) * 1e-2)
    mu.qmu, mu.qcov = rng.standard_normal((2, 1)), np.eye(2)
    y0 = nodes.Gaussian(2, mu, np.eye(2) * 4.0)
    y0.observe(np.array([[1.0], [2.0]]))
    mu.update()
    first = mu.qmu.copy()
    _close(first, np.linalg.solve(np.eye(2) * 4.01, 4.0 * np.array([[1.0], [2.0]])), "one child")
    y1 = nodes.Gaussian(2, mu, np.eye(2) * 4.0)
    y1.observe(np.array([[3.0], [0.0]]))
    _close(mu.qmu, first, "state survives the re-bind")
    mu.update()
    _close(mu.qmu, np.linalg.solve(np.eye(2) * 8.01, 4.0 * np.array([[4.0], [2.0]])), "two children")


def test_observing_a_node_of_a_bound_graph(numpy_executor):
    """observe() on a node whose graph already runs on the device: the observation must survive the re-bind (the plan's
    state is pulled back into the nodes before the new value is stored)."""
    from pyvb_amd import nodes
    rng = np.random.default_rng(4)
    mu = nodes.Gaussian(2, np.zeros((2, 1)), np.eye(2) * 1e-2)
    ys = [nodes.Gaussian(2, mu, np.eye(2) * 4.0) for _ in range(3)]
    for n in [mu] + ys:
        n.qmu, n.qcov = rng.standard_normal((2, 1)), np.eye(2)
    ys[0].observe(np.array([[1.0], [2.0]]))
    mu.update()
    ys[1].update()                                     # a latent child: follows its parent
    _close(ys[1].qmu, mu.qmu, "latent child")
    ys[1].observe(np.array([[5.0], [-1.0]]))            # ... and is observed afterwards
    _close(ys[1].qmu, np.array([[5.0], [-1.0]]), "the observation is what the node holds")
    assert np.abs(ys[1].qcov).max() == 0.0
    mu.update()
    P = np.eye(2) * (1e-2 + 12.0)
    w = 4.0 * (np.array([[1.0], [2.0]]) + np.array([[5.0], [-1.0]]) + ys[2].qmu)
    _close(mu.qmu, np.linalg.solve(P, w), "three children, two of them observed")


def test_constant_matrix_lds(numpy_executor):
    """The constant-parameter LDS of src/tests.py:223-286 (Constant A, C, Q, R).  The reference is numerically wrong
    there (Multiplication.pass_up_m1_m2 returns a bare matrix for a Constant left operand and
    Gaussian.update mis-indexes it); the build implements the intended message (A^T L A, A^T m2), checked here against
    the textbook mean-field update written out in numpy."""
    from pyvb_amd import nodes
    rng = np.random.default_rng(5)
    T = 12
    A = np.array([[0.9, 0.1], [-0.2, 0.8]])
    C = rng.standard_normal((1, 2))
    Qi, Ri = np.diag([4.0, 2.0]), np.array([[25.0]])
    Y = rng.standard_normal((T, 1))
    An, Cn, Qn, Rn = nodes.Constant(A), nodes.Constant(C), nodes.Constant(Qi), nodes.Constant(Ri)
    Z = [nodes.Gaussian(2, np.zeros((2, 1)), np.eye(2))]
    Ys = [nodes.Gaussian(1, Cn * Z[-1], Rn)]
    for t in range(1, T):
        Z.append(nodes.Gaussian(2, An * Z[-1], Qn))
        Ys.append(nodes.Gaussian(1, Cn * Z[-1], Rn))
    for y, v in zip(Ys, Y):
        y.observe(v.reshape(1, 1))
    mu = rng.standard_normal((T, 2))
    for z, m in zip(Z, mu):
        z.qmu, z.qcov = m.reshape(2, 1).copy(), np.eye(2)
    for sweep in range(2):
        order = list(range(T - 1, -1, -1)) if sweep == 0 else list(range(T))       # :268-273: reverse pass, forward pass
        for t in order:
            Z[t].update()
            P = (np.eye(2) if t == 0 else Qi) + C.T @ Ri @ C + (A.T @ Qi @ A if t < T - 1 else 0.0)
            w = C.T @ Ri @ Y[t].reshape(1, 1) + (Qi @ A @ mu[t - 1].reshape(2, 1) if t > 0 else 0.0) \
                + (A.T @ Qi @ mu[t + 1].reshape(2, 1) if t < T - 1 else 0.0)
            mu[t] = np.linalg.solve(P, w).reshape(-1)
            _close(Z[t].qmu, mu[t].reshape(2, 1), "Z[%d].qmu" % t)
            _close(Z[t].qcov, np.linalg.inv(P), "Z[%d].qcov" % t)


def test_emitters_refuse_what_the_reference_cannot_do(numpy_executor):
    from pyvb_amd import nodes
    mu = nodes.Gaussian(2, np.zeros((2, 1)), np.eye(2))
    with pytest.raises(NotImplementedError):
        nodes.Transpose(mu)
    h = nodes.hstack([mu, nodes.Gaussian(2, np.zeros((2, 1)), np.eye(2))])
    with pytest.raises(NotImplementedError):
        h.pass_down_ExTx()              # nodes_todo.py:40-41 raises too


PROGRAM_CASES = ["simple_PCA", "partial_observations", "lds_missing_outputs", "simple_regression"] + ["random_%d" % s for s in GS.RANDOM_SEEDS[:12]]


def check_programs(name, plan_of):
    """update_all / llb_sum tapes carry a PROGRAM: nodes whose records touch disjoint posteriors run side by side (device) or
    last-first (numpy executor).  Either must equal the one-node-at-a-time loop."""
    from pyvb_amd import nodes
    build, seed, _, _ = GS.SCENARIOS[name]
    order, named = build(nodes, np.random.default_rng(seed))
    order2, named2 = build(nodes, np.random.default_rng(seed))
    for _ in range(2):
        for n in order:
            n.update()
    plans = []                              # a random graph may fall into several unconnected components: one plan each
    for n in order2:
        p = plan_of(n)
        if p not in plans:
            plans.append(p)
    plan = plans[0]
    for _ in range(2):
        for p in plans:                     # components do not see each other: their order does not matter (Network groups by plan too)
            p.update_all([n for n in order2 if n._plan is p])
    for k, v in GS.snapshot(named).items():
        w = GS.snapshot(named2)[k]
        assert np.abs(np.asarray(v) - np.asarray(w)).max() <= 1e-12 * max(np.abs(np.asarray(v)).max(), 1.0), (name, k)
    for p in plans:
        keys = [k for k in named2 if named2[k]._plan is p and not getattr(named2[k], "observed", False)]
        try:
            want = np.array([float(np.asarray(named[k].log_lower_bound()).reshape(-1)[0]) for k in keys])
        except (AttributeError, NotImplementedError):
            continue
        got = p.llb_sum([named2[k] for k in keys])
        ok = np.isfinite(want)
        assert np.all(np.abs(got[ok] - want[ok]) <= 1e-10 * np.maximum(np.abs(want[ok]), 1.0)), (name, got, want)
    return plan


@pytest.mark.parametrize("name", PROGRAM_CASES)
def test_programs_of_many_node_tapes(name, numpy_executor):
    from pyvb_amd import nodes
    plan = check_programs(name, lambda n: nodes._plan_of(n))
    if name == "simple_PCA":        # 25 independent latent scalars: they must have ended up side by side
        progs = [p for p in plan._programs.values() if p is not None]
        assert progs and max(int(l[1]) for p in progs for l in p[1]) >= 25
