"""GPU: M LDS graphs of one structure built side by side through the node API run as ONE LDSBatch(N = M) -- the replicate
axis the headline is measured on, reached from the reference's own API (network.py:46-49 iterates any node list;
examples/Linear_Dynamic_System.py:46-77 per graph).  Same scenarios as tests/test_groups_cpu.py, on libpyvb_hip.so."""
import os

import numpy as np
import pytest

import group_scenarios as S
from conftest import load_golden, GOLDEN_DIR

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _fresh_pool():
    from pyvb_amd import _recognise
    _recognise._pool.clear()


def _fixture_problems():
    out = []
    for name in ("example_d2k5_t200", "example_b_d2k5_t200"):
        meta, Y, st0, pri, z = load_golden(os.path.join(GOLDEN_DIR, "lds_%s.npz" % name))
        out.append((Y, st0, pri, z))
    return out


def test_eight_graphs_share_one_handle_and_match_the_reference():
    """Eight graphs of the example's shape with distinct data and initial posteriors, two of them the data sets the
    reference itself has run (tests/golden/lds_example*_d2k5_t200.npz): one handle of eight replicates, results bitwise
    those of eight separately bound graphs (same time split and statistics chunks at this size: pyvb_lds_create) and equal
    to the reference's for its two."""
    from pyvb_amd import nodes
    from pyvb_amd.lds import LDSBatch
    fx = _fixture_problems()
    pri = fx[0][2]
    probs = [(Y, st0, pri) for Y, st0, _, _ in fx] + S.problems(200, 2, 5, 6, pri=pri)
    graphs = S.build(nodes, probs)
    for it in range(2):
        for g in graphs:
            S.loop_body(g)
    assert all(g["Xs"][0]._plan.group is None for g in graphs)      # nothing has been asked for yet: no handle
    snaps = [S.snapshot(g) for g in graphs]
    grp = graphs[0]["Xs"][0]._plan.group
    assert isinstance(grp.batch, LDSBatch) and grp.batch.N == 8
    assert [m is g["Xs"][0]._plan for m, g in zip(grp.members, graphs)] == [True] * 8
    for k in (0, 1):
        z = fx[k][3]
        for got, key in (("X", "it2_X"), ("A", "it2_A_mean"), ("C", "it2_C_mean"), ("Qb", "it2_Q_b"), ("Rb", "it2_R_b")):
            assert S.rel(snaps[k][got], z[key]) <= 1e-8, (k, key)
        for c in range(3):
            assert S.rel(snaps[k]["S%d" % c], z["it2_Sigma"][c]) <= 1e-8
        parts = graphs[k]["Xs"][0]._plan.elbo_parts()
        assert abs(parts.sum() - z["it2_elbo_parts"].sum()) <= 1e-8 * abs(z["it2_elbo_parts"].sum())
    twins = S.build(nodes, probs)
    for k, g in enumerate(twins):                                   # each bound on its own: read before the next one starts
        for it in range(2):
            S.loop_body(g)
        S.same(S.snapshot(g), snaps[k], exact=True)
        assert g["Xs"][0]._plan.group.batch.N == 1


def test_a_request_the_fused_kernels_do_not_serve_moves_one_graph_only():
    """One graph of eight gets a lone X_t.update(): it alone goes to the node-by-node plan, the other seven stay on the
    handle, and all eight end where eight separately run graphs end."""
    from pyvb_amd import nodes
    from pyvb_amd._recognise import LDSPlan
    probs = S.problems(12, 3, 4, 8)
    graphs, twins = S.build(nodes, probs), S.build(nodes, probs)

    def script(gs, together):
        for g in gs:
            S.loop_body(g)
            together or S.snapshot(g)
        gs[3]["Xs"][5].update()
        got = gs[3]["Xs"][5].qmu
        for g in gs:
            S.loop_body(g)
            together or S.snapshot(g)
        return got

    a = script(graphs, True)
    snaps = [S.snapshot(g) for g in graphs]
    grp = graphs[0]["Xs"][0]._plan.group
    assert grp.batch.N == 8 and len(grp.live()) == 7 and grp.members[3] is None
    assert all(isinstance(g["Xs"][0]._plan, LDSPlan) for g in graphs)
    p3 = graphs[3]["Xs"][0]._plan
    assert p3.group is not grp and p3.group.batch.N == 1 and p3.resume_left == LDSPlan.resume_left - 1
    b = script(twins, False)
    assert S.rel(a, b) <= 1e-10
    for k in range(8):
        S.same(S.snapshot(twins[k]), snaps[k], exact=False, tol=1e-9)


def test_network_learn_over_graphs_of_one_structure(monkeypatch):
    from pyvb_amd import nodes, _recognise
    from pyvb_amd.network import Network
    probs = S.problems(30, 3, 4, 6)
    graphs = S.build(nodes, probs)
    net = Network([n for g in graphs for n in S.all_nodes(g)])
    replayed = []
    run_script = _recognise.LDSGroup.run_script
    monkeypatch.setattr(_recognise.LDSGroup, "run_script", lambda self, sc: replayed.append(list(sc)) or run_script(self, sc))
    net.learn(4, tol=-np.inf, verbose=False)
    assert replayed == [[("F",), ("A", 0, 3), ("C", 0, 3), ("Q",), ("R",)]] * 3
    grp = graphs[0]["Xs"][0]._plan.group
    assert grp.batch.N == 6 and len(grp.live()) == 6
    total = 0.0
    for k in range(6):
        g = S.build(nodes, [probs[k]])[0]
        one = Network(S.all_nodes(g))
        one.learn(4, tol=-np.inf, verbose=False)
        total += one.llb
        S.same(S.snapshot(g), S.snapshot(graphs[k]), exact=True)
    assert abs(net.llb - total) <= 1e-12 * abs(total)
    # a second call over TWO of the six: the four that are not asked keep the handle, the two leave together
    sub = Network([n for g in graphs[:2] for n in S.all_nodes(g)])
    sub.learn(2, tol=-np.inf, verbose=False)
    new = graphs[0]["Xs"][0]._plan.group
    assert new is not grp and new is graphs[1]["Xs"][0]._plan.group and new.batch.N == 2
    assert len(grp.live()) == 4 and graphs[2]["Xs"][0]._plan.group is grp
    for k in (0, 1):
        g = S.build(nodes, [probs[k]])[0]
        one = Network(S.all_nodes(g))
        one.learn(4, tol=-np.inf, verbose=False)
        one.learn(2, tol=-np.inf, verbose=False)
        S.same(S.snapshot(g), S.snapshot(graphs[k]), exact=False, tol=1e-10)
    # a graph that leaves takes the covariances of its states along (pyvb_lds_set_posterior_classes): parameters first is fine
    from pyvb_amd._recognise import LDSPlan
    [a.update() for a in graphs[5]["As"]]
    got = np.hstack([a.qmu for a in graphs[5]["As"]])
    plan = graphs[5]["Xs"][0]._plan
    assert isinstance(plan, LDSPlan) and plan.group is not grp and len(grp.live()) == 3
    g = S.build(nodes, [probs[5]])[0]
    Network(S.all_nodes(g)).learn(4, tol=-np.inf, verbose=False)
    [a.update() for a in g["As"]]
    assert S.rel(got, np.hstack([a.qmu for a in g["As"]])) <= 1e-10


@pytest.mark.parametrize("kind", ["gamma", "wishart", "knowns", "missing"])
def test_other_graph_families_share_a_handle_too(kind):
    """Gamma and Wishart noise, known entries of A / C (examples/LDS_knowns_in_A.py:73-74) and outputs with NaN: three
    graphs on one handle against the same three on their own, the first of them a data set the reference has run."""
    from pyvb_amd import nodes
    name = {"gamma": "gamma_d4k5_t60", "wishart": "wishart_d3k4_t40", "knowns": "knowns_d3k4_t50", "missing": "missing_d3k4_t30"}[kind]
    meta, Y, st0, pri, z = load_golden(os.path.join(GOLDEN_DIR, "lds_%s.npz" % name))
    T, D, K = meta["T"], meta["D"], meta["K"]
    rng = np.random.default_rng(5)
    probs = [(Y, st0, pri)]
    for k in range(2):
        Y2 = Y * (1.0 + 0.1 * rng.standard_normal(Y.shape))        # NaN stays NaN: the same outputs are unobserved
        st2 = {key: v.copy() for key, v in st0.items()}
        st2["X"] = rng.standard_normal(st0["X"].shape)
        st2["A_mean"] = rng.standard_normal(st0["A_mean"].shape)
        probs.append((Y2, st2, pri))
    G = S.golden_module()

    def make():
        gs = S.build(nodes, probs)
        if kind == "missing":
            for g, (Yk, stk, _) in zip(gs, probs):
                for t, y in enumerate(g["Ys"]):
                    if not y.observed:
                        y.qmu = stk["Yq"][0, t].reshape(-1, 1).copy()
                        y.qcov = np.eye(K) * stk["Yrowvar"][0, t]
        return gs

    def body(g):
        Xs = g["Xs"]
        [x.update() for x in Xs]
        Xs.reverse(); [x.update() for x in Xs]; Xs.reverse()
        if kind == "missing":
            [y.update() for y in g["Ys"] if not y.observed]
        [a.update() for a in g["As"]]
        [c.update() for c in g["Cs"]]
        g["Q"].update(); g["R"].update()

    def read(g):
        out = {"X": np.hstack([x.qmu for x in g["Xs"]]).T, "A": np.hstack([a.qmu for a in g["As"]]), "C": np.hstack([c.qmu for c in g["Cs"]]),
               "Acov": np.stack([a.qcov for a in g["As"]]), "S0": g["Xs"][0].qcov}
        if kind == "wishart":
            out["Qw"], out["Rw"] = g["Q"].qw, g["R"].qw
        else:
            out["Qb"], out["Rb"] = np.asarray(g["Q"].qb, dtype=float), np.asarray(g["R"].qb, dtype=float)
        if kind == "missing":
            out["Yq"] = np.hstack([y.qmu for y in g["Ys"]]).T
        return out

    graphs = make()
    for g in graphs:
        body(g)
    got = [read(g) for g in graphs]
    grp = graphs[0]["Xs"][0]._plan.group
    assert grp.batch.N == 3 and len(grp.live()) == 3
    assert S.rel(got[0]["X"], z["it1_X"]) <= 1e-8 and S.rel(got[0]["A"], z["it1_A_mean"]) <= 1e-8 and S.rel(got[0]["C"], z["it1_C_mean"]) <= 1e-8
    for k, g in enumerate(make()):
        body(g)
        S.same(read(g), got[k], exact=True)


def test_many_graphs_through_network_learn_cost_what_the_handle_costs():
    """64 graphs of the example's shape through Network.learn: per iteration within 3x of LDSBatch(64) driven directly with
    the same operations and a lower bound read back every iteration (the probe, profiles/example_probe.py, does 256)."""
    import time
    from pyvb_amd import nodes, synth
    from pyvb_amd.lds import LDSBatch
    from pyvb_amd.network import Network
    M, T, D, K = 64, 200, 2, 5
    probs = S.problems(T, D, K, M)
    graphs = S.build(nodes, probs)
    net = Network([n for g in graphs for n in S.all_nodes(g)])
    net.learn(2, tol=-np.inf, verbose=False)
    t0 = time.perf_counter()
    net.learn(20, tol=-np.inf, verbose=False)
    per_learn = (time.perf_counter() - t0) / 20
    Y = np.concatenate([p[0] for p in probs]); st0 = {k: np.concatenate([p[1][k] for p in probs]) for k in probs[0][1]}
    b = LDSBatch.from_problem(Y, st0, probs[0][2])

    def one():
        b.sweep("forward"); b.update_columns("A", 0, D); b.update_columns("C", 0, D); b.update_Q(); b.update_R()
        return b.elbo().sum()
    one(); one()
    t0 = time.perf_counter()
    for _ in range(20):
        llb = one()
    per_direct = (time.perf_counter() - t0) / 20
    print("Network.learn over %d graphs: %.3f ms per iteration; LDSBatch(%d) directly: %.3f ms" % (M, per_learn * 1e3, M, per_direct * 1e3))
    assert abs(net.llb - llb) <= 1e-10 * abs(llb)                  # 22 iterations of the same operations on the same data
    assert per_learn <= 3.0 * per_direct + 2e-4
