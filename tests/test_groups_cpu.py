"""Host logic, no GPU: M LDS graphs of one structure built through the node API share ONE device handle, a replicate
each (pyvb_amd/_recognise.py: LDSGroup), their queued update() requests run in lock step, graphs whose requests part ways
leave with their state, and Network.learn replays per handle from its second iteration on (pyvb_amd/network.py:
_Schedule).  The handle is tests/oracle_batch.py's stand-in (the oracle behind LDSBatch's interface), the node-by-node
plan runs on the numpy interpreter; tests/test_groups_gpu.py repeats the scenarios on the HIP library."""
import os

import numpy as np
import pytest

import group_scenarios as S
from conftest import load_golden, GOLDEN_DIR


@pytest.fixture
def host_only(monkeypatch):
    from oracle.tape_ref import NumpyExecutor
    from oracle_batch import OracleBatch
    from pyvb_amd import generic, lds, _recognise
    OracleBatch.instances = []
    monkeypatch.setattr(lds, "LDSBatch", OracleBatch)
    monkeypatch.setattr(generic, "DeviceExecutor", NumpyExecutor)
    _recognise._pool.clear()
    return OracleBatch


def _fixture_problems():
    out = []
    for name in ("example_d2k5_t200", "example_b_d2k5_t200"):
        meta, Y, st0, pri, z = load_golden(os.path.join(GOLDEN_DIR, "lds_%s.npz" % name))
        out.append((Y, st0, pri, z))
    return out


def test_eight_graphs_share_one_handle_and_match_the_reference(host_only):
    from pyvb_amd import nodes
    fx = _fixture_problems()
    pri = fx[0][2]
    probs = [(Y, st0, pri) for Y, st0, _, _ in fx] + S.problems(200, 2, 5, 6, pri=pri)
    graphs = S.build(nodes, probs)
    for it in range(2):
        for g in graphs:
            S.loop_body(g)
    assert host_only.instances == []                    # nothing has been asked for yet: no handle
    snaps = [S.snapshot(g) for g in graphs]
    assert len(host_only.instances) == 1 and host_only.instances[0].N == 8
    grp = graphs[0]["Xs"][0]._plan.group
    assert [m is g["Xs"][0]._plan for m, g in zip(grp.members, graphs)] == [True] * 8
    assert host_only.instances[0].log == ["forward", "backward", ("A", 0, 2), ("C", 0, 2), "Q", "R"] * 2
    for k in (0, 1):                                    # the two graphs whose data the reference has run
        z = fx[k][3]
        assert S.rel(snaps[k]["X"], z["it2_X"]) <= 1e-8 and S.rel(snaps[k]["A"], z["it2_A_mean"]) <= 1e-8
        assert S.rel(snaps[k]["C"], z["it2_C_mean"]) <= 1e-8 and S.rel(snaps[k]["Qb"], z["it2_Q_b"]) <= 1e-8
        assert S.rel(snaps[k]["S1"], z["it2_Sigma"][1]) <= 1e-8 and S.rel(snaps[k]["Rb"], z["it2_R_b"]) <= 1e-8
        parts = graphs[k]["Xs"][0]._plan.elbo_parts()
        assert abs(parts.sum() - z["it2_elbo_parts"].sum()) <= 1e-8 * abs(z["it2_elbo_parts"].sum())
    # the same eight, each bound on its own (read before the next graph gets its first request)
    host_only.instances = []
    twins = S.build(nodes, probs)
    for k, g in enumerate(twins):
        for it in range(2):
            S.loop_body(g)
        S.same(S.snapshot(g), snaps[k], exact=False)
    assert len(host_only.instances) == 8 and all(b.N == 1 for b in host_only.instances)


def test_a_request_the_fused_kernels_do_not_serve_moves_one_graph_only(host_only):
    from pyvb_amd import nodes, generic
    from pyvb_amd._recognise import LDSPlan
    probs = S.problems(12, 3, 4, 8)
    graphs, twins = S.build(nodes, probs), S.build(nodes, probs)

    def script(gs, together):
        for g in gs:
            S.loop_body(g)
            together or S.snapshot(g)
        gs[3]["Xs"][5].update()                         # a lone X_t.update(): node by node
        got = gs[3]["Xs"][5].qmu
        for g in gs:
            S.loop_body(g)
            together or S.snapshot(g)
        return got

    a = script(graphs, True)
    snaps = [S.snapshot(g) for g in graphs]
    handle = host_only.instances[0]
    assert [b.N for b in host_only.instances] == [8, 1] and not handle.closed
    grp = graphs[0]["Xs"][0]._plan.group
    assert len(grp.live()) == 7 and grp.members[3] is None
    # graph 3 went to the node-by-node plan, and the loop's sweeps brought it back to the fused kernels on a handle of its own
    assert all(isinstance(g["Xs"][0]._plan, LDSPlan) for g in graphs)
    assert graphs[3]["Xs"][0]._plan.group is not grp and graphs[3]["Xs"][0]._plan.resume_left == LDSPlan.resume_left - 1
    host_only.instances = []
    b = script(twins, False)
    assert S.rel(a, b) <= 1e-11
    for k in range(8):
        S.same(S.snapshot(twins[k]), snaps[k], exact=False, tol=1e-10)


def test_network_learn_over_graphs_of_one_structure(host_only, monkeypatch):
    from pyvb_amd import nodes, _recognise
    from pyvb_amd.network import Network
    probs = S.problems(30, 3, 4, 6)
    graphs = S.build(nodes, probs)
    net = Network([n for g in graphs for n in S.all_nodes(g)])
    replayed, enqueued = [], []
    run_script, enqueue = _recognise.LDSGroup.run_script, _recognise.LDSPlan.enqueue
    monkeypatch.setattr(_recognise.LDSGroup, "run_script", lambda self, sc: replayed.append(list(sc)) or run_script(self, sc))
    monkeypatch.setattr(_recognise.LDSPlan, "enqueue", lambda self, n: enqueued.append(1) or enqueue(self, n))
    net.learn(4, tol=-np.inf, verbose=False)
    # the node list is walked once (30 + 3 + 3 + 2 update() calls per graph); iterations 2..4 replay what it spelt, per handle
    assert len(enqueued) == 6 * 38 and replayed == [[("F",), ("A", 0, 3), ("C", 0, 3), ("Q",), ("R",)]] * 3
    monkeypatch.undo()
    monkeypatch.setattr(__import__("pyvb_amd.lds").lds, "LDSBatch", host_only)
    monkeypatch.setattr(__import__("pyvb_amd.generic").generic, "DeviceExecutor", __import__("oracle.tape_ref").tape_ref.NumpyExecutor)
    assert len(host_only.instances) == 1 and host_only.instances[0].N == 6
    per_iter = ["forward", ("A", 0, 3), ("C", 0, 3), "Q", "R", "elbo"]     # list order: every node once (network.py:46-48)
    assert host_only.instances[0].log == per_iter * 4
    total = 0.0
    for k, (Y, st0, pri) in enumerate(probs):           # every graph alone
        g = S.build(nodes, [probs[k]])[0]
        one = Network(S.all_nodes(g))
        one.learn(4, tol=-np.inf, verbose=False)
        total += one.llb
        S.same(S.snapshot(g), S.snapshot(graphs[k]), exact=False)
    assert abs(net.llb - total) <= 1e-11 * abs(total)
    # a second call over TWO of the six: the four that are not asked keep the handle, the two leave together
    old = graphs[0]["Xs"][0]._plan.group
    before = len(host_only.instances)
    sub = Network([n for g in graphs[:2] for n in S.all_nodes(g)])
    sub.learn(2, tol=-np.inf, verbose=False)
    new = graphs[0]["Xs"][0]._plan.group
    assert new is not old and new is graphs[1]["Xs"][0]._plan.group and len(new.members) == 2
    assert len(old.live()) == 4 and graphs[2]["Xs"][0]._plan.group is old
    assert len(host_only.instances) == before + 1
    for k in (0, 1):
        g = S.build(nodes, [probs[k]])[0]
        one = Network(S.all_nodes(g))
        one.learn(4, tol=-np.inf, verbose=False)
        one.learn(2, tol=-np.inf, verbose=False)
        S.same(S.snapshot(g), S.snapshot(graphs[k]), exact=False)
    # a graph that leaves takes the covariances of its states along: its parameters can update right away on the new handle
    # (pyvb_lds_set_posterior_classes), no sweep first and no node-by-node plan
    from pyvb_amd._recognise import LDSPlan
    [a.update() for a in graphs[5]["As"]]
    got = np.hstack([a.qmu for a in graphs[5]["As"]])
    plan = graphs[5]["Xs"][0]._plan
    assert isinstance(plan, LDSPlan) and plan.group is not old and len(old.live()) == 3 and plan.group.batch.log == [("A", 0, 3)]
    g = S.build(nodes, [probs[5]])[0]
    Network(S.all_nodes(g)).learn(4, tol=-np.inf, verbose=False)
    [a.update() for a in g["As"]]
    assert S.rel(got, np.hstack([a.qmu for a in g["As"]])) <= 1e-11


def test_different_priors_or_shapes_do_not_share_a_handle(host_only):
    from pyvb_amd import nodes, synth
    probs = S.problems(10, 2, 3, 2)
    Y, st0, pri = synth.make_problem(10, 2, 3, 1, 77)
    pri = dict(pri, A_prior_prec=np.full((2, 2), 1e-2))
    probs.append((Y, st0, pri))
    probs += S.problems(11, 2, 3, 1, seed=50)
    graphs = S.build(nodes, probs)
    for g in graphs:
        S.loop_body(g)
    [S.snapshot(g) for g in graphs]
    assert sorted(b.N for b in host_only.instances) == [1, 1, 2]
    assert graphs[0]["Xs"][0]._plan.group is graphs[1]["Xs"][0]._plan.group


def test_assignments_reach_the_right_replicate(host_only):
    from pyvb_amd import nodes
    probs = S.problems(9, 2, 3, 3)
    graphs, twins = S.build(nodes, probs), S.build(nodes, probs)
    for gs, together in ((graphs, True), (twins, False)):
        for g in gs:
            S.loop_body(g)
            together or S.snapshot(g)
        gs[1]["As"][0].qmu = np.array([[0.25], [-0.5]])         # on the handle: patched in place
        gs[1]["Xs"][4].qmu = np.array([[1.0], [2.0]])
        gs[2]["Q"].qb = np.array([0.3, 0.4])
        for g in gs:
            S.loop_body(g)
            together or S.snapshot(g)
    assert len(host_only.instances) == 4 and host_only.instances[0].N == 3
    for g, t in zip(graphs, twins):
        S.same(S.snapshot(g), S.snapshot(t), exact=False)
    # before anything has run the nodes' own attributes are the state: an assignment needs no handle
    more = S.build(nodes, probs[:1])[0]
    [x.update() for x in more["Xs"]]
    n = len(host_only.instances)
    more["As"][1].qmu = np.array([[2.0], [3.0]])                # after the queued sweep
    assert len(host_only.instances) == n + 1
    assert np.array_equal(more["As"][1].qmu, np.array([[2.0], [3.0]]))


def test_random_interleavings_over_graphs_that_share_a_handle(host_only, capsys):
    """profiles/fuzz_groups.py on the host-only stand-ins: random operations to one, some or all of M graphs of one structure,
    every read against a twin on the node-by-node plan.  (The GPU box runs the same script on the HIP library with all noise
    kinds: profiles/r04/fuzz_groups.txt.)"""
    import importlib.util, sys
    prof = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles")
    sys.path.insert(0, prof)
    try:
        spec = importlib.util.spec_from_file_location("fuzz_groups", os.path.join(prof, "fuzz_groups.py"))
        F = importlib.util.module_from_spec(spec); spec.loader.exec_module(F)
        worst, shared, left = F.main(cases=12, seed=5, noises=("gamma", "diagonal_gamma"), noise_p=(0.5, 0.5), allow_missing=False)
    finally:
        sys.path.remove(prof)
    assert worst < 1e-8 and shared > 20 and left > 0
