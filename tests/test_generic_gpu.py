"""Generic (node-by-node) path on the GPU: the tapes of pyvb_amd/generic.py through the HIP interpreter
(pyvb_amd/csrc/k_tape.hip, C ABI pyvb_graph_*), against the reference's fixtures and against the numpy restatement of
the interpreter (oracle/tape_ref.py).  Also the single messages / single lower-bound terms of the fused LDS and PCA plans,
which are served by a generic mirror of their state."""
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))
sys.path.insert(0, HERE)
import generic_scenarios as GS  # noqa: E402
from test_generic_cpu import check_scenario, _close  # noqa: E402


@pytest.mark.parametrize("name", sorted(GS.SCENARIOS))
def test_generic_scenarios_on_device(name, monkeypatch):
    from pyvb_amd import generic, _recognise
    if name not in GS.NEEDS_DEVICE:     # node by node even where a fused plan would take the graph (lds_missing_outputs)
        monkeypatch.setattr(_recognise, "bind", lambda node: generic.GenericPlan(node))
    named = check_scenario(name)
    plan = next(iter(named.values()))._plan
    if name in GS.NEEDS_DEVICE and not isinstance(plan, generic.GenericPlan):
        # left to the recogniser: the requests no fused kernel serves ran node by node, then the sweeps took the graph back
        assert isinstance(plan, _recognise.LDSPlan) and plan.resume_left < _recognise.LDSPlan.resume_left
    else:
        assert isinstance(plan, generic.GenericPlan) and isinstance(plan.ex, generic.DeviceExecutor)


def test_lds_with_missing_outputs_on_the_fused_plan():
    """The same graph and reference fixture as the generic scenario `lds_missing_outputs`, left to the recogniser: the
    fused LDS kernels take it (outputs with NaN are nodes of their own, k_missing.hip), per-node lower-bound terms
    come from the mirror."""
    from pyvb_amd._recognise import LDSPlan
    named = check_scenario("lds_missing_outputs")
    assert isinstance(named["X00"]._plan, LDSPlan) and named["X00"]._plan.free_ys == [2, 4]


def test_every_opcode_against_the_numpy_interpreter():
    """A tape that uses every opcode with odd shapes, run by the device and by oracle/tape_ref.py on the same arena."""
    from oracle import tape_ref as R
    from pyvb_amd import generic as G

    class P(object):        # the slice of GenericPlan that Tape needs
        temp_base, temp_high = 4096, 4096
        vals = []

        def const(self, v):
            self.vals.append(float(v))
            return G.Ref(len(self.vals) - 1, 1, 1)

        def ones(self, n):
            o = len(self.vals)
            self.vals.extend([1.0] * n)
            return G.Ref(o, n, 1)
    plan = P()
    rng = np.random.default_rng(0)
    base = 2048
    M = rng.standard_normal((7, 5)); S = rng.standard_normal((9, 9)); S = S @ S.T + 9 * np.eye(9)
    v = rng.random(9) + 0.5
    a = G.Ref(base, 7, 5); s = G.Ref(base + 64, 9, 9); vec = G.Ref(base + 256, 9, 1)
    idx_r = G.Ref(base + 300, 3, 1); idx_c = G.Ref(base + 310, 2, 1)
    t = G.Tape(plan)
    g1 = t.gemm(a, a, tb=True)                      # 7 x 7
    g2 = t.gemm(a, a, ta=True)                      # 5 x 5
    t.gemm(a, a, ta=True, dst=g2, acc=True, neg=True)
    inv, o2 = t.cholinv(s)
    tr = t.trace(inv)
    dg = t.diag_of(s)
    dm = t.diag_matrix(vec)
    ax = t.axpby(0.5, s, -2.0, dm)
    sc = t.scale(ax, tr, divide=True)
    un = [t.unary(vec, f) for f in (G.U_LOG, G.U_DIGAMMA, G.U_LGAMMA, G.U_RECIP, G.U_NEG, G.U_EXP)]
    ml = t.mul(vec, dg)
    tot = t.total(ml)
    ga = t.gather(s, idx_r, idx_c)
    tp = t.transpose(a)
    li = t.lin(1.5, [(2.0, tr), (-1.0, tot)])
    ey = t.eye(4); ze = t.zeros(3, 2); cp = t.copy(g1)
    outs = [g1, g2, inv, o2, tr, dg, dm, ax, sc, ml, tot, ga, tp, li, ey, ze, cp] + un
    size = plan.temp_high + 64
    arena = np.zeros(size)
    arena[:len(plan.vals)] = plan.vals
    arena[a.off:a.off + 35] = M.reshape(-1); arena[s.off:s.off + 81] = S.reshape(-1); arena[vec.off:vec.off + 9] = v
    arena[idx_r.off:idx_r.off + 3] = [8, 0, 4]; arena[idx_c.off:idx_c.off + 2] = [2, 7]
    ref = arena.copy()
    R.run(ref, t.array())
    ex = G.DeviceExecutor(size)
    ex.write(0, arena)
    ex.run(ex.tape(t.array()))
    got = ex.read(0, size)
    ex.close()
    for i, r in enumerate(outs):
        _close(got[r.off:r.off + r.size], ref[r.off:r.off + r.size], "output %d of the opcode tape" % i, 1e-11)
    _close(got[inv.off:inv.off + 81].reshape(9, 9), np.linalg.inv(S), "inverse", 1e-11)


def test_digamma_of_degenerate_arguments_terminates():
    """U_DIGAMMA walks x up to 10 by the recurrence: -inf, NaN or a huge negative argument (a degenerate qv in the Wishart
    bound, bad state) must end in NaN, not in a workgroup that never finishes; large negative non-integers go through the
    reflection formula and agree with scipy."""
    from scipy.special import digamma
    from pyvb_amd import generic as G
    x = np.array([-np.inf, -1e300, np.nan, np.inf, -1e16, -70.5, -1000.25, -63.5, 0.3, 12.0])
    ex = G.DeviceExecutor(64)
    ex.write(0, x)
    ex.run(ex.tape([[G.T_UNARY, 16, 0, 0, len(x), 1, 0, G.U_DIGAMMA]]))
    got = ex.read(16, len(x))
    ex.close()
    assert np.all(np.isnan(got[[0, 1, 2, 4]])) and got[3] == np.inf
    np.testing.assert_allclose(got[5:], digamma(x[5:]), rtol=1e-11)


def test_arena_ranges_do_not_wrap():
    """pyvb_graph_write / _read with an offset near SIZE_MAX: offset + n wraps around in size_t; the call must be refused."""
    import ctypes
    from pyvb_amd import _capi, generic
    ex = generic.DeviceExecutor(64)
    buf = np.zeros(4)
    huge = ctypes.c_size_t(-2).value                 # SIZE_MAX - 1
    for fn in (_capi.lib.pyvb_graph_write, _capi.lib.pyvb_graph_read):
        assert fn(ex._h if hasattr(ex, "_h") else ex.h, huge, _capi.dptr(buf), 4) == _capi.E_ARG
    ex.close()


def test_multiplication_expectations_against_the_reference(monkeypatch):
    """Every executable branch of Multiplication.pass_down_Ex / pass_down_ExxT (node.py:235-276: column x scalar, row vector,
    hstack, DiagonalGaussian) and Addition.pass_down_ExxT, as user-callable accessors, against values the reference's classes
    produced (tests/golden/expect_multiplication.npz) -- before any update and after one and two rounds."""
    from test_generic_cpu import check_expectations
    check_expectations()


def test_malformed_tapes_are_refused_and_bad_indices_skipped():
    """pyvb_graph_tape_create checks every extent a record touches against the arena; gather indices are data and are
    checked by the kernel (error reported at the next sync, nothing read or written outside)."""
    from pyvb_amd import _capi, generic
    ex = generic.DeviceExecutor(64)
    G = generic
    bad = [
        [G.T_GEMM, 0, 16, 32, 4, 4, 4, 0],          # b = 32..47 fine, a = 16..31 fine, dst fine -> but k * n beyond: see next
        [G.T_GEMM, 56, 0, 16, 4, 4, 4, 0],          # dst 56..71 leaves the arena of 64
        [G.T_COPY2D, 0, 8, 2, 3, 4, 4, 0],          # leading dimension of dst (2) smaller than n (4)
        [G.T_CHOLINV, 0, 16, 32, 4, 0, 40, 0],      # scratch 40 .. 40 + 2 * 16 leaves the arena
        [G.T_UNARY, 0, 8, 0, 2, 2, 0, 9],           # no such function
        [99, 0, 0, 0, 1, 1, 0, 0],                  # no such opcode
        [G.T_AXPBY, 0, 8, -1, 70, 1, 1, 0],         # 70 elements
    ]
    ex.tape([bad[0]])                                # the first one is well formed
    for rec in bad[1:]:
        with pytest.raises(_capi.PyvbHipError):
            ex.tape([rec])
    # gather with a row index that points outside: rows at 40, cols at 44
    ex.write(0, np.arange(16.0))
    ex.write(40, np.array([1.0, 1e6])); ex.write(44, np.array([0.0, 2.0]))
    ex.write(48, np.full(4, -7.0))
    t = ex.tape([[G.T_GATHER, 48, 0, 40, 2, 2, 4, 44]])
    ex.run(t)
    with pytest.raises(_capi.PyvbHipError):
        ex.sync()
    np.testing.assert_array_equal(ex.read(48, 4), [4.0, 6.0, -7.0, -7.0])    # the valid row gathered, the other left alone
    ex.close()


def test_not_positive_definite_raises_linalgerror():
    from pyvb_amd import nodes
    mu = nodes.Gaussian(2, np.zeros((2, 1)), -np.eye(2))
    y = nodes.Gaussian(2, mu, np.eye(2) * 0.5)
    y.observe(np.ones((2, 1)))
    mu.update()
    with pytest.raises(np.linalg.LinAlgError):
        mu.qmu


def test_single_messages_and_terms_of_the_fused_lds_plan(golden):
    """pass_up_m1_m2 and per-node log_lower_bound() on the nodes of a graph that runs through the fused LDS kernels: a
    generic mirror of the plan's state evaluates them (gaussian.py:136-151, :179-183; node.py:182-232;
    nodes_todo.py:43-62).  The per-node terms must add up to the class sums the reference's fixture holds."""
    from pyvb_amd import nodes
    sys.path.insert(0, os.path.join(HERE, "golden"))
    import make_golden as MG
    meta, Y, st0, pri, z = golden
    wishart = meta["noise"] == "wishart"
    g = MG.build_graph(nodes, Y[0], pri, {k: v for k, v in st0.items()})
    Xs, Ys, As, Cs, Q, R = g["Xs"], g["Ys"], g["As"], g["Cs"], g["Q"], g["R"]
    for t, y in enumerate(Ys):          # outputs with missing entries: the fixture's explicit initial posterior
        if not y.observed:
            y.qmu = st0["Yq"][0, t].reshape(-1, 1).copy()
            y.qcov = np.eye(meta["K"]) * st0["Yrowvar"][0, t]
    it = meta["iters"][0]
    for _ in range(it):
        [x.update() for x in Xs]
        Xs.reverse(); [x.update() for x in Xs]; Xs.reverse()
        [y.update() for y in Ys if not y.observed]
        [a.update() for a in As]; [c.update() for c in Cs]
        Q.update(); R.update()
    # the class sums of the fused kernels (k_elbo / k_elbo_dense) and, where the reference has a lower bound (not with a Wishart
    # parent: SURVEY.md Q8, parity unpinned there), the fixture's
    fused_parts = Xs[0]._plan.elbo_parts()
    ref = fused_parts if wishart else z["it%d_elbo_parts" % it]
    scale = np.abs(ref).sum()
    got = [sum(n.log_lower_bound() for n in grp) for grp in (Xs, Ys, As, Cs)] + [Q.log_lower_bound(), R.log_lower_bound()]
    for v, r, f in zip(got, ref, fused_parts):
        assert abs(v - r) <= 1e-8 * scale and abs(v - f) <= 1e-8 * scale, (got, ref, fused_parts)
    m1, m2 = Ys[1].pass_up_m1_m2(Ys[1].mean_parent)             # gaussian.py:179-183
    Rbar = R.pass_down_Ex()
    _close(m1, Rbar, "m1 of an observed output", 1e-9)
    _close(m2, Rbar @ Ys[1].qmu, "m2 of an observed output", 1e-9)
    T = len(Xs)
    if T > 2 and meta["D"] <= 16 and not wishart:               # the D^4 tensor below is what it says
        # Mult(., X_1) asked by X_1: the hstack branch with the D^4 tensor, node.py:213-227.  The fused plan answers from a
        # mirror of its state; the same graph forced node by node (its posteriors assigned from the fused run) must agree,
        # and so must the closed form <A^T L A> = Abar^T L Abar + diag_i tr(S_i L) that k_prep uses
        mults = [m for m in Xs[1].children]
        fused = [m.pass_up_m1_m2(Xs[1]) for m in mults]
        from pyvb_amd import generic, _recognise
        state = {"X": [x.qmu.copy() for x in Xs], "Xc": [x.qcov.copy() for x in Xs],
                 "A": [(a.qmu.copy(), a.qcov.copy()) for a in As], "C": [(c.qmu.copy(), c.qcov.copy()) for c in Cs],
                 "Q": (Q.qw.copy() if meta["noise"] == "wishart" else np.copy(Q.qb)), "R": (R.qw.copy() if meta["noise"] == "wishart" else np.copy(R.qb)),
                 "Y": [(y.qmu.copy(), y.qcov.copy()) for y in Ys]}
        import pytest as _pt
        mp = _pt.MonkeyPatch()
        try:
            mp.setattr(_recognise, "bind", lambda node: generic.GenericPlan(node))
            g2 = MG.build_graph(nodes, Y[0], pri, {k: v for k, v in st0.items()})
            for x, m, c in zip(g2["Xs"], state["X"], state["Xc"]):
                x.qmu, x.qcov = m, c
            for cols, key in ((g2["As"], "A"), (g2["Cs"], "C")):
                for n, (m, c) in zip(cols, state[key]):
                    n.qmu, n.qcov = m, c
            for y, (m, c) in zip(g2["Ys"], state["Y"]):
                if not y.observed:
                    y.qmu, y.qcov = m, c
            g2["Q"].qb, g2["R"].qb = state["Q"], state["R"]
            slow = [m.pass_up_m1_m2(g2["Xs"][1]) for m in g2["Xs"][1].children]
            assert isinstance(g2["Xs"][1]._plan, generic.GenericPlan)
        finally:
            mp.undo()
        for (f1, f2), (s1, s2) in zip(fused, slow):
            _close(f1, s1, "hstack-branch m1: fused mirror vs node by node", 1e-10)
            _close(f2, s2, "hstack-branch m2: fused mirror vs node by node", 1e-10)
        Qbar = Q.pass_down_Ex()
        Abar = np.hstack([a.qmu for a in As])
        closed = Abar.T @ Qbar @ Abar + np.diag([np.trace(a.qcov @ Qbar) for a in As])
        which = [i for i, m in enumerate(mults) if m.A is g["A"]][0]
        _close(fused[which][0], closed, "hstack-branch m1 vs Abar^T Q Abar + diag tr(S_i Q)", 1e-10)


def test_fused_plan_follows_a_late_observation(monkeypatch):
    """A node of a graph that already runs on the fused LDS kernels is observed afterwards (a known entry of A,
    examples/LDS_knowns_in_A.py:73-74, but after the first iterations): the plan is re-bound with its state carried over.
    Checked against the same script run node by node on the generic plan."""
    from pyvb_amd import nodes, synth, generic, _recognise
    import make_golden as MG
    T, D, K = 25, 3, 4
    Y, st0, pri = synth.make_problem(T, D, K, 1, 61)

    def script(force_generic):
        if force_generic:
            monkeypatch.setattr(_recognise, "bind", lambda node: generic.GenericPlan(node))
        else:
            monkeypatch.undo()
        g = MG.build_graph(nodes, Y[0], pri, st0)
        Xs, As, Cs, Q, R = g["Xs"], g["As"], g["Cs"], g["Q"], g["R"]

        def iteration():
            [x.update() for x in Xs]
            Xs.reverse(); [x.update() for x in Xs]; Xs.reverse()
            [a.update() for a in As]; [c.update() for c in Cs]
            Q.update(); R.update()
        iteration()
        first_plan = Xs[0]._plan
        As[1].observe(np.array([[0.5], [np.nan], [np.nan]]))
        iteration(); iteration()
        return (np.hstack([x.qmu for x in Xs]), np.hstack([a.qmu for a in As]), np.asarray(Q.qb), As[1].qcov, first_plan, Xs[0]._plan)

    ref = script(True)
    got = script(False)
    assert isinstance(got[4], _recognise.LDSPlan) and isinstance(got[5], _recognise.LDSPlan) and got[4] is not got[5]
    assert abs(got[1][0, 1] - 0.5) < 1e-15 and got[3][0, 0] == 0.0
    for a, b, what in zip(got[:4], ref[:4], ("X", "A", "Q.qb", "cov of the column")):
        _close(a, b, what, 1e-9)


def test_network_learn_in_crawl_order_on_the_lds_graph():
    """`Network([A]).fetch_network(); learn(3)` (network.py:40-96) on the LDS graph: the crawl order is not a sweep order,
    so after binding to the fused plan the graph moves to the generic one inside learn(); states, parameters and
    Network.llb against the reference's fixture (generic_lds_network_crawl.npz)."""
    from pyvb_amd import nodes
    from pyvb_amd.network import Network
    from pyvb_amd.generic import GenericPlan
    build, seed, _, _ = GS.SCENARIOS["lds_network_crawl"]
    order, named = build(nodes, np.random.default_rng(seed))
    z = dict(np.load(os.path.join(HERE, "golden", "generic_lds_network_crawl.npz"), allow_pickle=False))
    A = named["a00"].children[0]
    net = Network([A])
    net.fetch_network(verbose=False)
    net.learn(3, tol=-np.inf, verbose=False)
    assert isinstance(named["X00"]._plan, GenericPlan)
    for k, v in GS.snapshot(named).items():
        if k.endswith(".qcov") and np.abs(z["it3." + k]).max() == 0.0:
            continue
        _close(v, z["it3." + k], "after learn(3): " + k, 1e-9)
    ref = sum(float(z[k]) for k in z if k.startswith("it3.") and k.endswith(".llb"))
    assert abs(net.llb - ref) <= 1e-8 * abs(ref), (net.llb, ref)


def test_single_terms_of_the_fused_pca_plan():
    """Per-node log_lower_bound() and pass_up_m1_m2 on a graph that runs through the fused VB-PCA kernels (mirror of the
    plan's state on the generic plan): the per-node terms add up to the class sums of the reference's fixture."""
    from pyvb_amd import nodes
    import pyvb_amd
    import make_golden as MG
    from test_pca_oracle_golden import load_pca
    N, d, q, init, pri, z = load_pca(os.path.join(HERE, "golden", "pca_n60_d12_q3.npz"))

    class Mod(object):
        pass
    mod = Mod(); mod.nodes = nodes; mod.Network = pyvb_amd.Network
    g = MG.pca_build_graph(mod, init, pri)
    net = g["net"]
    net.learn(1, tol=-np.inf, verbose=False)
    from pyvb_amd._recognise import PCAPlan
    assert isinstance(g["W"]._plan, PCAPlan)
    ref = z["it1_elbo_parts"]
    scale = np.abs(ref).sum()
    got = [sum(n.log_lower_bound() for n in grp) for grp in (g["Ws"], g["Zs"], g["Xs"], [g["Mu"]], [g["Beta"]])]
    for v, r in zip(got, ref):
        assert abs(v - r) <= 1e-8 * scale, (got, ref)
    m1, m2 = g["Xs"][5].pass_up_m1_m2(g["Xs"][5].mean_parent)
    _close(m1, g["Beta"].pass_down_Ex(), "m1 of an output row", 1e-9)


@pytest.mark.parametrize("name", ["simple_PCA", "partial_observations", "lds_missing_outputs"] + ["random_%d" % s for s in GS.RANDOM_SEEDS[:12]])
def test_programs_of_many_node_tapes_on_device(name, monkeypatch):
    """Network.learn's tapes (all updates of an iteration; all lower-bound terms) issue nodes that touch disjoint posteriors side
    by side, one workgroup each (pyvb_graph_tape_set_program): same results as the one-node-at-a-time loop."""
    from pyvb_amd import generic, _recognise, nodes
    from test_generic_cpu import check_programs
    monkeypatch.setattr(_recognise, "bind", lambda node: generic.GenericPlan(node))
    plan = check_programs(name, lambda n: nodes._plan_of(n))
    assert isinstance(plan.ex, generic.DeviceExecutor)


def test_wishart_noise_with_known_entries_fused_against_node_by_node(monkeypatch):
    """A Wishart-noise LDS whose A and C have known entries (examples/LDS_knowns_in_A.py:73-74 with the Wishart lines of
    Linear_Dynamic_System.py:55-56): the recogniser now binds it to the fused kernels (round 2 refused the combination);
    the same script forced onto the node-by-node plan -- the reference's own schedule, gaussian.py:125-134 on dense
    covariances -- must agree in every posterior and in the bound of the columns."""
    from pyvb_amd import nodes, synth, generic, _recognise
    import make_golden as MG
    T, D, K = 20, 3, 4
    Y, st0, pri = synth.make_problem(T, D, K, 1, 83)
    pri["noise"] = "wishart"
    pri["Q_a0"], pri["Q_b0"] = np.float64(0.5 * D + 1.0), np.eye(D) * 0.05
    pri["R_a0"], pri["R_b0"] = np.float64(0.5 * K + 1.0), np.eye(K) * 0.05
    A_obs = np.full((D, D), np.nan); C_obs = np.full((K, D), np.nan)
    A_obs[0, 0] = 0.8; A_obs[2, 1] = -0.3
    C_obs[1, 2] = 2.0; C_obs[:, 0] = np.arange(K) - 1.0
    pri["A_obs"], pri["C_obs"] = A_obs, C_obs

    def script(force_generic):
        if force_generic:
            monkeypatch.setattr(_recognise, "bind", lambda node: generic.GenericPlan(node))
        else:
            monkeypatch.undo()
        g = MG.build_graph(nodes, Y[0], pri, st0)
        Xs, As, Cs, Q, R = g["Xs"], g["As"], g["Cs"], g["Q"], g["R"]
        rng = np.random.default_rng(1)          # positive definite starting qw (the constructor's is rank one)
        W = rng.standard_normal((D, D)); Q.qw = W @ W.T + D * np.eye(D)
        W = rng.standard_normal((K, K)); R.qw = W @ W.T + K * np.eye(K)
        for _ in range(2):
            [x.update() for x in Xs]
            Xs.reverse(); [x.update() for x in Xs]; Xs.reverse()
            [a.update() for a in As]; [c.update() for c in Cs]
            Q.update(); R.update()
        llb = [float(n.log_lower_bound()) for n in As + Cs]
        return (np.hstack([x.qmu for x in Xs]), np.hstack([a.qmu for a in As]), np.hstack([c.qmu for c in Cs]),
                np.stack([a.qcov for a in As]), np.stack([c.qcov for c in Cs]), np.asarray(Q.qw), np.asarray(R.qw),
                np.array(llb), Xs[0]._plan)

    ref = script(True)
    got = script(False)
    assert isinstance(got[8], _recognise.LDSPlan) and isinstance(ref[8], generic.GenericPlan)
    assert got[1][0, 0] == 0.8 and got[2][1, 2] == 2.0 and got[3][0][0, 0] == 0.0
    for a, b, what in zip(got[:8], ref[:8], ("X", "A", "C", "cov A", "cov C", "Q.qw", "R.qw", "column bounds")):
        _close(a, b, what, 1e-8)


def test_expectation_of_a_product_when_the_queue_hands_the_graph_over():
    """Found by profiles/fuzz_ops.py (round 3): an output is re-observed (the plan is bound anew), a lone update is queued on
    the fresh fused plan -- a request only the node-by-node plan serves -- and the next call is pass_down_Ex / pass_down_ExxT
    of a product.  Issuing the queue hands the graph over while the accessor is looking for its plan; it must follow, not
    read from the closed handle.  (Round 3 queued a lone column update there, which a freshly bound plan could not serve
    before its first sweep.  Since round 4 a plan bound from swept states starts from their three covariance classes
    (_recognise._state_classes) and serves it on the fused kernels: second half of the test; the hand-over is now
    provoked by a lone state update.)"""
    from pyvb_amd import nodes, synth, generic
    import make_golden as MG
    T, D, K = 12, 3, 4
    Y, st0, pri = synth.make_problem(T, D, K, 1, 19)

    def script(g, lone):
        Xs, As, Cs, Q, R, Ys = g["Xs"], g["As"], g["Cs"], g["Q"], g["R"], g["Ys"]
        [x.update() for x in Xs]; Xs.reverse(); [x.update() for x in Xs]; Xs.reverse()
        [a.update() for a in As]; [c.update() for c in Cs]; Q.update(); R.update()
        Ys[5].observe(Ys[5].qmu * 1.5)
        lone(g).update()
        return Ys[2].mean_parent.pass_down_Ex(), Ys[2].mean_parent.pass_down_ExxT(), Xs[0]._plan

    from pyvb_amd import _recognise
    import pytest as _pt
    for lone, kind in ((lambda g: g["Xs"][3], generic.GenericPlan), (lambda g: g["As"][2], _recognise.LDSPlan)):
        ex, exxt, plan = script(MG.build_graph(nodes, Y[0], pri, st0), lone)
        assert isinstance(plan, kind)
        mp = _pt.MonkeyPatch()
        try:
            mp.setattr(_recognise, "bind", lambda node: generic.GenericPlan(node))
            ex2, exxt2, _ = script(MG.build_graph(nodes, Y[0], pri, st0), lone)
        finally:
            mp.undo()
        _close(ex, ex2, "pass_down_Ex after the hand-over", 1e-10)
        _close(exxt, exxt2, "pass_down_ExxT after the hand-over", 1e-10)


def _plan_stub():
    from pyvb_amd import generic as G

    class P(object):
        temp_base, temp_high = 4096, 4096
        vals = []

        def const(self, v):
            self.vals.append(float(v))
            return G.Ref(len(self.vals) - 1, 1, 1)

        def ones(self, n):
            o = len(self.vals)
            self.vals.extend([1.0] * n)
            return G.Ref(o, n, 1)
    return P()


def _run_both(t, plan, fill, size):
    """The tape on the device and in the numpy interpreter, same arena."""
    from oracle import tape_ref as R
    from pyvb_amd import generic as G
    arena = np.zeros(size)
    arena[:len(plan.vals)] = plan.vals
    for off, arr in fill:
        arena[off:off + arr.size] = arr.reshape(-1)
    ref = arena.copy()
    R.run(ref, t.array())
    ex = G.DeviceExecutor(size)
    ex.write(0, arena)
    ex.run(ex.tape(t.array()))
    got = ex.read(0, size)
    ex.close()
    return got, ref


def test_long_tapes_are_staged_in_chunks_and_windows_that_do_not_fit_fall_back():
    """The interpreter keeps a block's working set in LDS when it fits (k_tape.hip): (a) a tape of 1700 records -- more than
    the 512 staged at a time -- on a small working set, (b) a tape whose working set (three 80 x 80 matrices and their
    products) exceeds the window and stays on global memory, (c) a tape with a gather record (addresses that are data: never
    cached).  All three against the numpy interpreter on the same arena."""
    from pyvb_amd import generic as G
    rng = np.random.default_rng(4)
    # (a)
    plan = _plan_stub()
    a = G.Ref(2048, 6, 6); b = G.Ref(2100, 6, 6)
    A = rng.standard_normal((6, 6)) * 0.3; B = rng.standard_normal((6, 6)) * 0.3
    t = G.Tape(plan)
    acc = t.copy(a)
    for k in range(560):                    # three records per turn
        g = t.gemm(acc, b)
        t.axpby(0.5, g, 0.5, a, dst=acc)
        t.mul(acc, acc) if k % 7 == 0 else t.unary(acc, G.U_NEG)
    assert len(t.ops) > 3 * 512
    got, ref = _run_both(t, plan, [(a.off, A), (b.off, B)], plan.temp_high + 64)
    _close(got[acc.off:acc.off + 36], ref[acc.off:acc.off + 36], "result of the long tape", 1e-11)
    # (b)
    plan = _plan_stub()
    m = 80
    x, y, z = G.Ref(8192, m, m), G.Ref(8192 + m * m, m, m), G.Ref(8192 + 2 * m * m, m, m)
    plan.temp_base = plan.temp_high = 8192 + 3 * m * m
    t = G.Tape(plan)
    p1 = t.gemm(x, y); p2 = t.gemm(p1, z, tb=True); p3 = t.add(p2, t.transpose(p1))
    tr = t.trace(p3)
    mats = [rng.standard_normal((m, m)) / m for _ in range(3)]
    got, ref = _run_both(t, plan, [(x.off, mats[0]), (y.off, mats[1]), (z.off, mats[2])], plan.temp_high + 64)
    assert (plan.temp_high - 8192) > 12288      # larger than the LDS window
    _close(got[p3.off:p3.off + m * m], ref[p3.off:p3.off + m * m], "products beyond the window", 1e-11)
    _close(got[tr.off:tr.off + 1], ref[tr.off:tr.off + 1], "their trace", 1e-11)
    _close(ref[p3.off:p3.off + m * m].reshape(m, m), (mats[0] @ mats[1]) @ mats[2].T + (mats[0] @ mats[1]).T, "against numpy", 1e-11)
    # (c)
    plan = _plan_stub()
    s = G.Ref(2048, 5, 5); rows = G.Ref(2100, 2, 1); cols = G.Ref(2110, 3, 1)
    t = G.Tape(plan)
    sq = t.gemm(s, s, tb=True)
    ga = t.gather(sq, rows, cols)
    out = t.scale(ga, 2.0)
    t.axpby(1.0, out, 1.0, out, dst=out)
    S = rng.standard_normal((5, 5))
    got, ref = _run_both(t, plan, [(s.off, S), (rows.off, np.array([4.0, 1.0])), (cols.off, np.array([0.0, 2.0, 3.0]))], plan.temp_high + 64)
    _close(got[out.off:out.off + 6], ref[out.off:out.off + 6], "gather inside a tape", 1e-12)
    _close(ref[out.off:out.off + 6].reshape(2, 3), 4.0 * (S @ S.T)[[4, 1]][:, [0, 2, 3]], "against numpy", 1e-12)


def test_records_scheduled_into_bundles_compute_what_the_tape_order_computes():
    """k_tape.hip schedules the records of a window whose operands are node-sized into bundles of mutually independent records,
    a wavefront each (tape_bundle): any order that respects the read / write dependencies must give the tape's result.  Random
    tapes of small records on a crowded arena -- plenty of RAW, WAR and WAW hazards, accumulating records, reductions, a
    Cholesky inverse now and then, a few gathers that interrupt the windows -- against the numpy interpreter, which runs the
    records in tape order.  The last case is a chain whose working set is larger than one window: it is cut into several."""
    from pyvb_amd import generic as G
    from oracle import tape_ref as R
    rng = np.random.default_rng(11)
    for case in range(12):
        size = 4096 if case < 11 else 40000
        nslot = 24 if case < 11 else 4000
        base = 64
        shapes = [(2, 2), (3, 3), (2, 1), (3, 1), (1, 1), (4, 4)]
        slots = []                              # (offset, m, n): matrices laid out back to back, so neighbours never overlap
        off = base
        for k in range(nslot):
            m, n = shapes[int(rng.integers(len(shapes)))]
            slots.append((off, m, n)); off += m * n
        assert off + 64 < size
        by_shape = {}
        for sl in slots:
            by_shape.setdefault((sl[1], sl[2]), []).append(sl)
        scal = [o for o, m, n in slots if (m, n) == (1, 1)] or [base]
        ops = []
        nrec = 700 if case < 11 else 6000
        idx_rows, idx_cols = size - 40, size - 30
        for r in range(nrec):
            kind = int(rng.integers(9))
            (m, n) = shapes[int(rng.integers(len(shapes)))]
            cand = by_shape.get((m, n), [])
            if len(cand) < 3:
                continue
            pick = lambda: cand[int(rng.integers(len(cand)))][0] if case < 11 else cand[min(len(cand) - 1, (r * len(cand)) // nrec + int(rng.integers(3)))][0]
            d_, a_, b_ = pick(), pick(), pick()
            if kind == 0 and d_ != a_: ops.append([G.T_COPY2D, d_, a_, n, m, n, n, 0])
            elif kind == 1 and len({d_, a_, b_}) == 3: ops.append([G.T_AXPBY, d_, a_, b_, m, n, int(rng.choice(scal)), int(rng.choice(scal))])
            elif kind == 2 and m == n and len({d_, a_, b_}) == 3: ops.append([G.T_GEMM, d_, a_, b_, m, n, m, int(rng.integers(8)) & 7])
            elif kind == 3 and len({d_, a_, b_}) == 3: ops.append([G.T_MUL, d_, a_, b_, m, n, 0, 0])
            elif kind == 4 and d_ != a_: ops.append([G.T_UNARY, d_, a_, 0, m, n, 0, 4])
            elif kind == 5 and m == n: ops.append([G.T_TRACE, int(rng.choice(scal)), a_, 0, m, m, 0, int(rng.integers(2)) * 4])
            elif kind == 6 and len({a_, b_}) == 2: ops.append([G.T_DOT, int(rng.choice(scal)), a_, b_, m, n, 0, 0])
            elif kind == 7 and d_ != a_: ops.append([G.T_FILL, d_, 0, n, m, n, 0, int(rng.integers(2))])
            elif kind == 8 and case % 3 == 2 and r % 50 == 49 and m == n and m >= 2:
                ops.append([G.T_GATHER, d_, a_, idx_rows, m, n, n, idx_cols])
        ops = np.asarray(ops, dtype=np.int32)
        arena = np.zeros(size)
        arena[base:off] = rng.uniform(-0.9, 0.9, off - base)        # |values| < 1: products and sums of a few hundred records stay finite
        arena[idx_rows:idx_rows + 4] = [1, 0, 2, 1]; arena[idx_cols:idx_cols + 4] = [0, 1, 1, 0]
        # keep magnitudes in check: the tape is random, a chain of products may blow up -- rescale through a dry run
        ref = arena.copy()
        R.run(ref, ops)
        if not np.all(np.isfinite(ref)) or np.abs(ref).max() > 1e100:
            continue
        ex = G.DeviceExecutor(size)
        ex.write(0, arena)
        ex.run(ex.tape(ops))
        got = ex.read(0, size)
        ex.close()
        scale = max(1.0, np.abs(ref).max())
        assert np.abs(got - ref).max() <= 1e-11 * scale, "case %d: %g" % (case, np.abs(got - ref).max() / scale)


def test_cached_tapes_survive_a_growing_arena(monkeypatch):
    """The node-by-node plan uploads a node's tape once and keeps its id.  When a later request needs a larger arena the executor is
    replaced and every tape is uploaded again: a request for a tape cached on the OLD executor must not run its stale id (found by
    profiles/fuzz_ops_pca.py, seed 12: "no such tape" -- the id was read before the executor was replaced)."""
    from pyvb_amd import generic, _recognise, nodes
    monkeypatch.setattr(_recognise, "bind", lambda node: generic.GenericPlan(node))
    rng = np.random.default_rng(5)
    np.random.seed(3)                           # the constructors draw their initial posteriors from numpy's global generator
    mu = nodes.Gaussian(3, np.zeros((3, 1)), np.eye(3) * 1e-2)
    prec = nodes.DiagonalGamma(3, np.ones(3), np.ones(3))
    xs = [nodes.Gaussian(3, mu, prec) for _ in range(6)]
    for x in xs:
        x.observe(rng.standard_normal((3, 1)))
    mu.update(); first = np.array(mu.qmu)
    plan = nodes._plan_of(mu)
    old = plan.ex
    plan.temp_high = old.size + 50000           # what a request with large temporaries does to the plan
    prec.update(); _ = np.array(prec.qb)        # (requests are queued: the read issues them) runs on a new, larger executor
    assert plan.ex is not old
    mu.update()                                 # its tape was cached on the old executor
    again = np.array(mu.qmu)
    assert np.all(np.isfinite(again)) and again.shape == first.shape
    # the same schedule without the growth
    np.random.seed(3)
    mu2 = nodes.Gaussian(3, np.zeros((3, 1)), np.eye(3) * 1e-2)
    prec2 = nodes.DiagonalGamma(3, np.ones(3), np.ones(3))
    xs2 = [nodes.Gaussian(3, mu2, prec2) for _ in range(6)]
    for x, x0 in zip(xs2, xs):
        x.observe(np.array(x0.qmu))
    mu2.update(); _ = np.array(mu2.qmu); prec2.update(); _ = np.array(prec2.qb); mu2.update()
    assert np.allclose(again, np.array(mu2.qmu), rtol=1e-13, atol=0)


def test_launches_of_many_short_blocks_take_narrow_workgroups(monkeypatch):
    """A queued run whose launches have 512 blocks and more is interpreted by workgroups of four wavefronts with bundles of four
    records (k_tape_cached<4>), the others by eight (k_tape.hip): the loop over 600 latent nodes and their 600 outputs issued at
    once must leave what the same updates issued one node at a time leave (each of those is a one-block tape of its own)."""
    from pyvb_amd import generic, _recognise, nodes
    monkeypatch.setattr(_recognise, "bind", lambda node: generic.GenericPlan(node))
    N, d, q = 600, 6, 2

    def build():
        np.random.seed(11)
        rng = np.random.default_rng(3)
        X = rng.standard_normal((N, q)) @ rng.standard_normal((q, d)) + 0.1 * rng.standard_normal((N, d))
        X[rng.random((N, d)) < 0.1] = np.nan
        Ws = [nodes.Gaussian(d, np.zeros((d, 1)), np.eye(d) * 1e-3) for _ in range(q)]
        W = nodes.hstack(Ws)
        Mu = nodes.Gaussian(d, np.zeros((d, 1)), np.eye(d) * 1e-3)
        Beta = nodes.Gamma(d, 1e-3, 1e-3)
        Zs = [nodes.Gaussian(q, np.zeros((q, 1)), np.eye(q)) for _ in range(N)]
        Xs = [nodes.Gaussian(d, W * z + Mu, Beta) for z in Zs]
        for x, row in zip(Xs, X):
            x.observe(row.reshape(d, 1))
        return Ws, Mu, Beta, Zs, Xs

    Ws, Mu, Beta, Zs, Xs = build()
    for _ in range(2):                          # queued: the Z and X loops are launches of 600 blocks
        [w.update() for w in Ws]; [z.update() for z in Zs]; [x.update() for x in Xs]; Mu.update(); Beta.update()
        _ = np.array(Mu.qmu)
    plan = nodes._plan_of(Mu)
    progs = [p for p in plan._programs.values() if p is not None]
    assert progs and max(int(l[1]) for p in progs for l in p[1]) >= 512
    Ws2, Mu2, Beta2, Zs2, Xs2 = build()
    for _ in range(2):                          # one node at a time: a read after every update
        for n in Ws2 + Zs2 + Xs2 + [Mu2, Beta2]:
            n.update()
            _ = np.array(n.qb if isinstance(n, nodes.Gamma) else n.qmu)
    for a, b in zip(Ws + Zs + Xs + [Mu], Ws2 + Zs2 + Xs2 + [Mu2]):
        assert np.allclose(np.array(a.qmu), np.array(b.qmu), rtol=1e-11, atol=1e-13)
    assert np.allclose(np.array(Beta.qb), np.array(Beta2.qb), rtol=1e-11)
