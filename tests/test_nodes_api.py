"""CPU-only tests of the pyvb-compatible front end: constructors, error behaviour, the network
crawl order (against the order recorded from the reference) and the LDS recogniser.  No kernel runs."""
import importlib.util
import os

import numpy as np
import pytest

from pyvb_amd import nodes, synth, _recognise
from pyvb_amd.network import Network

HERE = os.path.dirname(os.path.abspath(__file__))


def _golden_module():
    spec = importlib.util.spec_from_file_location("make_golden", os.path.join(HERE, "golden", "make_golden.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def _graph(T=6, D=3, K=4, kind="diagonal_gamma", seed=3):
    Y, st0, pri = synth.make_problem(T, D, K, 1, seed)
    pri["noise"] = kind
    if kind == "gamma":
        for k in ("Q_a0", "Q_b0", "R_a0", "R_b0"):
            pri[k] = np.float64(1e-3)
    return _golden_module().build_graph(nodes, Y[0], pri, st0), Y, st0, pri


def test_constructor_checks_match_the_reference():
    with pytest.raises(AssertionError):                      # gaussian.py:46
        nodes.Gaussian(3, np.zeros((2, 1)), np.eye(3))
    with pytest.raises(AssertionError):                      # gaussian.py:55
        nodes.Gaussian(3, np.zeros((3, 1)), np.eye(2))
    g = nodes.Gaussian(2, np.zeros((2, 1)), np.eye(2))
    class Fake(object):
        def __init__(self, shape):
            self.shape = shape
    with pytest.raises(nodes.ConjugacyError):                # gaussian.py:52: not a Gaussian-family mean parent
        nodes.Gaussian(2, Fake((2, 1)), np.eye(2))
    with pytest.raises(nodes.ConjugacyError):                # gaussian.py:61: not a Gamma-family precision parent
        nodes.Gaussian(2, np.zeros((2, 1)), Fake((2, 2)))
    assert issubclass(nodes.ConjugacyError, ValueError)
    with pytest.raises(AssertionError):                      # node.py:166-167
        nodes.Multiplication(nodes.Constant(np.eye(3)), g)
    with pytest.raises(AssertionError):
        nodes.Addition(g, nodes.Constant(np.zeros((3, 1))))
    with pytest.raises(AssertionError):                      # nodes_todo.py:180
        nodes.DiagonalGamma(3, np.ones(3), np.ones(3)).addChild(g)
    with pytest.raises(AssertionError):
        g.observe(np.zeros((3, 1)))


def test_operator_overloads_and_array_wrapping():
    g = nodes.Gaussian(2, np.zeros((2, 1)), np.eye(2))
    m = np.ones((3, 2)) * g                                  # ndarray * node -> Multiplication with a Constant (Q4 fixed)
    assert isinstance(m, nodes.Multiplication) and isinstance(m.A, nodes.Constant) and m.shape == (3, 1)
    assert m in g.children and m in m.A.children
    s = g + np.ones((2, 1))
    assert isinstance(s, nodes.Addition) and isinstance(s.B, nodes.Constant)
    c = nodes.Constant(np.diag([2.0, 3.0]))
    assert np.isclose(c.pass_down_lndet(), np.log(6.0))


def test_noise_node_bookkeeping():
    Q = nodes.DiagonalGamma(2, np.ones(2) * 1e-3, np.ones(2) * 1e-3)
    G = nodes.Gamma(3, 1e-3, 1e-3)
    for _ in range(4):
        nodes.Gaussian(2, np.zeros((2, 1)), Q)
        nodes.Gaussian(3, np.zeros((3, 1)), G)
    assert np.allclose(Q.qa, 1e-3 + 0.5 * 4)                 # nodes_todo.py:183-186
    assert np.isclose(G.qa, 1e-3 + 0.5 * 3 * 4)              # nodes_todo.py:125-128
    assert Q.pass_down_Ex().shape == (2, 2) and G.pass_down_Ex().shape == (3, 3)


def test_observe_variants():
    g = nodes.Gaussian(3, np.zeros((3, 1)), np.eye(3))
    g.observe(np.full((3, 1), np.nan))                       # gaussian.py:90-91: nothing observed
    assert not g.observed and not g.partially_observed
    g.observe(np.array([[1.0], [np.nan], [2.0]]))            # :92-96
    assert g.partially_observed and list(g.obs_index) == [0, 2] and list(g.missing_index) == [1]
    h = nodes.Gaussian(2, np.zeros((2, 1)), np.eye(2))
    v = np.array([[1.0], [2.0]])
    h.observe(v)                                             # :97-100
    assert h.observed and np.array_equal(h.qmu, v) and not h.qcov.any()
    h.update()                                               # observed nodes return immediately (:109-110)


def test_fetch_network_order_matches_reference():
    z = np.load(os.path.join(HERE, "golden", "crawl_lds_t4.npz"), allow_pickle=False)
    got = _golden_module().crawl_labels(__import__("pyvb_amd"), 4, 2, 3)
    assert got == [str(s) for s in z["order"]]


def test_recogniser_extracts_the_lds(capsys):
    g, Y, st0, pri = _graph()
    d = _recognise.describe(g["Q"])                          # any node of the graph will do
    assert [x is y for x, y in zip(d["Xs"], g["Xs"])] == [True] * 6
    assert [x is y for x, y in zip(d["Ys"], g["Ys"])] == [True] * 6
    assert d["A"] is g["A"] and d["C"] is g["C"] and d["Q"] is g["Q"] and d["R"] is g["R"]
    for k in ("x0_mean", "x0_prec", "A_prior_mean", "A_prior_prec", "C_prior_mean", "C_prior_prec", "Q_a0", "Q_b0", "R_a0", "R_b0"):
        assert np.array_equal(np.asarray(d["pri"][k]), np.asarray(pri[k])), k
    net = Network([g["C"]])
    net.fetch_network(verbose=False)
    net.find_iterable()
    assert len(net.iterable_nodes) == 2 * 6 + 2 * 3 + 2


def test_graphs_without_a_fused_plan_go_node_by_node():
    """Graphs other than the LDS and the VB-PCA graph bind to the generic node-by-node plan (pyvb_amd/generic.py); the
    fused recognisers say why they decline.  (Binding needs no GPU: the arena is created at the first launch.)"""
    from pyvb_amd.generic import GenericPlan
    # simple mean inference (src/tests.py:9-19): a valid pyvb graph, but not the LDS path
    mu = nodes.Gaussian(1, np.zeros((1, 1)), np.eye(1) * 1e-3)
    prec = nodes.Gamma(1, 1e-3, 1e-3)
    xs = [nodes.Gaussian(1, mu, prec) for _ in range(5)]
    [x.observe(np.random.randn(1, 1)) for x in xs]
    with pytest.raises(NotImplementedError):
        _recognise.describe(mu)
    assert isinstance(_recognise.bind(mu), GenericPlan) and prec._plan is mu._plan and xs[3]._plan is mu._plan
    # an LDS whose outputs have missing values is still the fused plan's graph (the outputs concerned become nodes of
    # their own) -- unless their initial covariance is not a multiple of the identity
    g, Y, st0, pri = _graph()
    g["Ys"][2].observed = False
    g["Ys"][2].observe(np.array([[1.0], [np.nan], [0.5], [np.nan]]))
    d = _recognise.describe(g["Xs"][0])
    assert d["Ys"][2] is g["Ys"][2] and g["Ys"][2].partially_observed and len(d["Ys"]) == len(d["Xs"])
    g["Ys"][2].qcov = np.diag([1.0, 2.0, 1.0, 1.0])
    with pytest.raises(NotImplementedError):
        _recognise.describe(g["Xs"][0])
    assert isinstance(_recognise.bind(g["Xs"][0]), GenericPlan)
    with pytest.raises(NotImplementedError):
        nodes.Transpose(mu)


def test_recogniser_collects_known_matrix_entries():
    """examples/LDS_knowns_in_A.py:73-74: As[i].observe([[v], [nan]]) reaches the plan as A_obs."""
    g, Y, st0, pri = _graph()
    g["As"][0].observe(np.array([[1.0], [np.nan], [np.nan]]))
    g["Cs"][1].observe(np.array([[0.1], [0.2], [0.3], [0.4]]))
    d = _recognise.describe(g["Xs"][0])
    A_obs, C_obs = d["pri"]["A_obs"], d["pri"]["C_obs"]
    assert A_obs.shape == (3, 3) and C_obs.shape == (4, 3)
    assert A_obs[0, 0] == 1.0 and np.isnan(A_obs).sum() == 8
    assert np.array_equal(C_obs[:, 1], [0.1, 0.2, 0.3, 0.4]) and np.isnan(C_obs).sum() == 8


def test_recogniser_extracts_the_pca_graph():
    """examples/PCA_missing_data.py:31-42 built from pyvb_amd.nodes: data, mask, priors and crawl order."""
    import pyvb_amd
    G = _golden_module()
    init, pri = G.pca_problem(12, 5, 2, seed=3)
    g = G.pca_build_graph(pyvb_amd, init, pri)
    d = _recognise.describe_pca(g["Mu"])
    assert [a is b for a, b in zip(d["Xs"], g["Xs"])] == [True] * 12
    assert [a is b for a, b in zip(d["Zs"], g["Zs"])] == [True] * 12
    assert np.array_equal(d["init"]["obs"], init["obs"])
    for k in ("X", "W_mean", "Z", "Z_cov", "Mu_mean"):
        assert np.array_equal(d["init"][k], init[k]), k
    assert d["init"]["beta_b"] == float(init["beta_b"])
    for k in ("W_prior_mean", "W_prior_prec", "Mu_prior_mean", "Mu_prior_prec"):
        assert np.array_equal(d["pri"][k], pri[k]), k
    g["net"].find_iterable()
    order = g["net"].iterable_nodes
    assert order[:2] == g["Ws"] and order[2:14] == g["Zs"] and order[14] is g["Xs"][0] and order[15] is g["Mu"]
    assert order[16:27] == g["Xs"][1:] and order[27] is g["Beta"]
