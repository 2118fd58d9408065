"""Scenarios for M LDS graphs of one structure built side by side through the node API (the reference iterates any node
list, network.py:46-49; examples/Linear_Dynamic_System.py:46-77 per graph): shared by tests/test_groups_cpu.py (host logic
on the oracle-backed stand-in for the handle) and tests/test_groups_gpu.py (the HIP library)."""
import importlib.util
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))


def golden_module():
    spec = importlib.util.spec_from_file_location("make_golden", os.path.join(HERE, "golden", "make_golden.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def rel(a, b):
    a, b = np.asarray(a, dtype=float), np.asarray(b, dtype=float)
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-300)


def loop_body(g):
    """examples/Linear_Dynamic_System.py:69-77"""
    Xs = g["Xs"]
    [x.update() for x in Xs]
    Xs.reverse()
    [x.update() for x in Xs]
    Xs.reverse()
    [a.update() for a in g["As"]]
    [c.update() for c in g["Cs"]]
    g["Q"].update()
    g["R"].update()


def snapshot(g):
    """Everything the example reads of a graph (reads: the queued requests run)."""
    T = len(g["Xs"])
    out = {"X": np.hstack([x.qmu for x in g["Xs"]]).T, "A": np.hstack([a.qmu for a in g["As"]]), "C": np.hstack([c.qmu for c in g["Cs"]]),
           "Avar": np.stack([np.diag(a.qcov) for a in g["As"]]), "Cvar": np.stack([np.diag(c.qcov) for c in g["Cs"]]),
           "S0": g["Xs"][0].qcov, "S2": g["Xs"][T - 1].qcov, "Qb": np.asarray(g["Q"].qb, dtype=float), "Rb": np.asarray(g["R"].qb, dtype=float)}
    if T > 2:
        out["S1"] = g["Xs"][1].qcov
    return out


def problems(T, D, K, count, pri=None, seed=900):
    from pyvb_amd import synth
    out = []
    for k in range(count):
        Y, st0, p = synth.make_problem(T, D, K, 1, seed + k)
        out.append((Y, st0, pri if pri is not None else p))
    return out


def build(nodes, probs):
    G = golden_module()
    return [G.build_graph(nodes, Y[0], pri, {k: v for k, v in st0.items()}) for Y, st0, pri in probs]


def all_nodes(g):
    return g["Xs"] + g["Ys"] + g["As"] + g["Cs"] + [g["Q"], g["R"]]


def same(a, b, exact, tol=1e-11):
    for k in a:
        if exact:
            assert np.array_equal(a[k], b[k]), k
        else:
            assert rel(a[k], b[k]) <= tol, (k, rel(a[k], b[k]))
