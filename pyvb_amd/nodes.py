"""pyvb's node API on top of the MI355X kernels.

Same classes, constructor signatures, attributes and error behaviour as the reference's
src/pyvb/nodes/{node,gaussian,nodes_todo}.py, so a script such as
examples/Linear_Dynamic_System.py runs unchanged with `from pyvb_amd import nodes`.
The difference is where the arithmetic happens: the nodes only record the graph.  The first
call that needs a posterior (`update()`, reading `qmu`, `Network.learn` ...) hands the connected
graph to a recogniser; if it is the linear-dynamical-system graph of that example the state
moves to the GPU (pyvb_amd.lds.LDSBatch, one replicate) and from then on

  * `node.update()` only queues the request;
  * the queue is flushed when a result is read, and runs of requests that spell a whole sweep --
    `[x.update() for x in Xs]`, its reverse, `[a.update() for a in As]` ... -- become one kernel
    launch each (Linear_Dynamic_System.py:69-77); anything else falls back to per-node launches;
  * `qmu`, `qcov`, `qa`, `qb` ... read back from the device.

Every other graph runs the reference's own schedule, one node at a time, on the device as well: its
methods are restated as emitters of a small tape language (pyvb_amd/generic.py) that a HIP interpreter
executes (pyvb_amd/csrc/k_tape.hip).  There is no CPU execution of updates, messages or expectations in
this package: the `pass_down_*` accessors of the operation nodes (Addition, Multiplication, hstack),
which the reference's examples use for plotting, are evaluated by the same emitters -- on the graph's
own plan if it runs node by node, on a mirror of the fused plan's state otherwise.
"""
import numpy as np

__all__ = ["Node", "Addition", "Multiplication", "Constant", "hstack", "Transpose", "Gaussian",
           "DiagonalGaussian", "Gamma", "DiagonalGamma", "Wishart", "ConjugacyError"]


class ConjugacyError(ValueError):            # nodes_todo.py:8-10
    def __init__(self, message):
        ValueError.__init__(self, message)


# -------------------------------------------------------------------------------------------------
# operation nodes (node.py)
# -------------------------------------------------------------------------------------------------
class Node(object):
    """Base class: shape, children, operator overloads (node.py:6-50)."""
    __array_ufunc__ = None       # let `ndarray * node` reach __rmul__ instead of broadcasting over the array

    def __init__(self, shape):
        self.children = []
        self.shape = shape
        self._plan = None

    def addChild(self, child):
        _graph_changed(self)
        self.children.append(child)

    def update(self):
        pass

    def log_lower_bound(self):
        return 0.

    def pass_up_m1_m2(self, requester):
        """The two messages a parent needs from this node (gaussian.py:179-183, node.py:95-110, :182-232,
        nodes_todo.py:43-62), evaluated on the device; returns numpy arrays like the reference."""
        return _messages_of(self, requester)

    def __add__(self, other):
        return Addition(self, other)

    def __mul__(self, other):
        return Multiplication(self, other)

    def __rmul__(self, other):
        return Multiplication(other, self)


def _wrap(x):
    return Constant(x) if isinstance(x, np.ndarray) else x


def _graph_changed(node):
    """A node gained a child or an observation: the plan that was built for the old graph is stale (the generic plan's
    tapes spell the old message structure, a fused plan was recognised on the old graph).  At the next use its state is
    pulled back into the nodes and the graph is bound again as it is then."""
    plan = getattr(node, "_plan", None)
    if plan is not None:
        # requests made so far refer to the graph as it was: they are issued before it changes (the node-by-node plan keeps
        # update() calls back until a result is needed, and builds a node's tape from its attributes)
        if not getattr(plan, "stale", False) and not getattr(plan, "released", False):
            plan.flush()
        plan.stale = True
        handle = getattr(plan, "group", None)
        if handle is not None:
            handle.epoch += 1           # Network.learn's schedule for this handle is void (network._Schedule.valid)


class Addition(Node):
    """A + B (node.py:52-129).  Arrays are wrapped in Constant before they are linked, which
    the reference intends but does not do."""

    def __init__(self, A, B):
        assert A.shape == B.shape, "Bad shapes for addition"
        Node.__init__(self, A.shape)
        self.A, self.B = _wrap(A), _wrap(B)
        self.A.addChild(self)
        self.B.addChild(self)

    def pass_down_Ex(self):                     # node.py:112-119
        return _expectation_of(self, "Ex")

    def pass_down_ExxT(self):                   # node.py:121-129
        return _expectation_of(self, "ExxT")


class Multiplication(Node):
    """A * B with B a column vector (node.py:131-276)."""

    def __init__(self, A, B):
        m1, n1 = A.shape
        m2, n2 = B.shape
        assert n1 == m2, "incompatible multiplication dimensions"
        assert n2 == 1, "right hand object must be a vector"
        Node.__init__(self, (m1, n2))
        self.A, self.B = _wrap(A), _wrap(B)
        self.A.addChild(self)
        self.B.addChild(self)

    def pass_down_Ex(self):                     # node.py:235-242
        return _expectation_of(self, "Ex")

    def pass_down_ExxT(self):
        """node.py:244-276, all five branches -- column x scalar, row vector, Constant matrix (with the transpose the
        reference's line lacks), hstack, DiagonalGaussian -- through the emitter GenericPlan._exxt."""
        r = _expectation_of(self, "ExxT")
        if self.A.shape[0] == 1 and self.A.shape[1] != 1:
            return float(r.reshape(-1)[0])      # the row-vector branch is a trace: the reference returns a scalar
        return r


class Constant(Node):
    """A fixed array (node.py:279-311)."""

    def __init__(self, value):
        Node.__init__(self, value.shape)
        self.value = value
        self.value_xxT = np.dot(value, value.T)
        self.value_xTx = np.dot(value.T, value)
        if self.shape[0] == self.shape[1]:
            with np.errstate(divide="ignore", invalid="ignore"):
                self.lndet = np.log(np.linalg.det(self.value))

    def pass_down_Ex(self):
        return self.value

    def pass_down_ExxT(self):
        return self.value_xxT

    def pass_down_ExTx(self):
        return self.value_xTx

    def pass_down_lndet(self):
        return self.lndet


class hstack(Node):
    """A matrix whose columns are Gaussian nodes (nodes_todo.py:12-62)."""

    def __init__(self, parents):
        assert type(parents) == list
        dims = [e.shape[0] for e in parents]
        assert np.all(dims[0] == np.array(dims)), "dimensions incompatible"
        Node.__init__(self, (dims[0], len(parents)))
        self.parents = parents
        [e.addChild(self) for e in self.parents]

    def pass_down_Ex(self):                     # nodes_todo.py:33-34
        return _expectation_of(self, "Ex")

    def pass_down_ExxT(self):                   # nodes_todo.py:36-38
        return _expectation_of(self, "ExxT")

    def pass_down_ExTx(self):
        raise NotImplementedError


class Transpose(Node):
    """The reference's Transpose raises NameError on construction; there is no such path."""

    def __init__(self, parent):
        raise NotImplementedError("Transpose is broken in the reference (nodes_todo.py:65-72) and has no HIP path")


# -------------------------------------------------------------------------------------------------
# random-variable nodes
# -------------------------------------------------------------------------------------------------
def _plan_of(node):
    plan = node._plan
    if plan is not None and getattr(plan, "stale", False):
        plan.release()                  # device state back into the nodes, then bind the graph as it is now
    if node._plan is None:
        from . import _recognise
        _recognise.bind(node)
    return node._plan


def _generic_view(node):
    """The generic (tape) plan that can evaluate single messages / single lower-bound terms / expectations for `node`: its own
    plan if the graph runs node by node, else a mirror of the fused plan's current state.  Issuing the fused plan's queued
    requests can itself hand the graph to the node-by-node plan (a request no fused kernel serves): looked at again after
    the flush."""
    for _ in range(4):
        plan = _plan_of(node)
        if getattr(plan, "generic", False):
            return plan
        plan.flush()
        if node._plan is plan and not getattr(plan, "stale", False):
            return plan.mirror()
    raise RuntimeError("the graph kept changing plans")


def _messages_of(node, requester):
    return _generic_view(node).message(node, requester)


def _expectation_of(node, what):
    """node.pass_down_<what>() of an operation node, evaluated on the device (GenericPlan.expectation)."""
    return _generic_view(node).expectation(node, what)


class _DeviceAttr(object):
    """Attribute that lives on the host until the graph is bound, on the device afterwards."""

    def __init__(self, name):
        self.name = name

    def __get__(self, obj, objtype=None):
        if obj is None:
            return self
        if obj._plan is not None:
            return _plan_of(obj).read(obj, self.name)
        return obj.__dict__.get("_h_" + self.name)

    def __set__(self, obj, value):
        # The assignment must win.  A bound plan either patches its device state (write() returns True) or gives the graph
        # up: its state goes back into the nodes, the value is stored below, and the next use binds the graph anew.  A
        # release can itself leave the graph with another plan (queued requests that only the node-by-node plan serves;
        # sweeps among them that bring the fused plan back), bound from the OLD host value: hence the loop.
        for _ in range(8):
            plan = getattr(obj, "_plan", None)
            if plan is None:
                break
            if getattr(plan, "stale", False):
                plan.release()
                continue
            if plan.write(obj, self.name, value):
                break
            plan.release()
        else:
            raise RuntimeError("the graph kept changing plans while %s was assigned" % self.name)
        obj.__dict__["_h_" + self.name] = value


class Gaussian(Node):
    """q(x) = N(qmu, qcov) with a Gaussian-family mean parent and a Gamma-family precision parent
    (gaussian.py:9-183)."""
    qmu = _DeviceAttr("qmu")
    qcov = _DeviceAttr("qcov")
    q_ln_det = _DeviceAttr("q_ln_det")

    def __init__(self, dim, pmu, pprec):
        Node.__init__(self, (dim, 1))
        assert pmu.shape == self.shape, "Parent node (or array) has incorrect dimension"
        if type(pmu) == np.ndarray:
            self.mean_parent = Constant(pmu)
        elif isinstance(pmu, (Gaussian, Addition, Multiplication, Constant)):
            self.mean_parent = pmu
        else:
            raise ConjugacyError("mean parent for a Gaussian node should be one of:\nGaussian\nConstant\nAddition\n"
                                 "Multiplication\nnumpy array. \n\n" + str(type(pmu)) + " is invalid")
        assert pprec.shape == (self.shape[0], self.shape[0]), "Parent precision array has incorrect dimension"
        if type(pprec) == np.ndarray:
            self.precision_parent = Constant(pprec)
        elif isinstance(pprec, (Gamma, DiagonalGamma, Wishart, Constant)):
            self.precision_parent = pprec
        else:
            raise ConjugacyError("Precision parent for a Gaussian node should be one of:\nGamma\nDiagonalGamma\nWishart\n"
                                 "Constant\nnumpy array. \n\n" + str(type(pprec)) + " is invalid")
        self.mean_parent.addChild(self)
        self.precision_parent.addChild(self)
        self.observed = False
        self.partially_observed = False
        # random initial posterior, gaussian.py:70-72
        self.qmu = np.random.randn(self.shape[0], 1)
        self.qprec = np.eye(self.shape[0]) * np.random.rand()
        self.qcov = np.linalg.inv(self.qprec)

    def observe(self, val):                     # gaussian.py:74-100
        assert val.shape == self.shape, "Bad shape for observation data"
        _graph_changed(self)
        if np.isnan(val).all():
            return
        elif np.isnan(val).any():
            self.partially_observed = True
            self.obs_value = val
            self.obs_index = np.nonzero(1 - np.isnan(val))[0]
            self.missing_index = np.nonzero(np.isnan(val))[0]
        else:
            self.observed = True
            self.qmu = val.copy()               # the reference keeps a reference to the caller's array (Q10)
            self.qcov = np.zeros((self.shape[0], self.shape[0]))

    def update(self):                           # gaussian.py:102-134
        if self.observed:
            return
        _plan_of(self).enqueue(self)

    def log_lower_bound(self):
        return _plan_of(self).node_llb(self)

    def pass_up_m1_m2(self, requester):         # gaussian.py:179-183
        return _messages_of(self, requester)

    def pass_down_Ex(self):                     # gaussian.py:154-160: the posterior mean as it is
        return self.qmu

    def pass_down_ExxT(self):                   # gaussian.py:162-168, on the device (GenericPlan._exxt)
        return _expectation_of(self, "ExxT")

    def pass_down_ExTx(self):                   # gaussian.py:170-177
        return float(np.asarray(_expectation_of(self, "ExTx")).reshape(-1)[0])


class DiagonalGaussian(Gaussian):               # gaussian.py:185-203
    def __init__(self, dim, pmu, pprec):
        Gaussian.__init__(self, dim, pmu, pprec)
        self.shape = (self.shape[0], self.shape[0])

    def pass_down_Ex(self):                     # gaussian.py:198-199
        return _expectation_of(self, "Ex")

    def pass_down_ExxT(self):                   # gaussian.py:200-201
        return _expectation_of(self, "ExxT")

    def pass_down_ExTx(self):
        return self.pass_down_ExxT()


class _NoiseNode(object):
    """Shared behaviour of Gamma, DiagonalGamma and Wishart (nodes_todo.py:88-234)."""
    qb = _DeviceAttr("qb")

    def _init_common(self, dim):
        self.shape = (dim, dim)
        self.children = []
        self._plan = None

    def update(self):
        _plan_of(self).enqueue(self)

    def log_lower_bound(self):
        return _plan_of(self).node_llb(self)


class Gamma(_NoiseNode):                        # nodes_todo.py:88-157
    def __init__(self, dim, a0, b0):
        self._init_common(dim)
        self.a0, self.b0 = a0, b0
        self.update_a()
        self.qb = np.random.rand()

    def addChild(self, child):
        _graph_changed(self)
        self.children.append(child)
        self.update_a()

    def update_a(self):
        self.qa = self.a0
        for child in self.children:
            self.qa += 0.5 * child.shape[0]

    def pass_down_Ex(self):
        return np.eye(self.shape[0]) * self.qa / self.qb

    def pass_down_lndet(self):
        return self.shape[0] * (np.log(self.qa) - np.log(self.qb))


class DiagonalGamma(_NoiseNode):                # nodes_todo.py:159-204
    def __init__(self, dim, a0s, b0s):
        self._init_common(dim)
        assert a0s.size == self.shape[0]
        assert b0s.size == self.shape[0]
        self.a0s = a0s.flatten()
        self.b0s = b0s.flatten()
        self.update_a()
        self.qb = np.random.rand()

    def addChild(self, child):
        assert child.shape == (self.shape[0], 1)
        _graph_changed(self)
        self.children.append(child)
        self.update_a()

    def update_a(self):
        self.qa = self.a0s.copy()
        for child in self.children:
            self.qa += 0.5

    def pass_down_Ex(self):
        return np.diag(self.qa / self.qb)

    def pass_down_lndet(self):
        return np.log(np.prod(self.qa / self.qb))


class Wishart(_NoiseNode):                      # nodes_todo.py:205-234
    """Wishart(dim, v0, w0): E[Lambda] = qv * inv(qw), qv = v0 + 1/2 per child, qw = w0 + sum over the children of
    1/2<x x^T> + 1/2<mu mu^T> - <x><mu>^T.  The reference class is unfinished; the device path
    (pyvb_amd/csrc/k_wishart.hip) keeps its update formula and departs from it where it is broken: the prior is not
    mutated by update(), the expectation uses the symmetric part of qw, and pass_down_lndet / log_lower_bound exist.
    As in the reference, the random initial qw is rank one: reading E[Lambda] before the first update() (or before
    assigning a positive definite qw) raises numpy.linalg.LinAlgError."""
    qw = _DeviceAttr("qw")

    def __init__(self, dim, v0, w0):
        self._init_common(dim)
        assert w0.shape == self.shape
        self.v0, self.w0 = v0, w0
        self.update_v()
        l = np.random.randn(self.shape[0], 1)
        self.qw = np.dot(l, l.T)

    def addChild(self, child):
        assert child.shape == (self.shape[0], 1)
        _graph_changed(self)
        self.children.append(child)
        self.update_v()

    def update_v(self):
        self.qv = self.v0
        for child in self.children:
            self.qv += 0.5

    def pass_down_Ex(self):                     # nodes_todo.py:233-234, on the device (GenericPlan._ex: the symmetric part of qw)
        return _expectation_of(self, "Ex")

    def pass_down_lndet(self):                  # not in the reference (Q8): ln det of the expectation, as Gamma does (Q2)
        return float(np.asarray(_expectation_of(self, "lndet")).reshape(-1)[0])
