"""Graph recogniser and execution plan behind pyvb_amd.nodes.

`bind(node)` walks the graph a node belongs to.  If it is the linear-dynamical-system graph of
the reference's examples/Linear_Dynamic_System.py:46-66 --

    A = hstack(D Gaussian columns, Constant parents)      C = hstack(D Gaussian columns)
    X_0 ~ N(Constant, Constant)      X_t ~ N(A * X_{t-1}, Q)      Y_t ~ N(C * X_t, R), observed
    Q, R both DiagonalGamma or both Gamma

-- it builds an `LDSPlan`: the graph becomes one replicate of an LDSBatch, which it shares with every other
graph of the same structure that is waiting to run (LDSGroup), and every node of the graph is pointed at the
plan.  The VB-PCA graph of examples/PCA_missing_data.py gets a `PCAPlan`; any other graph -- `describe` signals
it with NotImplementedError, which `bind` catches -- the node-by-node plan of pyvb_amd/generic.py.  All of them
run on the device: there is no CPU execution path.
"""
import hashlib
import weakref

import numpy as np

from . import nodes as N


# Requests as the queue holds them: what update() records -- ("x", t), ("y", t), ("a", i), ("c", i), ("q", 0), ("r", 0) --
# and what those spell for the whole batch, which Network.learn replays directly from its second iteration on:
# ("F",) forward sweep, ("B",) backward sweep, ("Y",) the outputs that are not observed, ("A", lo, hi) / ("C", lo, hi)
# columns lo..hi-1 in order, ("Q",), ("R",).
_BATCH_OPS = ("F", "B", "Y", "A", "C", "Q", "R")

_pool = {}          # signature -> weak references to LDSPlans that are bound but not yet on the device, in binding order


class LDSGroup(object):
    """ONE LDSBatch for M graphs of the same structure: replicate r is the graph of members[r].

    The reference iterates any node list (network.py:46-49), so a user with M independent LDS graphs writes M graphs and
    one loop.  The kernels' parallel axis is the replicate axis of pyvb_amd.lds.LDSBatch; graphs that agree in everything
    a handle shares between its replicates -- T, D, K, the noise family, the Constant parents (priors), the known entries of
    A / C, which outputs are unobserved -- therefore share one handle.  The queued update() requests of the members are
    carried out in lock step: when a member needs its queue run (something of it is read), every operation of that queue
    is one launch for all members that ask for the same thing at that point (a forward sweep, the columns 0..D-1 of A ...);
    what the others have queued beyond it waits for their own reads.  Where the member and the others part ways the
    smaller side leaves the handle with its state (the replicates stay behind as dead rows that nobody reads) and is bound
    anew -- on its own, or together with the others that left the same way.  A request no fused kernel serves sends only
    the graph that made it to the node-by-node plan."""

    def __init__(self, members):
        from .lds import LDSBatch
        p0 = members[0]
        self.members = list(members)
        self.T, self.D, self.K, self.kind = p0.T, p0.D, p0.K, p0.kind
        M, T, D, K = len(members), self.T, self.D, self.K
        self.batch = b = LDSBatch(M, T, D, K, self.kind)
        b.set_priors(p0.pri)
        host = [m._host_state() for m in members]
        stack = lambda key: np.stack([h[key] for h in host])
        b.set_observations(stack("Y"))
        if p0.free_ys:
            b.set_output_state(stack("Yq"), stack("Yv"))
        state = {k: stack(k) for k in ("X", "A_mean", "A_colvar", "C_mean", "C_colvar")}
        if self.kind == "wishart":
            b.set_state(**state)
            b.set_wishart_state(stack("Q_w"), stack("R_w"))
        else:
            b.set_state(Q_b=stack("Q_b"), R_b=stack("R_b"), **state)
        if p0.pri.get("A_obs") is not None:
            b.set_column_observations(p0.pri["A_obs"], p0.pri["C_obs"])
        # members that come from another handle bring the covariances their states were last updated with
        self.x_updated = p0.classes is not None
        if self.x_updated:
            b.set_posterior_classes(np.stack([m.classes[0] for m in members]), np.stack([m.classes[1] for m in members]))
        self.ran = False                # anything at all has run on the device (the outputs can update before any sweep)
        self.cache = None
        self._elbo = None
        self.epoch = 0                  # counts the members that have left (Network.learn's schedule looks at it)
        for r, m in enumerate(members):
            m.group, m.r = self, r

    def live(self):
        return [m for m in self.members if m is not None]

    def invalidate(self):
        self.cache = None
        self._elbo = None

    def drop(self, member):
        """`member` has left (its replicate keeps being computed with the others; nobody reads it)."""
        self.members[member.r] = None
        member.group = None
        self.epoch += 1
        if not any(m is not None for m in self.members):
            self.batch.close()

    # -- the queue -------------------------------------------------------------------------------
    def execute(self, op):
        b = self.batch
        k = op[0]
        if k == "F":
            b.sweep("forward"); self.x_updated = True
        elif k == "B":
            b.sweep("backward"); self.x_updated = True
        elif k == "Y":
            b.update_Y()
        elif k in ("A", "C"):
            b.update_columns(k, op[1], op[2])
        elif k == "Q":
            b.update_Q()
        else:
            b.update_R()
        self.ran = True
        self.invalidate()

    def flush(self, who):
        """Carry out the queued requests of member `who` -- and, in the same launches, those of every member that is asking for
        the same things (see the class comment).  Nothing else is run: what the other members have queued beyond that waits
        for their own reads.  When `who` and the others part ways the smaller side leaves the handle: `who` and the members
        that go with it, or everybody else."""
        try:
            while who.group is self:
                op, n = who._peek()
                if n == 0:
                    return
                if op is None:                      # a request no fused kernel serves: this graph goes node by node
                    who._depart(fused=False)
                    return
                same, rest = [(who, n)], []
                for m in self.members:
                    if m is None or m is who:
                        continue
                    o2, n2 = m._peek()
                    if n2 and o2 == op:
                        same.append((m, n2))
                    else:
                        rest.append((m, n2))
                if len(same) < len(rest):
                    for m, _ in same:               # `who` among them: the loop ends, the caller goes on with its new plan
                        m._depart(fused=True)
                    return
                for m, n2 in rest:
                    if n2:
                        m._depart(fused=True)
                    else:
                        m._leave()
                self.execute(op)
                for m, k in same:
                    m._pos += k
        finally:
            for m in self.members:
                if m is not None and m._pos:
                    del m.pending[:m._pos]
                    m._pos = 0

    def run_script(self, script):
        """The same batch operations for every live member at once (Network.learn from its second iteration on); the caller
        has checked that no member has anything queued."""
        for op in script:
            if op[0] in ("A", "C", "Q", "R") and not self.x_updated:
                return False
        for op in script:
            self.execute(op)
        return True

    # -- results ---------------------------------------------------------------------------------
    def pull(self):
        if self.cache is None:
            b = self.batch
            st = b.get_state()
            Sig, qld = b.get_posterior_classes()
            qa, qc = b.get_column_qld()
            c = {"st": st, "Sigma": Sig, "qld_x": qld, "qld_A": qa, "qld_C": qc}
            if self.kind == "wishart":
                c["w"] = b.get_wishart_state()
                c["A_cov"], c["C_cov"] = b.get_column_cov()
            if self.members and any(m is not None and m.free_ys for m in self.members):
                c["Yq"], c["Yvar"], c["Yqld"] = b.get_outputs(with_qld=True)
            self.cache = c
        return self.cache

    def elbo(self):
        if self._elbo is None:
            self._elbo = self.batch.elbo()
        return self._elbo


class LDSPlan(object):
    """One LDS graph on the fused kernels: replicate `r` of its group's handle (LDSGroup)."""
    resume_left = 3         # how often a graph handed to the generic plan may still come back (GenericPlan._resume_fused)
    _count = 0

    def __init__(self, Xs, Ys, As, Cs, A, C, Q, R, pri, classes=None):
        self.Xs, self.Ys, self.As, self.Cs, self.A, self.C, self.Q, self.R = Xs, Ys, As, Cs, A, C, Q, R
        self.T, self.D, self.K = len(Xs), As[0].shape[0], Cs[0].shape[0]
        T = self.T
        self.pri = pri
        self.kind = pri["noise"]
        self.classes = classes          # (Sigma[3,D,D], q_ln_det[3]) of the states' last update on another handle, or None
        self.free_ys = [t for t, y in enumerate(Ys) if not y.observed]      # outputs with missing entries: nodes of their own
        self.index = {}
        for t, x in enumerate(Xs):
            self.index[id(x)] = ("x", t)
        for t, y in enumerate(Ys):
            self.index[id(y)] = ("y", t)
        for i, a in enumerate(As):
            self.index[id(a)] = ("a", i)
        for i, c in enumerate(Cs):
            self.index[id(c)] = ("c", i)
        self.index[id(Q)] = ("q", 0)
        self.index[id(R)] = ("r", 0)
        self._fwd = [("x", t) for t in range(T)]
        self._bwd = self._fwd[::-1]
        self._ys = [("y", t) for t in self.free_ys]
        self.pending, self._pos = [], 0
        self.group, self.r = None, -1
        self.stale = False              # set when a node of the graph gains a child or an observation after binding
        self.closed = False
        self.n_random_nodes = 2 * self.T + 2 * self.D + 2
        self.sig = _signature(self)
        LDSPlan._count += 1
        self._serial = LDSPlan._count
        waiting = [ref for ref in _pool.get(self.sig, []) if ref() is not None]
        _pool[self.sig] = waiting + [weakref.ref(self)]
        for n in _component(Xs[0]):         # the operation nodes and Constants too: their messages go through mirror()
            n._plan = self

    # -- the handle ------------------------------------------------------------------------------
    def _host_state(self):
        """The graph's posteriors as the nodes hold them on the host, in the C ABI's layout (one replicate)."""
        Xs, Ys, As, Cs, Q, R = self.Xs, self.Ys, self.As, self.Cs, self.Q, self.R
        T, D, K = self.T, self.D, self.K
        h = {"Y": np.hstack([y.__dict__["_h_qmu"] if y.observed else
                             (y.obs_value if y.partially_observed else np.full((K, 1), np.nan)) for y in Ys]).T.reshape(T, K)}
        if self.free_ys:
            h["Yq"] = np.hstack([y.__dict__["_h_qmu"] for y in Ys]).T.reshape(T, K)
            h["Yv"] = np.array([y.__dict__["_h_qcov"][0, 0] for y in Ys]).reshape(T)
        diag = lambda nodes_: np.stack([np.diag(n.__dict__["_h_qcov"]) for n in nodes_])
        qb = lambda nd, dim: np.broadcast_to(np.asarray(nd.__dict__["_h_qb"], dtype=float), (dim,)).reshape(dim)
        h.update(X=np.hstack([x.__dict__["_h_qmu"] for x in Xs]).T.reshape(T, D),
                 A_mean=np.hstack([a.__dict__["_h_qmu"] for a in As]).reshape(D, D), A_colvar=diag(As).reshape(D, D),
                 C_mean=np.hstack([c.__dict__["_h_qmu"] for c in Cs]).reshape(K, D), C_colvar=diag(Cs).reshape(D, K))
        if self.kind == "wishart":
            h["Q_w"] = np.asarray(Q.__dict__["_h_qw"], dtype=float).reshape(D, D)
            h["R_w"] = np.asarray(R.__dict__["_h_qw"], dtype=float).reshape(K, K)
        else:
            h["Q_b"], h["R_b"] = qb(Q, D), qb(R, K)
        return h

    def _materialize(self):
        """State to the device -- together with every other graph of the same signature that is bound, not yet on the device
        and waiting with the same first request (graphs that would part ways at once are not put on one handle)."""
        if self.group is not None:
            return self.group
        peers, later = [], []
        want = self._peek()
        for ref in _pool.pop(self.sig, []):
            p = ref()
            if p is None or p is self or p.group is not None or p.closed or p.stale or p.Xs[0]._plan is not p:
                continue
            (peers if want[1] and p._peek() == want else later).append(p)
        if later:
            _pool[self.sig] = [weakref.ref(p) for p in later]
        # binding order (the pool's), this graph at its own place
        peers.append(self)
        peers.sort(key=lambda p: p._serial)
        LDSGroup(peers)
        return self.group

    @property
    def batch(self):
        return self._materialize().batch

    @property
    def x_updated(self):
        return self.group.x_updated if self.group is not None else self.classes is not None

    @property
    def ran(self):
        return self.group is not None and self.group.ran

    @property
    def cache(self):
        return self.group.cache if self.group is not None else None

    # -- queue -----------------------------------------------------------------------------------
    def enqueue(self, node):
        self.pending.append(self.index[id(node)])

    def _peek(self, gate=True):
        """(op, n): the head of the queue as one operation on the batch that takes n entries off it -- the fused kernels
        serve whole sweeps, `[x.update() for x in Xs]` and its reverse, and, once the states have been swept, runs of column
        updates and the noise updates.  (None, n > 0): a request they do not serve (a single X_t.update(); parameters
        before the first sweep: the X_t then still have their individual initial covariances, gaussian.py:70-72) -- the
        graph goes to the node-by-node plan.  (None, 0): nothing queued."""
        ops, i = self.pending, self._pos
        while i < len(ops) and ops[i][0] not in ("x", "y", "a", "c", "q", "r") + _BATCH_OPS:
            i += 1                      # nothing to do for anything else (observed nodes never update, gaussian.py:109-110)
        self._pos = i
        if i >= len(ops):
            return None, 0
        head = ops[i]
        kind = head[0]
        op, n = None, 1
        if kind in _BATCH_OPS:
            op = head
        elif kind == "x":
            run = ops[i:i + self.T]
            if run == self._fwd:
                op, n = ("F",), self.T
            elif run == self._bwd:
                op, n = ("B",), self.T
        elif kind == "y":               # [y.update() for y in Ys if not y.observed]: all of them, in order, or node by node
            if ops[i:i + len(self._ys)] == self._ys:
                op, n = ("Y",), len(self._ys)
        elif kind in ("a", "c"):
            j = i
            while j + 1 < len(ops) and ops[j + 1] == (kind, ops[j][1] + 1):
                j += 1
            op, n = (kind.upper(), head[1], ops[j][1] + 1), j + 1 - i
        else:
            op = (kind.upper(),)
        if gate and op is not None and op[0] in ("A", "C", "Q", "R") and not self.x_updated:
            op = None
        return op, n

    def _spell(self, first):
        """pending[first:] as batch operations (what Network.learn replays), None if some request there is not one the
        fused kernels serve.  Nothing is taken off the queue."""
        keep, out = self._pos, []
        self._pos = first
        try:
            while True:
                op, n = self._peek(gate=False)
                if n == 0:
                    return out
                if op is None:
                    return None
                out.append(op)
                self._pos += n
        finally:
            self._pos = keep

    def _nodes_of(self, entry):
        """The nodes whose update() a queue entry stands for, in order."""
        k = entry[0]
        if k == "F":
            return list(self.Xs)
        if k == "B":
            return self.Xs[::-1]
        if k == "Y":
            return [self.Ys[t] for t in self.free_ys]
        if k in ("A", "C"):
            return (self.As if k == "A" else self.Cs)[entry[1]:entry[2]]
        if k in ("Q", "R"):
            return [self.Q if k == "Q" else self.R]
        return [{"x": self.Xs, "y": self.Ys, "a": self.As, "c": self.Cs, "q": [self.Q], "r": [self.R]}[k][entry[1]]]

    def flush(self):
        """Run the queued update() requests (LDSGroup.flush: in lock step with the other graphs on this handle)."""
        if self._pos >= len(self.pending):
            return
        self._materialize().flush(self)

    def _rest(self):
        rest, self.pending, self._pos = self.pending[self._pos:], [], 0
        return rest

    def _depart(self, fused):
        """Leave the group with the queue: to a fused plan of its own (bound anew from the state as it is now, the
        requests moved over) or, for a request the fused kernels do not serve, to the node-by-node plan."""
        if not fused:
            return self._demote(self._rest())
        rest, left = self._rest(), self.resume_left
        self._leave()
        plan = bind(self.Xs[0])
        plan.resume_left = left
        for entry in rest:
            if isinstance(plan, LDSPlan):
                plan.pending.append(entry)
            else:
                for nd in self._nodes_of(entry):
                    nd._plan.enqueue(nd)
        return plan

    def _sync_host(self):
        """Current device posteriors into the nodes' host attributes."""
        if not self.ran:
            return              # nothing has run: the host attributes are the state
        self._pull()
        for nd in self.Xs + self.As + self.Cs + [self.Ys[t] for t in self.free_ys]:
            for name in ("qmu", "qcov"):
                nd.__dict__["_h_" + name] = self.read(nd, name)
            try:
                q = self.read(nd, "q_ln_det")
                if np.isfinite(q):
                    nd.__dict__["_h_q_ln_det"] = q
            except Exception:
                pass
        if self.kind == "wishart":
            self.Q.__dict__["_h_qw"], self.R.__dict__["_h_qw"] = self.read(self.Q, "qw"), self.read(self.R, "qw")
        else:
            self.Q.__dict__["_h_qb"], self.R.__dict__["_h_qb"] = self.read(self.Q, "qb"), self.read(self.R, "qb")

    def _leave(self):
        """Device state back into the nodes, the graph unbound, the replicate given up."""
        self._sync_host()
        for n in _component(self.Xs[0]):
            if n._plan is self:
                n._plan = None
        self.closed = True
        if self.group is not None:
            self.group.drop(self)

    def release(self):
        """Device state back into the nodes and the graph unbound (it is bound anew, as it is now, at the next use)."""
        self.flush()
        if self.Xs[0]._plan is not self:
            return
        self._leave()

    def _demote(self, rest):
        """Hand the graph to the generic node-by-node plan and replay the remaining update() requests there."""
        from .generic import GenericPlan
        self._sync_host()
        for n in _component(self.Xs[0]):
            n._plan = None
        self.closed = True
        if self.group is not None:
            self.group.drop(self)
        gp = GenericPlan(self.Xs[0])
        if self.resume_left > 0:        # back to the fused kernels when a forward and a backward sweep follow each other again
            gp.resume = {"pattern": self.Xs + self.Xs[::-1], "left": self.resume_left}
        for entry in rest:
            for node in self._nodes_of(entry):
                if not getattr(node, "observed", False):
                    node._plan.enqueue(node)        # gp -- or the fused plan again, if these requests brought the graph back to it
        return gp

    # -- attribute traffic -----------------------------------------------------------------------
    def _pull(self):
        self.flush()
        return self._materialize().pull()

    def read(self, node, name):
        kind, i = self.index[id(node)]
        if kind == "y" and (node.observed or name not in ("qmu", "qcov", "q_ln_det")):
            return node.__dict__.get("_h_" + name)          # observations: host copy
        self.flush()
        if node._plan is not self:
            return N._plan_of(node).read(node, name)
        if not self.ran:
            return node.__dict__.get("_h_" + name)          # nothing has run on the device: the host attributes are the state
        if kind == "x" and name != "qmu" and not self.x_updated:
            return node.__dict__.get("_h_" + name)          # never updated: the constructor's draw (gaussian.py:70-72)
        c = self._pull()
        st, T, r = c["st"], self.T, self.r
        if kind == "y":
            if name == "q_ln_det":
                return float(c["Yqld"][r, i]) if np.isfinite(c["Yqld"][r, i]) else node.__dict__.get("_h_q_ln_det")
            return c["Yq"][r, i].reshape(-1, 1).copy() if name == "qmu" else np.diag(c["Yvar"][r, i])
        if kind == "x":
            cls = 0 if i == 0 else (2 if i == T - 1 else 1)
            if name == "qmu":
                return st["X"][r, i].reshape(-1, 1).copy()
            return c["Sigma"][r, cls].copy() if name == "qcov" else float(c["qld_x"][r, cls])
        if kind in ("a", "c"):
            M, V, q = ("A_mean", "A_colvar", "qld_A") if kind == "a" else ("C_mean", "C_colvar", "qld_C")
            if name == "qmu":
                return st[M][r][:, [i]].copy()
            if name == "qcov" and self.kind == "wishart":
                return c["A_cov" if kind == "a" else "C_cov"][r, i].copy()
            return np.diag(st[V][r, i]) if name == "qcov" else float(c[q][r, i])
        if name == "qw" and self.kind == "wishart":
            return c["w"]["Q_w" if kind == "q" else "R_w"][r].copy()
        if name == "qb" and self.kind != "wishart":
            v = st["Q_b" if kind == "q" else "R_b"][r]
            return float(v[0]) if self.kind == "gamma" else v.copy()
        raise AttributeError(name)

    def write(self, node, name, value):
        """A user assignment to a posterior attribute after binding: push it to the device.  False: not something this plan
        can patch in place (an observation, a state covariance): the caller releases the plan and the graph is bound anew."""
        self.flush()
        if node._plan is not self:              # the queue held something only the node-by-node plan serves, or the graph left its handle
            return N._plan_of(node).write(node, name, value)
        if name not in ("qmu", "qcov", "qb", "qw"):
            return True                         # q_ln_det, qprec: host-only bookkeeping
        kind, i = self.index[id(node)]
        if not ((kind == "x" and name == "qmu") or kind in ("a", "c") or (kind in ("q", "r") and name in ("qb", "qw"))):
            return False
        if self.group is None:                  # still on the host: the caller stores the value, the handle is made from it
            return name != "qcov"               # (a covariance is something the recogniser has to look at again)
        g, r, b = self.group, self.r, self.group.batch
        g.invalidate()
        st = b.get_state()
        if kind == "x" and name == "qmu":
            st["X"][r, i] = np.asarray(value).reshape(-1)
            b.set_state(X=st["X"])
        elif kind in ("a", "c") and name in ("qmu", "qcov"):
            M, V = ("A_mean", "A_colvar") if kind == "a" else ("C_mean", "C_colvar")
            if name == "qmu":
                st[M][r][:, i] = np.asarray(value).reshape(-1)
                b.set_state(**{M: st[M]})                       # the mean alone: with Wishart noise the covariances are dense
            elif self.kind == "wishart":
                covs = list(b.get_column_cov())
                covs[0 if kind == "a" else 1][r, i] = np.asarray(value, dtype=float)
                b.set_column_cov(**{"A_cov" if kind == "a" else "C_cov": covs[0 if kind == "a" else 1]})
            else:
                cov = np.asarray(value, dtype=float)
                if np.abs(cov - np.diag(np.diag(cov))).max() != 0.0:
                    return False                                # the fused kernels keep diagonal column covariances here
                st[V][r, i] = np.diag(cov)
                b.set_state(**{V: st[V]})
        elif kind in ("q", "r") and name == "qw":
            w = b.get_wishart_state()
            key = "Q_w" if kind == "q" else "R_w"
            w[key][r] = np.asarray(value, dtype=float).reshape(node.shape)
            b.set_wishart_state(**{key: w[key]})
        elif kind in ("q", "r") and name == "qb":
            key = "Q_b" if kind == "q" else "R_b"
            st[key][r] = np.broadcast_to(np.asarray(value, dtype=float), st[key][r].shape)
            b.set_state(**{key: st[key]})
        return True

    # -- lower bound -----------------------------------------------------------------------------
    def elbo_parts(self):
        self.flush()
        if self.Xs[0]._plan is not self:        # the graph has moved: to a handle of its own, or to the node-by-node plan
            plan = N._plan_of(self.Xs[0])
            if isinstance(plan, LDSPlan):
                return plan.elbo_parts()
            raise NotImplementedError("the graph runs node by node now: use Network.learn or the nodes' log_lower_bound()")
        return self._materialize().elbo()[self.r]

    def node_llb(self, node):
        self.flush()
        if node._plan is not self:
            return N._plan_of(node).node_llb(node)
        if not self.x_updated:                  # single terms before any sweep: only the generic plan knows the initial covariances
            self._demote(self._rest())
            return N._plan_of(node).node_llb(node)
        kind, _ = self.index[id(node)]
        if kind == "q":
            return float(self.elbo_parts()[4])
        if kind == "r":
            return float(self.elbo_parts()[5])
        # a single state / output / column node: the fused kernels only form class sums, so the term is evaluated by the
        # generic tape path on a mirror of the current state (gaussian.py:136-151)
        return self.mirror().node_llb(node)

    def mirror(self):
        """A generic (node-by-node) plan holding a copy of this plan's current posteriors: serves single messages
        (pass_up_m1_m2) and single lower-bound terms, which the fused kernels never materialise."""
        self.flush()
        if getattr(self, "_mirror", None) is None or self._mirror_of is not self.cache or self.cache is None:
            from .generic import GenericPlan
            c = self._pull()
            self._sync_host()
            if getattr(self, "_mirror", None) is not None:
                self._mirror.ex and self._mirror.ex.close()
            self._mirror = GenericPlan(self.Xs[0], adopt=False)
            self._mirror_of = c
        return self._mirror


def _signature(p):
    """Everything the graphs on one handle have in common (include/pyvb_hip.h: the Constant parents, the known entries of
    A / C and the set of unobserved outputs are per handle, not per replicate)."""
    h = hashlib.sha1()
    for k in sorted(p.pri):
        v = p.pri[k]
        h.update(k.encode())
        h.update(v.encode() if isinstance(v, str) else b"-" if v is None else np.ascontiguousarray(np.asarray(v, dtype=float)).tobytes())
    return (p.T, p.D, p.K, p.kind, tuple(p.free_ys), p.classes is not None, h.hexdigest())


# -------------------------------------------------------------------------------------------------
def _component(start):
    """All nodes connected to `start` (parents and children), in discovery order."""
    seen, order, stack = set(), [], [start]
    while stack:
        n = stack.pop()
        if id(n) in seen:
            continue
        seen.add(id(n))
        order.append(n)
        nxt = list(getattr(n, "children", []))
        for attr in ("mean_parent", "precision_parent", "A", "B"):
            if hasattr(n, attr):
                nxt.append(getattr(n, attr))
        nxt.extend(getattr(n, "parents", []))
        stack.extend(nxt)
    return order


def _fail(why):
    """Not the graph a fused plan serves: `bind` catches this and gives the graph the node-by-node plan."""
    raise NotImplementedError("not a graph of the fused kernels (%s)" % why)


def _diag_constant(node, what):
    if not isinstance(node, N.Constant):
        _fail("%s must be a Constant" % what)
    v = node.value
    if np.abs(v - np.diag(np.diag(v))).max() != 0.0:
        _fail("%s must be diagonal" % what)
    return np.diag(v).copy()


def describe(start):
    """Recognise the LDS graph around `start`; returns the pieces and the priors (no GPU involved)."""
    comp = _component(start)
    stacks = [n for n in comp if isinstance(n, N.hstack)]
    noise = [n for n in comp if isinstance(n, (N.Gamma, N.DiagonalGamma, N.Wishart))]
    if len(stacks) != 2 or len(noise) != 2:
        _fail("expected two hstack matrices and two noise-precision nodes, found %d and %d" % (len(stacks), len(noise)))
    if type(noise[0]) is not type(noise[1]):
        _fail("Q and R must both be DiagonalGamma, both Gamma or both Wishart")
    if any(isinstance(n, N.Addition) for n in comp):
        _fail("Addition nodes")
    # the chain start: a Gaussian with Constant parents that is multiplied by an hstack
    gauss = [n for n in comp if isinstance(n, N.Gaussian)]
    cols = set(id(p) for s in stacks for p in s.parents)
    heads = [g for g in gauss if id(g) not in cols and isinstance(g.mean_parent, N.Constant)]
    if len(heads) != 1 or not isinstance(heads[0].precision_parent, N.Constant):
        _fail("expected exactly one state with Constant parents (X_0)")
    X0 = heads[0]
    Xs, Ys, A, C = [X0], [], None, None
    x = X0
    while True:
        nxt = None
        for m in x.children:
            if not isinstance(m, N.Multiplication) or m.B is not x or not isinstance(m.A, N.hstack) or len(m.children) != 1:
                _fail("a state has a child that is not hstack * state feeding one Gaussian")
            child = m.children[0]
            if not child.children:          # an output: observed, partially observed or not at all (NaN entries, gaussian.py:90-96)
                if C is None:
                    C = m.A
                if m.A is not C:
                    _fail("outputs use different observation matrices")
                Ys.append(child)
            else:
                if A is None:
                    A = m.A
                if m.A is not A or nxt is not None:
                    _fail("states use different transition matrices")
                nxt = child
        if len(Ys) != len(Xs):
            _fail("every state needs exactly one fully observed output")
        if nxt is None:
            break
        Xs.append(nxt)
        x = nxt
    if A is None or C is None or A is C:
        _fail("need distinct transition and observation matrices")
    T, D, K = len(Xs), A.shape[0], C.shape[0]
    if T < 2 or A.shape != (D, D) or C.shape != (K, D) or X0.shape[0] != D:
        _fail("shapes")
    if D > 128 or K > 128:
        _fail("the fused LDS kernels take D, K <= 128")
    big = D > 64 or K > 64          # the workgroup-per-replicate kernels (pyvb_amd/csrc/k_big.hip): the plain graph only
    Q, R = Xs[1].precision_parent, Ys[0].precision_parent
    if big and isinstance(Q, N.Wishart) and (any(not y.observed for y in Ys) or any(c.observed or c.partially_observed for c in A.parents + C.parents)):
        _fail("above 64 dimensions Wishart noise does not combine with known entries of A / C or outputs that hold NaN")
    if Q is R or any(x.precision_parent is not Q for x in Xs[1:]) or any(y.precision_parent is not R for y in Ys):
        _fail("noise precisions are not shared along the chain")
    if any(x.partially_observed or x.observed for x in Xs):
        _fail("observed states")
    for y in Ys:
        if not y.observed:
            cov = y.__dict__["_h_qcov"]
            if np.abs(cov - np.eye(K) * cov[0, 0]).max() != 0.0:
                _fail("the initial covariance of an output with missing entries must be a multiple of the identity")
    As, Cs = A.parents, C.parents

    def known_entries(cols, rows):
        """As[i].observe(...) (examples/LDS_knowns_in_A.py:73-74) -> [rows, D] array, NaN = unknown"""
        obs = np.full((rows, len(cols)), np.nan)
        for i, col in enumerate(cols):
            if col.observed:
                obs[:, i] = col.__dict__["_h_qmu"].reshape(-1)
            elif col.partially_observed:
                obs[:, i] = col.obs_value.reshape(-1)
        return obs

    kind = "diagonal_gamma" if isinstance(Q, N.DiagonalGamma) else ("wishart" if isinstance(Q, N.Wishart) else "gamma")
    pri = {
        "noise": kind,
        "x0_mean": X0.mean_parent.value.reshape(-1).astype(float), "x0_prec": np.asarray(X0.precision_parent.value, dtype=float),
        "A_prior_mean": np.hstack([a.mean_parent.value for a in As]).astype(float),
        "A_prior_prec": np.stack([_diag_constant(a.precision_parent, "a column's prior precision") for a in As]),
        "C_prior_mean": np.hstack([c.mean_parent.value for c in Cs]).astype(float),
        "C_prior_prec": np.stack([_diag_constant(c.precision_parent, "a column's prior precision") for c in Cs]),
    }
    for a in As + Cs:
        if not isinstance(a.mean_parent, N.Constant):
            _fail("matrix columns need Constant mean parents")
    if kind == "diagonal_gamma":
        pri.update(Q_a0=Q.a0s, Q_b0=Q.b0s, R_a0=R.a0s, R_b0=R.b0s)
    elif kind == "wishart":
        pri.update(Q_a0=float(Q.v0), Q_b0=np.array(Q.w0, dtype=float), R_a0=float(R.v0), R_b0=np.array(R.w0, dtype=float))
    else:
        pri.update(Q_a0=float(Q.a0), Q_b0=float(Q.b0), R_a0=float(R.a0), R_b0=float(R.b0))
    for col in As + Cs:
        cov = col.__dict__["_h_qcov"]
        if np.abs(cov - np.diag(np.diag(cov))).max() != 0.0:
            _fail("initial column covariances must be diagonal")
    if any(c.observed or c.partially_observed for c in As + Cs):
        pri["A_obs"], pri["C_obs"] = known_entries(As, D), known_entries(Cs, K)
    return dict(Xs=Xs, Ys=Ys, As=As, Cs=Cs, A=A, C=C, Q=Q, R=R, pri=pri, classes=_state_classes(Xs))


def _state_classes(Xs):
    """(Sigma[3,D,D], q_ln_det[3]) if the states' covariances take the three values a sweep under frozen parameters leaves
    -- X_0, the interior X_t, X_{T-1} -- as they do on a graph that comes from another handle (LDSPlan._sync_host) or from
    complete sweeps on the node-by-node plan; None for the individual random covariances of the constructors
    (gaussian.py:70-72), with which only a sweep can start (pyvb_lds_set_posterior_classes)."""
    T = len(Xs)
    cov = [x.__dict__["_h_qcov"] for x in Xs]
    qld = [x.__dict__.get("_h_q_ln_det") for x in Xs]
    if any(q is None for q in qld):
        return None
    mid = 1 if T > 2 else 0
    for t in range(2, T - 1):
        if qld[t] != qld[mid] or not np.array_equal(cov[t], cov[mid]):
            return None
    return (np.stack([cov[0], cov[mid], cov[T - 1]]).astype(float), np.array([qld[0], qld[mid], qld[T - 1]], dtype=float))


def bind(node):
    """Give the graph `node` belongs to an execution plan: the fused LDS or VB-PCA plan if it is one of those graphs,
    else the generic node-by-node plan (pyvb_amd/generic.py).  All three run on the device."""
    comp = _component(node)
    try:
        if any(isinstance(n, N.Addition) for n in comp):
            return PCAPlan(**describe_pca(node))
        return LDSPlan(**describe(node))
    except NotImplementedError:
        from .generic import GenericPlan
        return GenericPlan(node)


# -------------------------------------------------------------------------------------------------
# VB-PCA with missing data: examples/PCA_missing_data.py:31-42
# -------------------------------------------------------------------------------------------------
def describe_pca(start):
    """Recognise  X_n ~ N(W * Z_n + Mu, Beta)  with W = hstack of Gaussian columns, Mu Gaussian, Beta Gamma,
    Z_n ~ N(0, I); returns the pieces, the priors and the observation mask (no GPU involved)."""
    comp = _component(start)
    stacks = [n for n in comp if isinstance(n, N.hstack)]
    gammas = [n for n in comp if isinstance(n, (N.Gamma, N.DiagonalGamma, N.Wishart))]
    adds = [n for n in comp if isinstance(n, N.Addition)]
    if len(stacks) != 1 or len(gammas) != 1 or not isinstance(gammas[0], N.Gamma) or not adds:
        _fail("expected one hstack matrix, one Gamma precision and Addition nodes (the PCA graph)")
    W, Beta = stacks[0], gammas[0]
    Ws = W.parents
    d, q = W.shape
    if d > 256 or q > 32:
        _fail("the fused PCA kernels are built for d <= 256, q <= 32")
    Mu, Zs, Xs = None, [], []
    # the X_n in construction order = the order in which Beta adopted them as children
    for x in Beta.children:
        add = x.mean_parent
        if not isinstance(x, N.Gaussian) or not isinstance(add, N.Addition) or x.precision_parent is not Beta:
            _fail("a child of the precision node is not a Gaussian on an Addition")
        mult, mu = add.A, add.B
        if not isinstance(mult, N.Multiplication) or mult.A is not W or not isinstance(mult.B, N.Gaussian) or len(add.children) != 1:
            _fail("the mean of an observation is not  W * z + Mu")
        if Mu is None:
            Mu = mu
        if mu is not Mu or not isinstance(Mu, N.Gaussian):
            _fail("observations use different offsets")
        z = mult.B
        if not isinstance(z.mean_parent, N.Constant) or not isinstance(z.precision_parent, N.Constant) \
                or z.mean_parent.value.any() or not np.array_equal(z.precision_parent.value, np.eye(q)) or len(z.children) != 1:
            _fail("latent variables must be N(0, I) with a single use")
        if x.children:
            _fail("observations must be leaves")
        Zs.append(z)
        Xs.append(x)
    if len(W.children) != len(Xs) or len(Mu.children) != len(Xs):
        _fail("W or Mu feed other nodes too")
    for col in Ws + [Mu]:
        if not isinstance(col.mean_parent, N.Constant) or col.observed or col.partially_observed:
            _fail("matrix columns and the offset need Constant parents and no observations")
        cov = col.__dict__["_h_qcov"]
        if np.abs(cov - np.diag(np.diag(cov))).max() != 0.0:
            _fail("initial covariances of the columns must be diagonal")
    # The kernels keep ONE covariance for all Z_n (they share it from their first update on).  The constructors draw an
    # individual one per node (gaussian.py:70-72); everything that reads them before that first update -- the W columns,
    # Beta, the lower bound's trace terms -- is linear in them and sums over n, so their mean serves exactly.
    zc = Zs[0].__dict__["_h_qcov"]
    if any(np.abs(z.__dict__["_h_qcov"] - zc).max() != 0.0 for z in Zs):
        zc = np.mean([z.__dict__["_h_qcov"] for z in Zs], axis=0)
    pri = {
        "W_prior_mean": np.hstack([w.mean_parent.value for w in Ws]).astype(float),
        "W_prior_prec": np.stack([_diag_constant(w.precision_parent, "a column's prior precision") for w in Ws]),
        "Mu_prior_mean": Mu.mean_parent.value.reshape(-1).astype(float),
        "Mu_prior_prec": _diag_constant(Mu.precision_parent, "the offset's prior precision"),
        "beta_a0": float(Beta.a0), "beta_b0": float(Beta.b0),
    }
    N_ = len(Xs)
    obs = np.ones((N_, d), dtype=bool)
    X = np.empty((N_, d))
    X_full, X_var0 = np.empty((N_, d)), np.zeros(N_)
    for n, x in enumerate(Xs):
        X[n] = X_full[n] = x.__dict__["_h_qmu"].reshape(-1)
        if x.partially_observed:
            obs[n] = ~np.isnan(x.obs_value.reshape(-1))
            X[n] = np.where(obs[n], x.obs_value.reshape(-1), X[n])
        elif not x.observed:
            obs[n] = False
        if not x.observed:
            # until its first update the row is what the constructor (or the caller) made it: a mean at ALL entries and a
            # covariance, gaussian.py:70-72 / :90-96; the kernels keep one variance per row
            cov = x.__dict__["_h_qcov"]
            if np.abs(cov - cov[0, 0] * np.eye(d)).max() != 0.0 or not cov[0, 0] > 0.0:
                _fail("the initial covariance of a row with missing entries must be a positive multiple of the identity")
            X_var0[n] = cov[0, 0]
    init = {"obs": obs, "X": X, "X_full": X_full, "X_var0": X_var0, "W_mean": np.hstack([w.__dict__["_h_qmu"] for w in Ws]),
            "W_var": np.stack([np.diag(w.__dict__["_h_qcov"]) for w in Ws]).copy(), "Mu_var": np.diag(Mu.__dict__["_h_qcov"]).copy(),
            "Z": np.hstack([z.__dict__["_h_qmu"] for z in Zs]).T.copy(), "Z_cov": zc.copy(),
            "Mu_mean": Mu.__dict__["_h_qmu"].reshape(-1).copy(),
            "beta_b": float(np.asarray(Beta.__dict__["_h_qb"], dtype=float).reshape(-1)[0])}
    return dict(Ws=Ws, W=W, Mu=Mu, Beta=Beta, Zs=Zs, Xs=Xs, init=init, pri=pri)


class PCAPlan(object):
    def __init__(self, Ws, W, Mu, Beta, Zs, Xs, init, pri):
        from .pca import PCABatch
        self.Ws, self.W, self.Mu, self.Beta, self.Zs, self.Xs = Ws, W, Mu, Beta, Zs, Xs
        self.N, self.d, self.q = len(Xs), W.shape[0], W.shape[1]
        self.obs = init["obs"]
        nmiss = (~self.obs).sum(1)
        self.unpinned = (nmiss > 0) & (nmiss < self.d)       # rows not yet conditioned on their observed entries (first update)
        self.batch = PCABatch.from_problem(init, pri)
        self.index = {}
        for i, w in enumerate(Ws):
            self.index[id(w)] = ("w", i)
        for n, z in enumerate(Zs):
            self.index[id(z)] = ("z", n)
        for n, x in enumerate(Xs):
            self.index[id(x)] = ("x", n)
        self.index[id(Mu)] = ("mu", 0)
        self.index[id(Beta)] = ("beta", 0)
        self.pending, self.cache = [], None
        self.stale = False
        self.z_updated = False          # until then Z_n.qcov reads the node's own initial covariance (host)
        self.n_random_nodes = 2 * self.N + self.q + 2
        for n in _component(W):
            n._plan = self

    def enqueue(self, node):
        self.pending.append(self.index[id(node)])

    def flush(self):
        ops, self.pending = self.pending, []
        if not ops:
            return
        self.cache = None
        b, i = self.batch, 0
        while i < len(ops):
            kind, idx = ops[i]
            j = i
            while j + 1 < len(ops) and ops[j + 1] == (kind, ops[j][1] + 1):
                j += 1
            lo, hi = idx, ops[j][1] + 1
            if kind == "w":
                if (lo, hi) != (0, self.q):     # a single column: not what the fused kernels serve -> node by node
                    return self._demote(ops[i:])
                b.update_W()
            elif kind == "z":
                if (lo, hi) != (0, self.N):
                    return self._demote(ops[i:])
                b.update_Z()
                self.z_updated = True
            elif kind == "x":
                b.update_X(lo, hi)
                self.unpinned[lo:hi] = False
            elif kind == "mu":
                b.update_Mu()
            elif kind == "beta":
                b.update_Beta()
            i = j + 1

    def _pull(self):
        self.flush()
        if self.cache is None:
            self.cache = self.batch.get_state()
        return self.cache

    def read(self, node, name):
        kind, i = self.index[id(node)]
        self.flush()
        if node._plan is not self:
            return node._plan.read(node, name)
        st = self._pull()
        if kind == "w":
            return st["W_mean"][:, [i]].copy() if name == "qmu" else (np.diag(st["W_var"][i]) if name == "qcov" else node.__dict__.get("_h_" + name))
        if kind == "z":
            if name == "qcov" and not self.z_updated:
                return node.__dict__.get("_h_qcov")
            return st["Z"][i].reshape(-1, 1).copy() if name == "qmu" else (st["Z_cov"].copy() if name == "qcov" else node.__dict__.get("_h_" + name))
        if kind == "x":
            if name == "qmu":
                return st["X"][i].reshape(-1, 1).copy()
            if name == "qcov":
                return np.diag(np.where(self.obs[i] & ~self.unpinned[i], 0.0, st["X_rowvar"][i]))
            return node.__dict__.get("_h_" + name)
        if kind == "mu":
            return st["Mu_mean"].reshape(-1, 1).copy() if name == "qmu" else (np.diag(st["Mu_var"]) if name == "qcov" else node.__dict__.get("_h_" + name))
        if name == "qb":
            return float(st["beta_b"])
        raise AttributeError(name)

    def write(self, node, name, value):
        """Nothing is patched in place: the caller releases the plan and the graph is bound anew with the assignment."""
        self.flush()
        if node._plan is not self:
            return node._plan.write(node, name, value)
        return name not in ("qmu", "qcov", "qb", "qw")

    def elbo_parts(self):
        self.flush()
        if self.W._plan is not self:
            raise NotImplementedError("the graph runs node by node now: use Network.learn or the nodes' log_lower_bound()")
        return self.batch.elbo()

    def node_llb(self, node):
        self.flush()
        if node._plan is not self:
            return node._plan.node_llb(node)
        kind, _ = self.index[id(node)]
        if kind == "beta":
            return float(self.elbo_parts()[4])
        # single Gaussian nodes: the fused kernels form class sums (and keep Mu's q_ln_det only from its last update on THIS
        # handle: a graph bound anew has it on the host), so the term comes from the generic tape path on a mirror of the state
        return self.mirror().node_llb(node)

    def _sync_host(self):
        st = self._pull()
        for nd in self.Ws + self.Zs + self.Xs + [self.Mu]:
            for name in ("qmu", "qcov"):
                nd.__dict__["_h_" + name] = self.read(nd, name)
        self.Beta.__dict__["_h_qb"] = self.read(self.Beta, "qb")
        # q_ln_det (gaussian.py:120, quirk Q1): what the updates on this handle left on the device (pyvb_pca_get_qld).  A node
        # that has not been updated on it keeps the value it came with -- the reference sets q_ln_det only in update() too
        qld = self.batch.get_qld()
        for i, nd in enumerate(self.Ws):
            if np.isfinite(qld["W"][i]):
                nd.__dict__["_h_q_ln_det"] = float(qld["W"][i])
        if np.isfinite(qld["Mu"]):
            self.Mu.__dict__["_h_q_ln_det"] = qld["Mu"]
        if self.z_updated and np.isfinite(qld["Z"]):
            for z in self.Zs:
                z.__dict__["_h_q_ln_det"] = qld["Z"]
        for n in np.nonzero(np.isfinite(qld["X"]))[0]:
            self.Xs[n].__dict__["_h_q_ln_det"] = float(qld["X"][n])

    def release(self):
        """See LDSPlan.release."""
        self.flush()
        if self.W._plan is not self:
            return
        self._sync_host()
        for n in _component(self.W):
            if n._plan is self:
                n._plan = None
        self.batch.close()

    def _demote(self, rest):
        """See LDSPlan._demote."""
        from .generic import GenericPlan
        lookup = {v: k for k, v in self.index.items()}
        by_id = {id(n): n for n in self.Ws + self.Zs + self.Xs + [self.Mu, self.Beta]}
        self._sync_host()
        for n in _component(self.W):
            n._plan = None
        self.batch.close()
        gp = GenericPlan(self.W)
        for key in rest:
            node = by_id[lookup[key]]
            if not getattr(node, "observed", False):
                gp.enqueue(node)
        return gp

    def mirror(self):
        """See LDSPlan.mirror."""
        self.flush()
        if getattr(self, "_mirror", None) is None or self._mirror_of is not self.cache or self.cache is None:
            from .generic import GenericPlan
            c = self._pull()
            self._sync_host()
            if getattr(self, "_mirror", None) is not None:
                self._mirror.ex and self._mirror.ex.close()
            self._mirror = GenericPlan(self.W, adopt=False)
            self._mirror_of = c
        return self._mirror
