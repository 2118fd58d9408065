"""Network: the reference's container and learning loop (src/pyvb/network.py) over pyvb_amd.nodes."""
import numpy as np

from . import nodes as N

__all__ = ["Network"]


def _bound(p, nodes):
    """The nodes still sit on the plan they were grouped under (a plan covers one connected graph, so one node tells)."""
    return nodes[0]._plan is p and not (getattr(p, "stale", False) or getattr(p, "released", False) or getattr(p, "closed", False))


class _Schedule(object):
    """One pass of Network.learn over a node list whose graphs are bound: groups = [[plan, nodes, whole], ...] in list
    order (Network._groups).  Graphs on the fused LDS kernels that share a device handle (_recognise.LDSGroup) are served
    per HANDLE once their first pass has shown what their update() calls spell for it: the cost of an iteration over M
    graphs of one structure is then that of the handle's launches, not of M walks over node lists."""

    def __init__(self, groups):
        from ._recognise import LDSPlan
        self.groups = groups
        self.lds = [k for k, (p, _, _) in enumerate(groups) if isinstance(p, LDSPlan)]
        self.other = [k for k, (p, _, _) in enumerate(groups) if not isinstance(p, LDSPlan)]
        self.scripts = {}               # position in groups -> the batch operations its nodes' update() calls spell
        self.handles = None             # after the first pass: [handle, positions, rows, script or None, epoch, whole]

    def valid(self):
        g = self.groups
        if not all(_bound(g[k][0], g[k][1]) for k in self.other):
            return False
        if self.handles is None:
            return all(_bound(g[k][0], g[k][1]) for k in self.lds)
        return all(h[0].epoch == h[4] for h in self.handles)

    def settle(self):
        """Whatever the plans have queued is carried out."""
        for p, _, _ in self.groups:
            p.flush()

    def _summarise(self):
        """The LDS graphs per handle, once they are all on the device."""
        by = {}
        for k in self.lds:
            p = self.groups[k][0]
            if p.group is None:
                return
            by.setdefault(id(p.group), []).append(k)
        self.handles = []
        for ks in by.values():
            grp = self.groups[ks[0]][0].group
            sc = self.scripts.get(ks[0])
            same = sc is not None and all(self.scripts.get(k) == sc for k in ks) and len(ks) == len(grp.live())
            self.handles.append([grp, ks, np.array([self.groups[k][0].r for k in ks]), sc if same else None, grp.epoch,
                                 all(self.groups[k][2] for k in ks), [self.groups[k][0] for k in ks], np.array(ks)])

    def update(self):
        g = self.groups
        for k in self.other:
            p, nodes, _ = g[k]
            if getattr(p, "generic", False):
                p.update_all(nodes)
            else:
                for n in nodes:
                    n.update()
                p.flush()
        if self.handles is None:        # first pass: the calls themselves, queued on every graph before anything runs
            for k in self.lds:
                p, nodes, _ = g[k]
                first = len(p.pending)
                for n in nodes:
                    n.update()
                self.scripts[k] = p._spell(first)
            for k in self.lds:
                g[k][0].flush()
            if all(_bound(g[k][0], g[k][1]) for k in self.lds):
                self._summarise()
            return
        for grp, ks, rows, sc, epoch, whole, members, _ in self.handles:
            if sc is not None and not any(m.pending for m in members) and grp.run_script(sc):
                continue
            for k, m in zip(ks, members):
                if self.scripts.get(k) is not None:
                    m.pending.extend(self.scripts[k])
                else:
                    for n in g[k][1]:
                        n.update()
            for m in members:
                m.flush()

    def llb(self):
        """sum of log_lower_bound() over the list (network.py:49), accumulated plan by plan in list order."""
        g = self.groups
        vals = np.zeros(len(g))
        for k in self.other:
            p, nodes, whole = g[k]
            if getattr(p, "generic", False):
                vals[k] = float(p.llb_sum(nodes).sum())
            elif whole:
                vals[k] = float(np.sum(p.elbo_parts()))         # every random node of the graph is listed, once: the class sums
            else:
                vals[k] = float(sum(n.log_lower_bound() for n in nodes))    # a part of a fused graph: its terms one by one
        rest = self.lds
        if self.handles is not None:
            rest = []
            for grp, ks, rows, sc, epoch, whole, members, where in self.handles:
                if whole and grp.epoch == epoch:
                    vals[where] = grp.elbo()[rows].sum(1)       # one launch and one copy for all graphs of the handle
                else:
                    rest += ks
        for k in rest:
            p, nodes, whole = g[k]
            vals[k] = float(np.sum(p.elbo_parts())) if whole else float(sum(n.log_lower_bound() for n in nodes))
        llb = 0.0
        for v in vals.tolist():
            llb += v
        return llb


class Network(object):
    """A list of nodes with `learn()`: update every iterable node in list order, evaluate the lower
    bound, stop when it improves by less than `tol` (network.py:40-56).

    The updates are queued on the graph's device plan and flushed once per iteration, so a node
    list that contains the states in chain order costs one sweep launch per iteration; the lower
    bound is the sum of the per-class parts computed on the device.
    """

    def __init__(self, nodes=[]):
        self.nodes = []
        [self.addnode(n) for n in nodes]

    def addnode(self, n):
        if type(n) is list:
            self.nodes.extend(n)
        else:
            self.nodes.append(n)

    def find_iterable(self):            # network.py:35-37
        self.iterable_nodes = [e for e in self.nodes if isinstance(e, (N.Gaussian, N.Gamma, N.DiagonalGamma, N.Wishart))]

    def _groups(self):
        """The iterable nodes grouped by the device plan their graph is bound to, plans in order of first appearance, nodes
        in list order inside a group.  Unconnected graphs do not exchange messages, so running the groups one after the
        other gives what the reference's single pass over the list gives (network.py:46-49).  Entries: [plan, nodes,
        whether the nodes are all random nodes of the plan's graph, each once]."""
        plans, groups = [], {}
        for n in self.iterable_nodes:
            p = N._plan_of(n)
            if id(p) not in groups:
                plans.append(p)
                groups[id(p)] = []
            groups[id(p)].append(n)
        return [[p, groups[id(p)], len(groups[id(p)]) == len(set(id(n) for n in groups[id(p)])) == p.n_random_nodes] for p in plans]

    def learn(self, niters, tol=1e-3, verbose=True):
        """network.py:40-56.  The node list may span any number of unconnected graphs, each on its own plan: a fused LDS
        or VB-PCA plan takes its group's update() calls as queued requests (whole sweeps become single launches) and gives
        the lower bound as the sum of its class parts; a graph that runs node by node gets one launch for all its updates
        and one for the sum of its log_lower_bound() terms.  LDS graphs of the same structure share one device handle, a
        replicate each (_recognise.LDSGroup): their requests are queued first and carried out together, one launch per
        operation for all of them, and from the second iteration on what the first one's requests spelt for the handle is
        replayed without walking the node list again (_Schedule).  The grouping is redone whenever an update order the
        fused kernels do not serve has moved a graph to the node-by-node plan (or sweeps have brought it back)."""
        # a second call over the same list starts where the first one stopped: the list is not walked again
        kept = getattr(self, "_kept", None)
        sched = kept[1] if kept is not None and kept[0] == self.nodes and kept[1].valid() else None
        if sched is None:
            self.find_iterable()
        if verbose:
            print('Found' + str(len(self.iterable_nodes)) + ' iterable nodes\n')
        if not self.iterable_nodes:
            return
        old_llb = -np.inf
        try:
            for i in range(niters):
                if sched is None or not sched.valid():
                    sched = _Schedule(self._groups())
                sched.update()                                  # network.py:46-48
                for _ in range(8):
                    # carrying out the requests can move a graph to another plan (a handle of its own, the node-by-node plan,
                    # back to the fused one), with requests still queued: group again and settle those too
                    if sched.valid():
                        break
                    sched = _Schedule(self._groups())
                    sched.settle()
                else:
                    raise RuntimeError("the graphs kept changing plans")
                self.llb = sched.llb()                          # network.py:49
                if verbose:
                    print(niters - i, self.llb)
                if self.llb - old_llb < tol:                    # also fires when the bound decreases
                    if verbose:
                        print("Convergence!")
                    break
                old_llb = self.llb
        finally:
            self._kept = (list(self.nodes), sched) if sched is not None else None

    def fetch_network(self, verbose=True):
        """Add every node connected to the ones already listed, in the order the reference's
        crawl (network.py:58-96) finds them: repeated passes over the growing list, each node
        contributing its unseen children, then its unseen parents."""
        n_start = len(self.nodes)
        seen = set(id(n) for n in self.nodes)

        def add(cands):
            new = []
            for e in cands:
                if id(e) not in seen:
                    seen.add(id(e))
                    new.append(e)
            self.nodes.extend(new)
            return len(new)

        new_nodes = True
        while new_nodes:
            new_nodes = 0
            for n in self.nodes:            # the list grows while it is being walked, as in the reference
                if isinstance(n, N.Gaussian):
                    new_nodes += add(n.children)
                    new_nodes += add([n.mean_parent, n.precision_parent])
                elif isinstance(n, (N.Addition, N.Multiplication)):
                    new_nodes += add(n.children)
                    new_nodes += add([n.A, n.B])
                elif isinstance(n, N.hstack):
                    new_nodes += add(n.children)
                    new_nodes += add(n.parents)
                if isinstance(n, (N.Gamma, N.DiagonalGamma, N.Wishart)):
                    new_nodes += add(n.children)
        if verbose:
            print("Found " + str(len(self.nodes) - n_start) + " new nodes.")
