"""Network: the reference's container and learning loop (src/pyvb/network.py) over pyvb_amd.nodes."""
import numpy as np

from . import nodes as N

__all__ = ["Network"]


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

    def learn(self, niters, tol=1e-3, verbose=True):
        self.find_iterable()
        if verbose:
            print('Found' + str(len(self.iterable_nodes)) + ' iterable nodes\n')
        if not self.iterable_nodes:
            return
        plan = N._plan_of(self.iterable_nodes[0])
        if getattr(plan, "generic", False):
            return self._learn_generic(niters, tol, verbose)
        missing = [n for n in self.iterable_nodes if n._plan is not plan]
        if missing:
            raise NotImplementedError("the network spans nodes outside one recognised LDS graph; no HIP plan")
        needed = plan.n_random_nodes
        if len(set(id(n) for n in self.iterable_nodes)) != needed:
            raise NotImplementedError("Network.learn needs every random-variable node of the graph (the lower bound is a "
                                      "sum over all of them): call fetch_network() first")
        old_llb = -np.inf
        for i in range(niters):
            for n in self.iterable_nodes:
                n.update()
            plan.flush()
            if self.iterable_nodes[0]._plan is not plan:        # an update order the fused kernels do not serve: the graph
                self.llb = float(sum(n.log_lower_bound() for n in self.iterable_nodes))     # runs node by node from here on
                if verbose:
                    print(niters - i, self.llb)
                if self.llb - old_llb < tol:
                    if verbose:
                        print("Convergence!")
                    return
                return self._learn_generic(niters - i - 1, tol, verbose, old_llb=self.llb)
            self.llb = float(np.sum(plan.elbo_parts()))        # network.py:49
            if verbose:
                print(niters - i, self.llb)
            if self.llb - old_llb < tol:                        # also fires when the bound decreases
                if verbose:
                    print("Convergence!")
                break
            old_llb = self.llb

    def _learn_generic(self, niters, tol, verbose, old_llb=-np.inf):
        """network.py:40-56 for a graph that runs node by node: per iteration one launch for all update() calls in
        list order and one for the sum of the log_lower_bound() terms; nodes of several unconnected graphs are grouped
        by plan."""
        plans = []
        for n in self.iterable_nodes:
            p = N._plan_of(n)
            if not getattr(p, "generic", False):
                raise NotImplementedError("a Network that mixes a fused (LDS / PCA) graph with other graphs")
            if p not in plans:
                plans.append(p)
        groups = [(p, [n for n in self.iterable_nodes if n._plan is p]) for p in plans]
        for i in range(niters):
            for p, group in groups:
                p.update_all(group)
            self.llb = float(sum(p.llb_sum(group).sum() for p, group in groups))        # network.py:49
            if verbose:
                print(niters - i, self.llb)
            if self.llb - old_llb < tol:
                if verbose:
                    print("Convergence!")
                break
            old_llb = self.llb

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
