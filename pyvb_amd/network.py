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

    def _groups(self):
        """The iterable nodes grouped by the device plan their graph is bound to, plans in order of first appearance, nodes
        in list order inside a group.  Unconnected graphs do not exchange messages, so running the groups one after the
        other gives what the reference's single pass over the list gives (network.py:46-49)."""
        plans, groups = [], {}
        for n in self.iterable_nodes:
            p = N._plan_of(n)
            if id(p) not in groups:
                plans.append(p)
                groups[id(p)] = []
            groups[id(p)].append(n)
        return [(p, groups[id(p)]) for p in plans]

    def learn(self, niters, tol=1e-3, verbose=True):
        """network.py:40-56.  The node list may span any number of unconnected graphs, each on its own plan: a fused LDS
        or VB-PCA plan takes its group's update() calls as queued requests (whole sweeps become single launches) and gives
        the lower bound as the sum of its class parts; a graph that runs node by node gets one launch for all its updates
        and one for the sum of its log_lower_bound() terms.  The grouping is redone around every step because an update
        order the fused kernels do not serve moves a graph to the node-by-node plan (and sweeps bring it back)."""
        self.find_iterable()
        if verbose:
            print('Found' + str(len(self.iterable_nodes)) + ' iterable nodes\n')
        if not self.iterable_nodes:
            return
        old_llb = -np.inf
        for i in range(niters):
            for p, group in self._groups():
                if getattr(p, "generic", False):
                    p.update_all(group)
                else:
                    for n in group:
                        n.update()
                    p.flush()
            llb = 0.0
            for p, group in self._groups():
                if getattr(p, "generic", False):
                    llb += float(p.llb_sum(group).sum())
                elif len(group) == len(set(id(n) for n in group)) == p.n_random_nodes:
                    llb += float(np.sum(p.elbo_parts()))        # every random node of the graph is listed, once: the class sums
                else:
                    llb += float(sum(n.log_lower_bound() for n in group))     # a part of a fused graph: its terms one by one
            self.llb = llb                                      # network.py:49
            if verbose:
                print(niters - i, self.llb)
            if self.llb - old_llb < tol:                        # also fires when the bound decreases
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
