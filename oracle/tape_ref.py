"""ORACLE (test infrastructure, not product code): numpy restatement of the tape interpreter of
pyvb_amd/csrc/k_tape.hip, record by record.

The generic per-node path emits, for every reference method (gaussian.py:102-183, node.py:95-276,
nodes_todo.py:33-234), a tape of small dense operations on one arena of doubles (pyvb_amd/generic.py); on the GPU one
workgroup interprets it.  This file interprets the same records on a numpy array, so that the host-side emitters can be
checked against the reference's fixtures without a GPU (tests/test_generic_cpu.py) and the device interpreter against
this one (tests/test_generic_gpu.py).  Parity status: pinned through tests/golden/generic_*.npz (reference outputs).
Only tests/ may import this module.
"""
import numpy as np
from scipy.special import digamma, gammaln

(T_NOP, T_COPY2D, T_FILL, T_AXPBY, T_GEMM, T_SCALE, T_TRACE, T_DIAG, T_CHOLINV, T_DOT, T_UNARY, T_GATHER, T_SCATTER,
 T_MUL) = range(14)


class LinAlgStatus(Exception):
    pass


def run(arena, ops):
    """arena: 1-D float64 array, modified in place; ops: int array [nops, 8]."""
    A = arena
    bad = False
    for o in np.asarray(ops, dtype=np.int64).reshape(-1, 8):
        op, d, a, b, m, n, p, fl = [int(v) for v in o]
        if op == T_NOP:
            continue
        if op == T_COPY2D:
            for i in range(m):
                A[d + i * b:d + i * b + n] = A[a + i * p:a + i * p + n].copy()
        elif op == T_FILL:
            for i in range(m):
                A[d + i * b:d + i * b + n] = 0.0
                if (fl & 1) and i < n:
                    A[d + i * b + i] = 1.0
        elif op == T_AXPBY:
            x = A[a:a + m * n]
            A[d:d + m * n] = A[p] * x if b < 0 else A[p] * x + A[fl] * A[b:b + m * n]
        elif op == T_GEMM:
            k = p
            X = A[a:a + m * k].reshape((k, m)).T if (fl & 1) else A[a:a + m * k].reshape((m, k))
            Y = A[b:b + k * n].reshape((n, k)).T if (fl & 2) else A[b:b + k * n].reshape((k, n))
            Z = X @ Y
            if fl & 8:
                Z = -Z
            cur = A[d:d + m * n].reshape((m, n))
            A[d:d + m * n] = (cur + Z if (fl & 4) else Z).reshape(-1)
        elif op == T_SCALE:
            s = A[b]
            A[d:d + m * n] = A[a:a + m * n] / s if (fl & 1) else A[a:a + m * n] * s
        elif op == T_TRACE:
            v = np.trace(A[a:a + m * m].reshape((m, m)))
            A[d] = A[d] + v if (fl & 4) else v
        elif op == T_DIAG:
            if fl & 1:
                A[d:d + m * m] = np.diag(A[a:a + m]).reshape(-1)
            else:
                A[d:d + m] = np.diag(A[a:a + m * m].reshape((m, m)))
        elif op == T_CHOLINV:
            M = A[a:a + m * m].reshape((m, m))
            Ms = np.tril(M) + np.tril(M, -1).T          # the kernel reads the lower triangle
            try:
                L = np.linalg.cholesky(Ms)
                s = np.sum(np.log(np.diag(L)))
                X = np.linalg.inv(L)
                A[d:d + m * m] = (X.T @ X).reshape(-1)
                A[b] = 0.5 / s
                A[b + 1] = s
            except np.linalg.LinAlgError:
                bad = True
                A[d:d + m * m] = np.nan
        elif op == T_DOT:
            v = float(np.dot(A[a:a + m * n], A[b:b + m * n]))
            A[d] = A[d] + v if (fl & 4) else v
        elif op == T_UNARY:
            x = A[a:a + m * n]
            with np.errstate(all="ignore"):
                A[d:d + m * n] = [np.log, digamma, gammaln, lambda v: 1.0 / v, lambda v: -v, np.exp][fl](x)
        elif op == T_GATHER:
            r = A[b:b + m].astype(int)
            c = A[fl:fl + n].astype(int)
            A[d:d + m * n] = np.array([[A[a + ri * p + cj] for cj in c] for ri in r]).reshape(-1)
        elif op == T_SCATTER:
            acc = bool(fl & 0x40000000)
            co = fl & ~0x40000000
            r = A[b:b + m].astype(int)
            c = A[co:co + n].astype(int)
            for i, ri in enumerate(r):
                for j, cj in enumerate(c):
                    A[d + ri * p + cj] = (A[d + ri * p + cj] if acc else 0.0) + A[a + i * n + j]
        elif op == T_MUL:
            A[d:d + m * n] = A[a:a + m * n] * A[b:b + m * n]
        else:
            raise ValueError("unknown opcode %d" % op)
    if bad:
        raise LinAlgStatus()


class NumpyExecutor(object):
    """Drop-in for pyvb_amd.generic.DeviceExecutor in CPU tests: same interface, the arena is a numpy array."""

    def __init__(self, arena_doubles):
        self.size = int(arena_doubles)
        self.arena = np.zeros(self.size)
        self.tapes = []
        self.programs = []
        self.bad = False

    def write(self, off, arr):
        a = np.asarray(arr, dtype=float).reshape(-1)
        self.arena[off:off + a.size] = a

    def read(self, off, n):
        self.sync()
        return self.arena[off:off + n].copy()

    def tape(self, ops, program=None):
        self.tapes.append(np.array(ops, dtype=np.int64).reshape(-1, 8))
        self.programs.append(program)
        return len(self.tapes) - 1

    def drop(self, tid):
        self.tapes[tid] = None
        self.programs[tid] = None

    def run(self, tid):
        """With a program (pyvb_graph_tape_set_program: record ranges the device runs side by side) the blocks of a launch are
        interpreted LAST FIRST: if the independence the program claims did not hold, the results would depend on that order and
        the comparisons with the reference's fixtures would fail."""
        try:
            ops, prog = self.tapes[tid], self.programs[tid]
            if prog is None:
                run(self.arena, ops)
            else:
                blocks, launches = [np.asarray(a).reshape(-1, 2) for a in prog]
                for first, count in launches:
                    for a, n in reversed(blocks[first:first + count].tolist()):
                        run(self.arena, ops[a:a + n])
        except LinAlgStatus:
            self.bad = True

    def sync(self):
        if self.bad:
            self.bad = False
            raise np.linalg.LinAlgError("a posterior precision was not positive definite")

    def close(self):
        pass
