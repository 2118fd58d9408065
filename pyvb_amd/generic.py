"""Generic per-node execution: graphs without a fused plan (pyvb_amd/_recognise.py knows the LDS and the VB-PCA
graph) run node by node on the device, the way the reference runs them on the CPU.

Every posterior, constant, message and temporary of the graph lives in one device arena of doubles
(include/pyvb_hip.h: pyvb_graph_*).  The methods of the reference's node classes are restated here as EMITTERS: where
the reference computes with numpy, an emitter appends records to a tape -- small dense operations on arena offsets --
and returns references to the results.  A tape is uploaded once and replayed: `node.update()` is one launch of the tape
interpreter (pyvb_amd/csrc/k_tape.hip), `Network.learn` one launch per iteration.  Nothing is computed on the host: this
module only decides WHICH operations run (the graph walk the reference does at every call, done once per node here).

Reference methods restated (paths relative to /root/reference/src/pyvb/nodes/):
    Gaussian.update                      gaussian.py:102-134     -> GenericPlan._emit_update_gaussian
    Gaussian.log_lower_bound             gaussian.py:136-151     -> _emit_llb_gaussian
    Gaussian.pass_up_m1_m2               gaussian.py:179-183     -> _pass_up (Gaussian branch)
    Addition.pass_up_m1_m2 / pass_down_* node.py:95-129          -> _pass_up / _ex / _exxt (Addition branches)
    Multiplication.pass_up_m1_m2         node.py:182-232         -> _pass_up (Multiplication branch)
    Multiplication.pass_down_Ex / ExxT   node.py:235-276         -> _ex / _exxt
    hstack.pass_down_* / pass_up_m1_m2   nodes_todo.py:33-62     -> _ex / _exxt / _pass_up (hstack branch)
    Gamma.update / lndet / llb           nodes_todo.py:130-157   -> _emit_update_noise / _lndet / _emit_llb_noise
    DiagonalGamma.*                      nodes_todo.py:187-204
    Wishart.update / pass_down_Ex        nodes_todo.py:228-234   (deviations: pyvb_amd/csrc/k_wishart.hip header)
    Constant.*                           node.py:304-311
Deviations from the reference, all where it is broken: a Multiplication whose left operand is a
Constant matrix or a row vector returns a proper (m1, m2) pair (Q3; node.py:209,212 return a bare matrix, which
Gaussian.update then mis-indexes) and its pass_down_ExxT uses A <BB^T> A^T (node.py:258 has a typo and no transpose, Q6).
"""
import bisect

import numpy as np

from . import nodes as N

(T_NOP, T_COPY2D, T_FILL, T_AXPBY, T_GEMM, T_SCALE, T_TRACE, T_DIAG, T_CHOLINV, T_DOT, T_UNARY, T_GATHER, T_SCATTER,
 T_MUL) = range(14)
U_LOG, U_DIGAMMA, U_LGAMMA, U_RECIP, U_NEG, U_EXP = range(6)
LN2PI = float(np.log(2.0 * np.pi))

CONST_CAP = 8192          # doubles reserved for the constant pool at the start of the arena


# which fields of a record hold arena offsets, per opcode (pyvb_amd/csrc/k_tape.hip: tape_extents); used to move a node's
# temporaries when its records are issued beside another node's (GenericPlan._relocate)
_OFFSET_FIELDS = {T_COPY2D: (1, 2), T_FILL: (1,), T_AXPBY: (1, 2, 3, 6, 7), T_GEMM: (1, 2, 3), T_SCALE: (1, 2, 3), T_TRACE: (1, 2),
                  T_DOT: (1, 2, 3), T_DIAG: (1, 2), T_CHOLINV: (1, 2, 3, 6), T_UNARY: (1, 2), T_GATHER: (1, 2, 3, 7),
                  T_SCATTER: (1, 2, 3, 7), T_MUL: (1, 2, 3)}
T_ACC_BIT = 0x40000000


class Ref(object):
    """An m x n row-major block of the arena."""
    __slots__ = ("off", "m", "n")

    def __init__(self, off, m, n):
        self.off, self.m, self.n = int(off), int(m), int(n)

    @property
    def size(self):
        return self.m * self.n

    def elem(self, i, j=0):
        return Ref(self.off + i * self.n + j, 1, 1)


class Tape(object):
    """Records of one tape plus a bump allocator for its temporaries."""

    def __init__(self, plan):
        self.plan = plan
        self.ops = []
        self.top = plan.temp_base
        self.marks = []                 # record indices where a block begins: a node of a many-node tape, one child's message
        self.barriers = set()           # marks before which everything emitted so far must have finished (temporaries are read)

    def tmp(self, m, n=1):
        r = Ref(self.top, m, n)
        self.top += max(1, m * n)
        self.plan.temp_high = max(self.plan.temp_high, self.top)
        return r

    def mark(self, barrier=False):
        self.marks.append(len(self.ops))
        if barrier:
            self.barriers.add(len(self.ops))

    def emit(self, op, dst, a=0, b=0, m=0, n=0, p=0, flags=0):
        self.ops.append((op, int(dst), int(a), int(b), int(m), int(n), int(p), int(flags)))

    # -- elementary operations; all return a Ref ------------------------------------------------
    def const(self, v):
        return self.plan.const(v)

    def copy(self, a, dst=None):
        dst = dst or self.tmp(a.m, a.n)
        self.emit(T_COPY2D, dst.off, a.off, dst.n, a.m, a.n, a.n)
        return dst

    def eye(self, m):
        r = self.tmp(m, m)
        self.emit(T_FILL, r.off, 0, m, m, m, 0, 1)
        return r

    def zeros(self, m, n=1):
        r = self.tmp(m, n)
        self.emit(T_FILL, r.off, 0, n, m, n, 0, 0)
        return r

    def axpby(self, alpha, a, beta=None, b=None, dst=None):
        """alpha a + beta b; alpha, beta: python floats or 1 x 1 Refs."""
        al = alpha if isinstance(alpha, Ref) else self.const(alpha)
        dst = dst or self.tmp(a.m, a.n)
        if b is None:
            self.emit(T_AXPBY, dst.off, a.off, -1, a.m, a.n, al.off, 0)
        else:
            assert (a.m, a.n) == (b.m, b.n), "shape mismatch in axpby: %r vs %r" % ((a.m, a.n), (b.m, b.n))
            be = beta if isinstance(beta, Ref) else self.const(beta)
            self.emit(T_AXPBY, dst.off, a.off, b.off, a.m, a.n, al.off, be.off)
        return dst

    def add(self, a, b):
        return self.axpby(1.0, a, 1.0, b)

    def sub(self, a, b):
        return self.axpby(1.0, a, -1.0, b)

    def gemm(self, a, b, ta=False, tb=False, dst=None, acc=False, neg=False):
        m, k = (a.n, a.m) if ta else (a.m, a.n)
        k2, n = (b.n, b.m) if tb else (b.m, b.n)
        assert k == k2, "inner dimensions differ: %d vs %d" % (k, k2)
        dst = dst or self.tmp(m, n)
        self.emit(T_GEMM, dst.off, a.off, b.off, m, n, k, (1 if ta else 0) | (2 if tb else 0) | (4 if acc else 0) | (8 if neg else 0))
        return dst

    def transpose(self, a):
        if a.m == 1 or a.n == 1:
            return Ref(a.off, a.n, a.m)
        return self.gemm(a, self.eye(a.m), ta=True)

    def scale(self, a, s, divide=False, dst=None):
        """a * s (or a / s) with s a 1 x 1 Ref (any arena element) or a python float."""
        s = s if isinstance(s, Ref) else self.const(s)
        dst = dst or self.tmp(a.m, a.n)
        self.emit(T_SCALE, dst.off, a.off, s.off, a.m, a.n, 0, 1 if divide else 0)
        return dst

    def trace(self, a):
        assert a.m == a.n
        r = self.tmp(1, 1)
        self.emit(T_TRACE, r.off, a.off, 0, a.m, a.m, 0, 0)
        return r

    def dot(self, a, b):
        assert a.size == b.size
        r = self.tmp(1, 1)
        self.emit(T_DOT, r.off, a.off, b.off, a.m, a.n, 0, 0)
        return r

    def total(self, a):
        return self.dot(a, self.plan.ones(a.size))

    def diag_of(self, a):
        r = self.tmp(a.m, 1)
        self.emit(T_DIAG, r.off, a.off, 0, a.m, a.m, 0, 0)
        return r

    def diag_matrix(self, v):
        r = self.tmp(v.size, v.size)
        self.emit(T_DIAG, r.off, v.off, 0, v.size, v.size, 0, 1)
        return r

    def cholinv(self, a, dst=None, out2=None):
        """(inverse, [q_ln_det (quirk Q1), sum log diag chol]) of a symmetric positive definite block."""
        assert a.m == a.n
        dst = dst or self.tmp(a.m, a.m)
        out2 = out2 or self.tmp(2, 1)
        scratch = self.tmp(2 * a.m * a.m, 1)
        self.emit(T_CHOLINV, dst.off, a.off, out2.off, a.m, a.m, scratch.off, 0)
        return dst, out2

    def unary(self, a, f):
        r = self.tmp(a.m, a.n)
        self.emit(T_UNARY, r.off, a.off, 0, a.m, a.n, 0, f)
        return r

    def mul(self, a, b):
        assert a.size == b.size
        r = self.tmp(a.m, a.n)
        self.emit(T_MUL, r.off, a.off, b.off, a.m, a.n, 0, 0)
        return r

    def gather(self, a, rows, cols):
        """a.take(rows, 0).take(cols, 1); rows / cols: Refs to index vectors stored as doubles."""
        r = self.tmp(rows.size, cols.size)
        self.emit(T_GATHER, r.off, a.off, rows.off, rows.size, cols.size, a.n, cols.off)
        return r

    def lin(self, c0, terms):
        """c0 + sum_i c_i * ref_i for scalars (1 x 1 Refs); c_i python floats."""
        acc = self.copy(self.const(c0))
        for c, r in terms:
            self.axpby(1.0, acc, c, r, dst=acc)
        return acc

    def array(self):
        return np.array(self.ops, dtype=np.int32).reshape(-1, 8)


class DeviceExecutor(object):
    """Arena + tape runner behind GenericPlan: binds include/pyvb_hip.h's pyvb_graph_*."""

    def __init__(self, arena_doubles, device=0):
        from . import _capi as C
        self.C = C
        h = C.ctypes.c_void_p()
        C.check(C.lib.pyvb_graph_create(C.ctypes.byref(h), int(device), int(arena_doubles)))
        self._h = h
        self.size = int(arena_doubles)

    def write(self, off, arr):
        a = np.ascontiguousarray(arr, dtype=np.float64).reshape(-1)
        if a.size:
            self.C.check(self.C.lib.pyvb_graph_write(self._h, int(off), self.C.dptr(a), a.size))

    def read(self, off, n):
        out = np.empty(int(n))
        if n:
            self.C.check(self.C.lib.pyvb_graph_read(self._h, int(off), self.C.dptr(out), int(n)))
        return out

    def tape(self, ops, program=None):
        """program: (blocks [nb, 2], launches [nl, 2]) -- record ranges that may run side by side (pyvb_graph_tape_set_program)."""
        ops = np.ascontiguousarray(ops, dtype=np.int32)
        tid = self.C.ctypes.c_int()
        self.C.check(self.C.lib.pyvb_graph_tape_create(self._h, ops.ctypes.data_as(self.C._ip), int(ops.shape[0]), self.C.ctypes.byref(tid)))
        if program is not None:
            blocks, launches = [np.ascontiguousarray(a, dtype=np.int32).reshape(-1, 2) for a in program]
            self.C.check(self.C.lib.pyvb_graph_tape_set_program(self._h, tid.value, blocks.ctypes.data_as(self.C._ip), int(blocks.shape[0]),
                                                                launches.ctypes.data_as(self.C._ip), int(launches.shape[0])))
        return tid.value

    def run(self, tid):
        self.C.check(self.C.lib.pyvb_graph_tape_run(self._h, int(tid)))

    def drop(self, tid):
        self.C.check(self.C.lib.pyvb_graph_tape_destroy(self._h, int(tid)))

    def sync(self):
        self.C.check(self.C.lib.pyvb_graph_sync(self._h))

    def close(self):
        if getattr(self, "_h", None):
            self.C.lib.pyvb_graph_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def _component(start):
    from ._recognise import _component as comp
    return comp(start)


class GenericPlan(object):
    generic = True

    def __init__(self, start, executor_factory=None, adopt=True):
        self.nodes = _component(start)
        self.stale = False
        self.temp_high = 0
        self._consts, self._const_vals, self._const_dirty = {}, [], True
        self._ones = None
        # ---- state layout
        self.slot = {}
        off = CONST_CAP

        def alloc(n):
            nonlocal off
            r = off
            off += max(1, int(n))
            return r

        init = []       # (offset, array) to upload
        for nd in self.nodes:
            s = {}
            if isinstance(nd, N.Gaussian):
                d = nd.__dict__["_h_qmu"].shape[0]
                s["qmu"] = Ref(alloc(d), d, 1)
                s["qcov"] = Ref(alloc(d * d), d, d)
                s["qld"] = Ref(alloc(2), 2, 1)
                init.append((s["qmu"].off, nd.__dict__["_h_qmu"]))
                init.append((s["qcov"].off, nd.__dict__["_h_qcov"]))
                init.append((s["qld"].off, np.array([nd.__dict__.get("_h_q_ln_det") if nd.__dict__.get("_h_q_ln_det") is not None else np.nan, np.nan])))
                if nd.partially_observed:
                    oi, mi = np.asarray(nd.obs_index, dtype=float), np.asarray(nd.missing_index, dtype=float)
                    s["obs_index"] = Ref(alloc(oi.size), oi.size, 1)
                    s["missing_index"] = Ref(alloc(mi.size), mi.size, 1)
                    s["all_index"] = Ref(alloc(d), d, 1)
                    s["zero_index"] = Ref(alloc(1), 1, 1)
                    s["obs_value"] = Ref(alloc(d), d, 1)
                    init += [(s["obs_index"].off, oi), (s["missing_index"].off, mi), (s["all_index"].off, np.arange(d, dtype=float)),
                             (s["zero_index"].off, np.zeros(1)), (s["obs_value"].off, np.nan_to_num(nd.obs_value))]
            elif isinstance(nd, N.Constant):
                v = np.asarray(nd.value, dtype=float)
                s["value"] = Ref(alloc(v.size), v.shape[0], v.shape[1])
                s["xxT"] = Ref(alloc(v.shape[0] ** 2), v.shape[0], v.shape[0])
                s["xTx"] = Ref(alloc(v.shape[1] ** 2), v.shape[1], v.shape[1])
                s["lndet"] = Ref(alloc(1), 1, 1)
                init += [(s["value"].off, v), (s["xxT"].off, nd.value_xxT), (s["xTx"].off, nd.value_xTx),
                         (s["lndet"].off, np.array([getattr(nd, "lndet", np.nan)], dtype=float))]
            elif isinstance(nd, N.Gamma):
                s["qa"], s["qb"] = Ref(alloc(1), 1, 1), Ref(alloc(1), 1, 1)
                init += [(s["qa"].off, np.array([nd.qa], dtype=float)), (s["qb"].off, np.array([nd.__dict__["_h_qb"]], dtype=float).reshape(-1)[:1])]
            elif isinstance(nd, N.DiagonalGamma):
                dim = nd.shape[0]
                s["qa"], s["qb"] = Ref(alloc(dim), dim, 1), Ref(alloc(dim), dim, 1)
                s["a0"], s["b0"] = Ref(alloc(dim), dim, 1), Ref(alloc(dim), dim, 1)
                init += [(s["qa"].off, np.asarray(nd.qa, dtype=float)), (s["qb"].off, np.broadcast_to(np.asarray(nd.__dict__["_h_qb"], dtype=float), (dim,))),
                         (s["a0"].off, nd.a0s), (s["b0"].off, nd.b0s)]
            elif isinstance(nd, N.Wishart):
                dim = nd.shape[0]
                s["qv"], s["qw"], s["w0"] = Ref(alloc(1), 1, 1), Ref(alloc(dim * dim), dim, dim), Ref(alloc(dim * dim), dim, dim)
                init += [(s["qv"].off, np.array([nd.qv], dtype=float)), (s["qw"].off, nd.__dict__["_h_qw"]), (s["w0"].off, np.asarray(nd.w0, dtype=float))]
            self.slot[id(nd)] = s
        self.temp_base = off
        self.temp_high = off
        self._init = init
        self._tapes = {}                  # key -> (tape id or None while not uploaded, records, result refs)
        self._programs = {}               # key -> (blocks, launches) for tapes whose nodes can run side by side
        starts = sorted((ref.off, id_) for id_, sl in self.slot.items() for ref in sl.values())
        self._slot_start, self._slot_owner = [a for a, _ in starts], [b for _, b in starts]
        # what the tapes run on: the device.  (executor_factory exists for tests/test_generic_cpu.py, which checks the emitters
        # without a GPU by handing in the numpy restatement of the interpreter; nothing in the package passes it.)
        self._executor_factory = executor_factory or DeviceExecutor
        self.ex = None
        self.n_random_nodes = len([n for n in self.nodes if isinstance(n, (N.Gaussian, N.Gamma, N.DiagonalGamma, N.Wishart))])
        # A graph a fused plan has handed over (a single X_t.update(), say) goes back to it when the standard loop resumes:
        # resume = {"pattern": the update() requests of one forward and one backward sweep, "left": hand-backs still allowed},
        # set by LDSPlan._demote.  Requests that follow the pattern wait in _buf; anything else runs them node by node.
        self.resume = None
        self._buf = []
        self._node_temp = {}            # per-node tape -> doubles of temporaries it uses
        self._pending = []              # update() requests not yet issued: a run of them becomes ONE tape (see _flush_pending)
        self.released = False
        if adopt:
            for nd in self.nodes:
                nd._plan = self

    # -- constants ------------------------------------------------------------------------------
    def const(self, v):
        v = float(v)
        key = np.float64(v).tobytes()
        if key not in self._consts:
            if len(self._const_vals) >= CONST_CAP:
                raise MemoryError("constant pool of the generic plan exhausted")
            self._consts[key] = Ref(len(self._const_vals), 1, 1)
            self._const_vals.append(v)
            self._const_dirty = True
        return self._consts[key]

    def ones(self, n):
        """A vector of n ones in the constant pool (for sums)."""
        if self._ones is None or self._ones.size < n:
            start = len(self._const_vals)
            if start + n > CONST_CAP:
                raise MemoryError("constant pool of the generic plan exhausted")
            self._const_vals.extend([1.0] * n)
            self._ones = Ref(start, n, 1)
            self._const_dirty = True
        return Ref(self._ones.off, n, 1)

    # -- running --------------------------------------------------------------------------------
    def _ensure_executor(self, need):
        if self.released:
            raise RuntimeError("this plan has been released: its graph is bound to another plan now")
        if self.ex is None or need > self.ex.size:
            old = self.ex
            size = max(need, self.temp_high) + 4096
            new = self._executor_factory(size)
            if old is None:
                for off, arr in self._init:
                    new.write(off, arr)
            else:                       # grown: carry the state over (constants and tapes are re-uploaded below)
                new.write(CONST_CAP, old.read(CONST_CAP, self.temp_base - CONST_CAP))
                old.close()
                self._tapes = {k: (None, ops, res) for k, (tid, ops, res) in self._tapes.items()}
            self.ex = new
            self._const_dirty = True

    def _run(self, key, build):
        """Run the tape `key`, building it with build(tape) -> result refs on first use."""
        if key not in self._tapes:
            t = Tape(self)
            res = build(t)
            self._tapes[key] = (None, t.array(), res)
            if t.marks:
                self._programs[key] = self._program(t)
        self._ensure_executor(self.temp_high)       # may replace the executor (a larger arena): every cached tape id is forgotten then,
        tid, ops, res = self._tapes[key]            # so the entry is read only afterwards
        if self._const_dirty:
            self.ex.write(0, np.array(self._const_vals, dtype=float))
            self._const_dirty = False
        if tid is None:
            tid = self.ex.tape(ops, self._programs.get(key))
            self._tapes[key] = (tid, ops, res)
        self.ex.run(tid)
        return res

    # Many-node tapes (update_all, llb_sum) mark where each node's records begin.  Nodes whose records neither read nor write a
    # posterior another one writes can run side by side: one workgroup per node instead of one workgroup for all of them.
    PAR_MIN = 8         # shorter independent runs stay in the sequential block: a launch costs more than they gain

    def _owner(self, off):
        """The node whose state slot holds arena offset `off` (None: constants, temporaries)."""
        if off < CONST_CAP or off >= self.temp_base:
            return None
        return self._slot_owner[bisect.bisect_right(self._slot_start, off) - 1]

    def _program(self, t):
        ops = t.ops
        cuts = sorted(set(m for m in t.marks if m < len(ops))) + [len(ops)]
        segs = []                                   # (first record, count, nodes read, nodes written)
        prologue = cuts[0] > 0                      # records before the first node (the result vector of llb_sum): they set up
        if prologue:                                # temporaries the nodes write into, so they run first and alone
            cuts = [0] + cuts
        for a, b in zip(cuts[:-1], cuts[1:]):
            if b == a:
                continue
            reads, writes = set(), set()
            for o in ops[a:b]:
                for v in (o[1], o[2], o[3], o[6], o[7]):          # any field may be an offset; small values are dimensions
                    w = self._owner(v)
                    if w is not None:
                        reads.add(w)
                for v in ((o[1], o[3]) if o[0] == T_CHOLINV else (o[1],)):
                    w = self._owner(v)
                    if w is not None:
                        writes.add(w)
            segs.append((a, b - a, reads, writes))
        blocks, launches = [], []

        def sequential(run):
            for a, n, _, _ in run:
                if launches and launches[-1][1] == 1 and blocks[-1][2] and blocks[-1][0] + blocks[-1][1] == a:
                    blocks[-1][1] += n              # extend the sequential block before it
                else:
                    blocks.append([a, n, True]); launches.append([len(blocks) - 1, 1])

        def flush(run):
            if len(run) >= self.PAR_MIN:
                launches.append([len(blocks), len(run)])
                blocks.extend([a, n, False] for a, n, _, _ in run)
            else:
                sequential(run)
        run, rr, rw = [], set(), set()
        for k, seg in enumerate(segs):
            _, _, reads, writes = seg
            if writes & (rr | rw) or reads & rw or (prologue and k == 1) or seg[0] in t.barriers:
                flush(run)
                run, rr, rw = [], set(), set()
            run.append(seg); rr |= reads; rw |= writes
        flush(run)
        if len(launches) == 1 and launches[0][1] == 1:
            return None                              # nothing to run side by side: the plain single-workgroup tape
        return (np.array([[a, n] for a, n, _ in blocks], dtype=np.int32), np.array(launches, dtype=np.int32))

    def _read(self, ref):
        self._ensure_executor(self.temp_high)
        return self.ex.read(ref.off, ref.size).reshape(ref.m, ref.n)

    # -- plan interface used by pyvb_amd.nodes / network -----------------------------------------
    def enqueue(self, node):
        if self.resume is not None and self._buffer(node):
            return
        self._pending.append(node)      # issued at the next read / write / other request (_flush_buf), together

    def _update_now(self, node):
        self._run(self._node_tape(node), None)

    def _buffer(self, node):
        """True if the request was taken into the buffer (it runs later: fused, or node by node at the next flush)."""
        pat = self.resume["pattern"]
        if node is pat[len(self._buf)]:
            self._buf.append(node)
            if len(self._buf) == len(pat):
                self._resume_fused()
            return True
        self._flush_buf()
        if node is pat[0]:
            self._buf.append(node)
            return True
        return False

    def _flush_buf(self):
        buf, self._buf = self._buf, []
        self._pending.extend(buf)
        self._flush_pending()

    # A hand-written loop -- [x.update() for x in Xs]; [z.update() for z in Zs]; Q.update() -- used to cost a launch per node.
    # The requests now wait until something needs their results and are then issued as ONE tape: the per-node tapes (built
    # once per node, as before) are concatenated, and the same analysis as for Network.learn's tapes (_program) finds the
    # nodes that touch disjoint state and lets them run side by side, one workgroup each; the temporaries of nodes that run
    # side by side are moved apart (every per-node tape allocates from the same base: _relocate).  The combined tape is kept
    # under the sequence of nodes, so the second pass of a loop re-uses it.  Nodes whose own tape already carries a program
    # (many children: their messages run side by side) are issued on their own.
    SEQ_CACHE = 64

    def _node_tape(self, node):
        key = ("update", id(node))
        if key not in self._tapes:
            t = Tape(self)
            res = self._emit_update_gaussian(t, node) if isinstance(node, N.Gaussian) else self._emit_update_noise(t, node)
            self._node_temp[key] = t.top - self.temp_base
            self._tapes[key] = (None, t.array(), res)
            if t.marks:
                self._programs[key] = self._program(t)
        return key

    def _relocate(self, ops, shift):
        """ops with every temporary (offset fields at or beyond temp_base) moved up by `shift`."""
        if shift == 0:
            return ops
        out = ops.copy()
        for code, fields in _OFFSET_FIELDS.items():
            rows = np.nonzero(ops[:, 0] == code)[0]
            if rows.size == 0:
                continue
            for f in fields:
                col = out[rows, f]
                if code == T_SCATTER and f == 7:
                    acc = col & T_ACC_BIT
                    val = col & ~T_ACC_BIT
                    out[rows, f] = np.where(val >= self.temp_base, val + shift, val) | acc
                else:
                    out[rows, f] = np.where(col >= self.temp_base, col + shift, col)
        return out

    def _flush_pending(self):
        seq, self._pending = self._pending, []
        run = []

        def issue(run):
            if not run:
                return
            if len(run) == 1:
                self._update_now(run[0])
                return
            key = ("seq",) + tuple(id(n) for n in run)
            if key not in self._tapes:
                if sum(1 for k in self._tapes if k[0] == "seq") >= self.SEQ_CACHE:       # bounded: forget the oldest sequences
                    for k in [k for k in self._tapes if k[0] == "seq"][:self.SEQ_CACHE // 2]:
                        tid = self._tapes.pop(k)[0]
                        self._programs.pop(k, None)
                        if tid is not None and self.ex is not None:
                            self.ex.drop(tid)
                parts = [self._tapes[self._node_tape(n)][1] for n in run]
                sizes = [len(p) for p in parts]
                t = Tape(self)
                t.ops = [tuple(int(v) for v in r) for p in parts for r in p]
                pos = 0
                for sz in sizes:
                    t.marks.append(pos)
                    pos += sz
                prog = self._program(t)
                ops = np.concatenate(parts).astype(np.int32)
                if prog is not None:
                    # nodes that share a launch get temporaries of their own: node k of a launch is moved up by k times the
                    # largest temporary area among that launch's nodes
                    blocks, launches = prog
                    starts = np.cumsum([0] + sizes)
                    node_of = {int(a): i for i, a in enumerate(starts[:-1])}
                    need = 0
                    plan_ok = True
                    moves = []
                    for first, count in launches:
                        if count == 1:
                            continue
                        idx = [node_of[int(blocks[first + k][0])] for k in range(count)]
                        assert all(int(blocks[first + k][1]) == sizes[i] for k, i in enumerate(idx)), "a parallel block is exactly one node's records"
                        span = max(self._node_temp[("update", id(run[i]))] for i in idx)
                        if self.temp_base + count * span >= (1 << 28):          # the arena is addressed with 32-bit offsets
                            plan_ok = False
                            break
                        moves.append((idx, span))
                        need = max(need, count * span)
                    if plan_ok:
                        for idx, span in moves:
                            for k, i in enumerate(idx):
                                a = int(starts[i])
                                ops[a:a + sizes[i]] = self._relocate(ops[a:a + sizes[i]], k * span)
                        self.temp_high = max(self.temp_high, self.temp_base + need)
                        self._programs[key] = prog
                self._tapes[key] = (None, ops, None)
            self._run(key, None)

        for nd in seq:
            if self._programs.get(self._node_tape(nd)) is not None:
                issue(run); run = []
                self._update_now(nd)
            else:
                run.append(nd)
        issue(run)

    def _resume_fused(self):
        """A forward and a backward sweep have queued up with nothing in between: the standard loop is back.  The state goes
        to the host attributes, the recogniser binds the graph anew (the fused plan, if it still is that graph) and the two
        sweeps are its first requests."""
        from . import _recognise
        buf, self._buf = self._buf, []
        left = self.resume["left"] - 1
        self.resume = None
        self.release()
        plan = _recognise.bind(buf[0])
        plan.resume_left = left
        for nd in buf:
            plan.enqueue(nd)

    def flush(self):
        self._flush_buf()

    def read(self, node, name):
        self._flush_buf()
        s = self.slot[id(node)]
        if isinstance(node, N.Gaussian):
            if name == "qmu":
                return self._read(s["qmu"])
            if name == "qcov":
                return self._read(s["qcov"])
            if name == "q_ln_det":
                v = float(self._read(s["qld"])[0, 0])
                if np.isnan(v):
                    raise AttributeError("q_ln_det is set by the first update()")
                return v
        elif name == "qb" and "qb" in s:
            v = self._read(s["qb"]).reshape(-1)
            return float(v[0]) if isinstance(node, N.Gamma) else v.copy()
        elif name == "qw" and "qw" in s:
            return self._read(s["qw"])
        raise AttributeError(name)

    def write(self, node, name, value):
        self._flush_buf()
        s = self.slot[id(node)]
        if name in ("qmu", "qcov", "qb", "qw") and name in s:
            self._ensure_executor(self.temp_high)
            v = np.asarray(value, dtype=float).reshape(-1)
            self.ex.write(s[name].off, np.ascontiguousarray(np.broadcast_to(v, (s[name].size,))))
        return True     # other attributes (q_ln_det, qprec) are host-only bookkeeping

    def pull(self):
        """Device state back into the nodes' host attributes (before the graph is re-bound)."""
        self._flush_buf()
        if self.ex is None:
            return
        for nd in self.nodes:
            s = self.slot[id(nd)]
            if isinstance(nd, N.Gaussian):
                nd.__dict__["_h_qmu"] = self._read(s["qmu"])
                nd.__dict__["_h_qcov"] = self._read(s["qcov"])
                q = float(self._read(s["qld"])[0, 0])
                if not np.isnan(q):
                    nd.__dict__["_h_q_ln_det"] = q
            elif isinstance(nd, (N.Gamma, N.DiagonalGamma)):
                v = self._read(s["qb"]).reshape(-1)
                nd.__dict__["_h_qb"] = float(v[0]) if isinstance(nd, N.Gamma) else v.copy()
            elif isinstance(nd, N.Wishart):
                nd.__dict__["_h_qw"] = self._read(s["qw"])

    def release(self):
        self.pull()
        for nd in self.nodes:
            if nd._plan is self:
                nd._plan = None
        if self.ex is not None:
            self.ex.close()
            self.ex = None
        self.released = True

    def node_llb(self, node):
        self._flush_buf()
        res = self._run(("llb", id(node)), lambda t: self._emit_llb(t, node))
        return float(self._read(res)[0, 0])

    def llb_sum(self, node_list):
        """sum of log_lower_bound() over the nodes (network.py:49), one launch."""
        self._flush_buf()
        key = ("llbsum",) + tuple(id(n) for n in node_list)

        def build(t):
            parts = t.zeros(len(node_list), 1)
            for i, n in enumerate(node_list):
                t.mark()
                t.copy(self._emit_llb(t, n), dst=parts.elem(i))
            return parts
        return self._read(self._run(key, build)).reshape(-1)

    def update_all(self, node_list):
        """[n.update() for n in node_list] as one launch (Network.learn, network.py:46-48)."""
        self._flush_buf()
        key = ("updall",) + tuple(id(n) for n in node_list)

        def build(t):
            for n in node_list:
                t.mark()
                if isinstance(n, N.Gaussian):
                    if not n.observed:
                        self._emit_update_gaussian(t, n)
                else:
                    self._emit_update_noise(t, n)
            return None
        self._run(key, build)

    def message(self, node, requester):
        """node.pass_up_m1_m2(requester) evaluated on the device; returns numpy arrays."""
        self._flush_buf()
        res = self._run(("msg", id(node), id(requester)), lambda t: self._pass_up(t, node, requester))
        return tuple(self._read(r) for r in res)

    def expectation(self, node, what):
        self._flush_buf()
        fn = {"Ex": self._ex, "ExxT": self._exxt, "ExTx": self._extx, "lndet": self._lndet}[what]
        return self._read(self._run(("exp", what, id(node)), lambda t: fn(t, node)))

    # =============================================================================================
    # emitters: the reference's methods
    # =============================================================================================
    def _ex(self, t, nd):
        """pass_down_Ex"""
        s = self.slot[id(nd)]
        if isinstance(nd, N.DiagonalGaussian):                   # gaussian.py:198-199
            return t.diag_matrix(s["qmu"])
        if isinstance(nd, N.Gaussian):                           # gaussian.py:154-160
            return s["qmu"]
        if isinstance(nd, N.Constant):                           # node.py:304-305
            return s["value"]
        if isinstance(nd, N.Addition):                           # node.py:112-119
            return t.add(self._ex(t, nd.A), self._ex(t, nd.B))
        if isinstance(nd, N.Multiplication):                     # node.py:235-242
            return t.gemm(self._ex(t, nd.A), self._ex(t, nd.B))
        if isinstance(nd, N.hstack):                             # nodes_todo.py:33-34
            rows, q = nd.shape
            out = t.tmp(rows, q)
            for i, p in enumerate(nd.parents):
                e = self._ex(t, p)
                t.emit(T_COPY2D, out.off + i, e.off, q, rows, 1, 1)
            return out
        if isinstance(nd, N.Gamma):                              # nodes_todo.py:140-142
            r = t.scale(s["qa"], s["qb"], divide=True)
            return t.scale(t.eye(nd.shape[0]), r)
        if isinstance(nd, N.DiagonalGamma):                      # nodes_todo.py:192-193
            return t.diag_matrix(t.mul(s["qa"], t.unary(s["qb"], U_RECIP)))
        if isinstance(nd, N.Wishart):                            # nodes_todo.py:233-234 (symmetric part, see k_wishart.hip)
            sym = t.axpby(0.5, s["qw"], 0.5, t.transpose(s["qw"]))
            inv, _ = t.cholinv(sym)
            return t.scale(inv, s["qv"])
        raise NotImplementedError("pass_down_Ex of %s" % type(nd).__name__)

    def _exxt(self, t, nd):
        """pass_down_ExxT"""
        s = self.slot[id(nd)]
        if isinstance(nd, N.DiagonalGaussian):                   # gaussian.py:200-201
            return t.diag_matrix(t.add(t.mul(s["qmu"], s["qmu"]), t.diag_of(s["qcov"])))
        if isinstance(nd, N.Gaussian):                           # gaussian.py:162-168
            return t.add(t.gemm(s["qmu"], s["qmu"], tb=True), s["qcov"])
        if isinstance(nd, N.Constant):
            return s["xxT"]
        if isinstance(nd, N.Addition):                           # node.py:121-129
            outer = t.gemm(self._ex(t, nd.A), self._ex(t, nd.B), tb=True)
            r = t.add(self._exxt(t, nd.A), self._exxt(t, nd.B))
            r = t.add(r, outer)
            return t.add(r, t.transpose(outer))
        if isinstance(nd, N.hstack):                             # nodes_todo.py:36-38
            r = None
            for p in nd.parents:
                e = self._exxt(t, p)
                r = e if r is None else t.add(r, e)
            return r
        if isinstance(nd, N.Multiplication):                     # node.py:244-276
            A, B = nd.A, nd.B
            if A.shape[1] == 1:                                  # rhs is a scalar (:251-252)
                return t.scale(self._exxt(t, A), self._exxt(t, B))
            if A.shape[0] == 1:                                  # lhs is a row vector (:253-254)
                return t.trace(t.gemm(self._exxt(t, B), self._extx(t, A)))
            BBT = self._exxt(t, B)
            if isinstance(A, N.Constant):                        # (:257-258, with the transpose it lacks)
                Am = self._ex(t, A)
                return t.gemm(Am, t.gemm(BBT, Am, tb=True))
            if hasattr(A, "parents"):                            # hstack (:260-271): sum_ij <a_i a_j^T> BBT[i,j]
                Am = self._ex(t, A)
                r = t.gemm(Am, t.gemm(BBT, Am, tb=True))
                for i, p in enumerate(A.parents):
                    t.axpby(1.0, r, BBT.elem(i, i), self.slot[id(p)]["qcov"], dst=r)
                return r
            sa = self.slot[id(A)]                                # DiagonalGaussian (:273-276)
            aaT = t.add(t.gemm(sa["qmu"], sa["qmu"], tb=True), sa["qcov"])
            return t.mul(BBT, aaT)
        raise NotImplementedError("pass_down_ExxT of %s" % type(nd).__name__)

    def _extx(self, t, nd):
        """pass_down_ExTx"""
        if isinstance(nd, N.DiagonalGaussian):
            return self._exxt(t, nd)
        if isinstance(nd, N.Gaussian):                           # gaussian.py:170-177
            return t.trace(self._exxt(t, nd))
        if isinstance(nd, N.Constant):
            return self.slot[id(nd)]["xTx"]
        raise NotImplementedError("pass_down_ExTx of %s" % type(nd).__name__)

    def _lndet(self, t, nd):
        """pass_down_lndet (quirk Q2: log det of the expected precision)"""
        s = self.slot[id(nd)]
        if isinstance(nd, N.Constant):                           # node.py:301-302, :310-311
            return s["lndet"]
        if isinstance(nd, N.Gamma):                              # nodes_todo.py:144-147
            d = t.sub(t.unary(s["qa"], U_LOG), t.unary(s["qb"], U_LOG))
            return t.scale(d, float(nd.shape[0]))
        if isinstance(nd, N.DiagonalGamma):                      # nodes_todo.py:195-197
            return t.total(t.sub(t.unary(s["qa"], U_LOG), t.unary(s["qb"], U_LOG)))
        if isinstance(nd, N.Wishart):                            # not in the reference (Q8): dim ln qv - ln det sym(qw)
            sym = t.axpby(0.5, s["qw"], 0.5, t.transpose(s["qw"]))
            _, o2 = t.cholinv(sym)
            return t.axpby(float(nd.shape[0]), t.unary(s["qv"], U_LOG), -2.0, o2.elem(1))
        raise NotImplementedError("pass_down_lndet of %s" % type(nd).__name__)

    def _ordered_sums(self, t, inits, count, item):
        """K sums at once: inits[k] + the terms item(0)[k], item(1)[k], ... (lists of Refs shaped like inits[k], or (factor, Ref)
        pairs, the factor a python float or a 1 x 1 Ref; inits[k] may be None: the sum then starts with the first term), added in
        that order.  From PAR_MIN items on, each item is emitted as a block of its own and its terms land in rows of one matrix
        per sum, which a single product with a row of ones adds up after a barrier -- the interpreter adds a short sum in that
        order and splits a long one (64 terms and more into few elements) over neighbouring lanes with a fixed tree of partial
        sums: deterministic, equal to the chain of additions up to rounding -- and the items (the messages of a node's children)
        are issued side by side (_program)."""
        K = len(inits)
        if count < self.PAR_MIN:
            accs = [None if a is None else t.copy(a) for a in inits]
            for i in range(count):
                for k, terms in enumerate(item(i)):
                    for term in terms:
                        f, term = term if isinstance(term, tuple) else (1.0, term)
                        if accs[k] is None:
                            accs[k] = t.copy(term) if (not isinstance(f, Ref) and f == 1.0) else t.axpby(f, term)
                        else:
                            t.axpby(1.0, accs[k], f, term, dst=accs[k])
            return accs
        rows, per, shape = [None] * K, [0] * K, [None] * K
        for i in range(count):
            t.mark()
            for k, terms in enumerate(item(i)):
                terms = [term if isinstance(term, tuple) else (1.0, term) for term in terms]
                if rows[k] is None:
                    per[k] = len(terms)
                    shape[k] = (terms[0][1].m, terms[0][1].n) if inits[k] is None else (inits[k].m, inits[k].n)
                    rows[k] = t.tmp((0 if inits[k] is None else 1) + count * per[k], shape[k][0] * shape[k][1])
                size, base = rows[k].n, 0 if inits[k] is None else 1
                assert len(terms) == per[k]
                for j, (f, term) in enumerate(terms):
                    assert term.size == size, "terms of different shapes cannot be summed (the reference raises there too)"
                    dst = Ref(rows[k].off + (base + i * per[k] + j) * size, 1, size)
                    if not isinstance(f, Ref) and f == 1.0:
                        t.copy(Ref(term.off, 1, size), dst=dst)
                    else:
                        t.axpby(f, Ref(term.off, 1, size), dst=dst)
        t.mark(barrier=True)
        out = []
        for k in range(K):
            size = rows[k].n
            if inits[k] is not None:
                t.copy(Ref(inits[k].off, 1, size), dst=Ref(rows[k].off, 1, size))
            ones = t.unary(t.zeros(1, rows[k].m), 5)           # exp(0): a row of ones of any length without the constant pool
            r = t.gemm(ones, rows[k])
            out.append(Ref(r.off, shape[k][0], shape[k][1]))
        return out

    def _sum_child_messages(self, t, nd):
        kids = nd.children
        if not kids:
            raise NotImplementedError("a %s without children was asked for a message" % type(nd).__name__)
        msgs = []

        def item(i):
            m = self._pass_up(t, kids[i], nd)
            msgs.append(m)
            return [m[0]], [m[1]]
        m1, m2 = self._ordered_sums(t, [None, None], len(kids), item)
        return m1, m2, msgs

    def _pass_up(self, t, nd, requester):
        """pass_up_m1_m2(requester): (m1, m2) -- or the 4-tuple of node.py:202 for an hstack requester"""
        if isinstance(nd, N.Gaussian):                           # gaussian.py:179-183
            pp = self._ex(t, nd.precision_parent)
            return pp, t.gemm(pp, self.slot[id(nd)]["qmu"])
        if isinstance(nd, N.Addition):                           # node.py:95-110
            m1, m2, _ = self._sum_child_messages(t, nd)
            other = nd.B if requester is nd.A else nd.A
            oe = self._ex(t, other)
            if m1.size == 1 and oe.size > 1:
                return m1, t.axpby(1.0, m2, t.scale(m1, -1.0), oe)
            return m1, t.gemm(m1, oe, dst=t.copy(m2), acc=True, neg=True)
        if isinstance(nd, N.Multiplication):                     # node.py:182-232
            m1s, m2s, _ = self._sum_child_messages(t, nd)
            A, B = nd.A, nd.B
            if requester is A:
                if A.shape[1] == 1:                              # lhs is a column, rhs a scalar (:195-197)
                    return t.scale(m1s, self._exxt(t, B)), t.scale(m2s, self._ex(t, B))
                if A.shape[0] == 1:                              # lhs a row vector (:198-200; `sumC` there is sum_child_m1s)
                    return t.scale(self._exxt(t, B), m1s), t.scale(t.transpose(self._ex(t, B)), m2s)
                return m1s, m2s, self._ex(t, B), self._exxt(t, B)        # hstack lhs (:201-202)
            Am = self._ex(t, A)
            m2 = t.gemm(Am, m2s, ta=True)                        # (:204)
            if A.shape[1] == 1:                                  # (:205-207)
                return t.trace(t.gemm(self._exxt(t, A), m1s if m1s.size > 1 else t.scale(t.eye(A.shape[0]), m1s))), m2
            if A.shape[0] == 1:                                  # (:208-209, returned as a pair here: Q3)
                return t.scale(self._extx(t, A), m1s), m2
            if m1s.size == 1:
                m1s = t.scale(t.eye(A.shape[0]), m1s)
            if isinstance(A, N.Constant):                        # (:211-212, as a pair: Q3)
                return t.gemm(Am, t.gemm(m1s, Am), ta=True), m2
            if hasattr(A, "parents"):                            # hstack (:213-227): <A^T L A> = Am^T L^T Am + diag_i tr(S_i L)
                m1 = t.gemm(Am, t.gemm(m1s, Am, ta=True), ta=True)
                for i, p in enumerate(A.parents):
                    tr = t.trace(t.gemm(self.slot[id(p)]["qcov"], m1s))
                    t.axpby(1.0, m1.elem(i, i), 1.0, tr, dst=m1.elem(i, i))
                return m1, m2
            sa = self.slot[id(A)]                                # DiagonalGaussian (:228-230)
            aaT = t.add(t.gemm(sa["qmu"], sa["qmu"], tb=True), sa["qcov"])
            return t.mul(aaT, m1s), m2
        if isinstance(nd, N.hstack):                             # nodes_todo.py:43-62
            if nd.shape[1] == 1:
                m1, m2, _ = self._sum_child_messages(t, nd)
                return m1, m2
            kids = nd.children
            i = nd.parents.index(requester)
            rows = nd.shape[0]

            def item(c):
                m = self._pass_up(t, kids[c], nd)
                if len(m) != 4:
                    raise NotImplementedError("an hstack matrix must be the left operand of its Multiplication children")
                cm1, cm2, bex, bbt = m
                if cm1.size == 1:
                    cm1 = t.scale(t.eye(rows), cm1)
                t2 = [(bex.elem(i), cm2)]                                       # (:60)
                for j, pj in enumerate(nd.parents):                             # (:61)
                    if j != i:
                        t2.append((t.scale(bbt.elem(i, j), -1.0), t.gemm(cm1, self._ex(t, pj))))
                return [(bbt.elem(i, i), cm1)], t2                              # (:56)
            m1, m2 = self._ordered_sums(t, [t.zeros(rows, rows), t.zeros(rows, 1)], len(kids), item)
            return m1, m2
        raise NotImplementedError("pass_up_m1_m2 of %s" % type(nd).__name__)

    # -- updates ----------------------------------------------------------------------------------
    def _emit_update_gaussian(self, t, nd):
        """Gaussian.update, gaussian.py:102-134"""
        s = self.slot[id(nd)]
        d = s["qmu"].m
        pmu = self._ex(t, nd.mean_parent)
        pprec = self._ex(t, nd.precision_parent)
        kids = nd.children

        def item(i):
            m = self._pass_up(t, kids[i], nd)
            m1, m2 = m[0], m[1]
            if m1.size == 1 and d > 1:
                m1 = t.scale(t.eye(d), m1)
            return [m1], [m2]
        qprec, wex = self._ordered_sums(t, [pprec, t.gemm(pprec, pmu)], len(kids), item)    # (:117, :122)
        t.cholinv(qprec, dst=s["qcov"], out2=s["qld"])                           # (:118-120)
        t.gemm(s["qcov"], wex, dst=s["qmu"])                                     # (:123)
        if nd.partially_observed:                                                # (:125-134)
            oi, ai, z = s["obs_index"], s["all_index"], s["zero_index"]
            cov_obs = t.gather(s["qcov"], oi, oi)
            cov_obs_inv, _ = t.cholinv(cov_obs)
            cov_obs_all = t.gather(s["qcov"], ai, oi)
            delta = t.sub(t.gather(s["obs_value"], oi, z), t.gather(s["qmu"], oi, z))
            gain = t.gemm(cov_obs_all, cov_obs_inv)
            t.gemm(gain, delta, dst=s["qmu"], acc=True)
            t.gemm(gain, cov_obs_all, tb=True, dst=s["qcov"], acc=True, neg=True)

    def _emit_update_noise(self, t, nd):
        s = self.slot[id(nd)]
        kids = nd.children

        def residual(i):                        # per child: <x x^T>, <mu mu^T>, <x><mu>^T (nodes_todo.py:138, :190, :231)
            c = kids[i]
            return self._exxt(t, c), self._exxt(t, c.mean_parent), t.gemm(self._ex(t, c), self._ex(t, c.mean_parent), tb=True)
        if isinstance(nd, N.Gamma):                              # nodes_todo.py:130-138
            def item(i):
                xx, mm, xm = residual(i)
                return [[(0.5, t.trace(xx)), (0.5, t.trace(mm)), (-1.0, t.trace(xm))]]
            acc, = self._ordered_sums(t, [t.const(float(nd.b0))], len(kids), item)
            t.copy(acc, dst=s["qb"])
        elif isinstance(nd, N.DiagonalGamma):                    # nodes_todo.py:187-190
            def item(i):
                xx, mm, xm = residual(i)
                return [[(0.5, t.diag_of(xx)), (0.5, t.diag_of(mm)), (-1.0, t.diag_of(xm))]]
            acc, = self._ordered_sums(t, [s["b0"]], len(kids), item)
            t.copy(acc, dst=s["qb"])
        elif isinstance(nd, N.Wishart):                          # nodes_todo.py:228-231 (w0 is not mutated: SURVEY Q7)
            def item(i):
                xx, mm, xm = residual(i)
                return [[(0.5, xx), (0.5, mm), (-1.0, xm)]]
            acc, = self._ordered_sums(t, [s["w0"]], len(kids), item)
            t.copy(acc, dst=s["qw"])
        else:
            raise NotImplementedError("update of %s" % type(nd).__name__)

    # -- lower bound ------------------------------------------------------------------------------
    def _emit_llb(self, t, nd):
        if isinstance(nd, N.Gaussian):
            return self._emit_llb_gaussian(t, nd)
        if isinstance(nd, (N.Gamma, N.DiagonalGamma, N.Wishart)):
            return self._emit_llb_noise(t, nd)
        return t.copy(t.const(0.0))                              # Node.log_lower_bound, node.py:42-43

    def _emit_llb_gaussian(self, t, nd):
        """Gaussian.log_lower_bound, gaussian.py:136-151"""
        s = self.slot[id(nd)]
        d = s["qmu"].m
        pp = self._ex(t, nd.precision_parent)
        inner = t.add(self._exxt(t, nd), self._exxt(t, nd.mean_parent))
        t.gemm(s["qmu"], self._ex(t, nd.mean_parent), tb=True, dst=inner, acc=True, neg=True)
        t.gemm(s["qmu"], self._ex(t, nd.mean_parent), tb=True, dst=inner, acc=True, neg=True)      # -2 qmu <mu>^T
        tr = t.trace(t.gemm(pp, inner))
        ret = t.lin(-0.5 * d * LN2PI, [(0.5, self._lndet(t, nd.precision_parent)), (-0.5, tr)])
        if not (nd.observed or nd.partially_observed):           # (:145-147)
            ret = t.lin(0.5 * d * LN2PI + 0.5 * d, [(1.0, ret), (0.5, s["qld"].elem(0))])
        elif nd.partially_observed:                              # (:148-150)
            mi = s["missing_index"]
            nm = mi.size
            _, o2 = t.cholinv(t.gather(s["qcov"], mi, mi))       # ln det = 2 sum log diag chol
            ret = t.lin(-0.5 * nm * LN2PI + 0.5 * nm, [(1.0, ret), (1.0, o2.elem(1))])
        return ret

    def _emit_llb_noise(self, t, nd):
        s = self.slot[id(nd)]
        if isinstance(nd, N.Wishart):
            # not in the reference (Q8): E ln p - E ln q in the (a, B) form, see pyvb_amd/csrc/k_wishart.hip
            dim = nd.shape[0]
            sym = t.axpby(0.5, s["qw"], 0.5, t.transpose(s["qw"]))
            inv, o2 = t.cholinv(sym)
            lndw = t.scale(o2.elem(1), 2.0)
            _, o0 = t.cholinv(s["w0"])
            lndw0 = t.scale(o0.elem(1), 2.0)
            EL = t.scale(inv, s["qv"])

            def multi(f, base, shift_consts):
                acc = t.copy(t.const(0.0))
                for i in range(dim):
                    t.axpby(1.0, acc, 1.0, t.unary(t.lin(-0.5 * i, [(1.0, base)]), f), dst=acc)
                return acc
            a0 = t.const(float(nd.v0))
            Eln = t.sub(multi(U_DIGAMMA, s["qv"], None), lndw)
            lg0 = t.lin(0.25 * dim * (dim - 1) * float(np.log(np.pi)), [(1.0, multi(U_LGAMMA, a0, None))])
            lgq = t.lin(0.25 * dim * (dim - 1) * float(np.log(np.pi)), [(1.0, multi(U_LGAMMA, s["qv"], None))])
            half = 0.5 * (dim + 1)
            p = t.lin(0.0, [(float(nd.v0) - half, Eln), (-1.0, lg0), (float(nd.v0), lndw0), (-1.0, t.trace(t.gemm(s["w0"], EL)))])
            qa_m = t.lin(-half, [(1.0, s["qv"])])
            q = t.lin(0.0, [(1.0, t.mul(qa_m, Eln)), (-1.0, lgq), (1.0, t.mul(s["qv"], lndw)), (-float(dim), s["qv"])])
            return t.sub(p, q)
        # Gamma nodes_todo.py:149-157, DiagonalGamma :199-204
        if isinstance(nd, N.Gamma):
            a0, b0 = t.const(float(nd.a0)), t.const(float(nd.b0))
        else:
            a0, b0 = s["a0"], s["b0"]
        qa, qb = s["qa"], s["qb"]
        n = qa.size
        one = self.ones(n)
        Elnx = t.sub(t.unary(qa, U_DIGAMMA), t.unary(qb, U_LOG))
        ratio = t.mul(qa, t.unary(qb, U_RECIP))
        p = t.mul(t.sub(a0, one), Elnx)
        p = t.sub(p, t.unary(a0, U_LGAMMA))
        p = t.add(p, t.mul(a0, t.unary(b0, U_LOG)))
        p = t.sub(p, t.mul(b0, ratio))
        q = t.mul(t.sub(qa, one), Elnx)
        q = t.sub(q, t.unary(qa, U_LGAMMA))
        q = t.add(q, t.mul(qa, t.unary(qb, U_LOG)))
        q = t.sub(q, t.mul(qb, ratio))
        return t.total(t.sub(p, q))
