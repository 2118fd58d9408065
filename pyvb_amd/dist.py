"""Host-side plumbing for multi-GPU runs: one process per GPU, replicates sharded contiguously.

The data path has exactly one exchange, the all-reduce of the six lower-bound parts; on GPUs it
runs inside libpyvb_hip.so over RCCL (pyvb_lds_comm_init / pyvb_lds_iterate).  This module
only provides the rendezvous around it -- barrier, max over ranks for timing, and the broadcast
of the RCCL unique id.

Two implementations of the same small interface:
  SocketComm  plain TCP over the loopback / MASTER_ADDR, standard library only: what bench.py uses.  It keeps torch out of
              the process: torch's wheel bundles its own libhsa-runtime64 / librccl, and a process that has loaded
              libpyvb_hip.so (system ROCm) AND torch ends up with two HSA runtimes -- RCCL's topology probe then runs in
              the uninitialised one and ncclCommInitRank fails ("no ROCm-capable device is detected"; seen on the GPU box,
              profiles/r02/README.md).
  GlooComm    torch.distributed's gloo backend, for the CPU tests of the sharding logic (tests/test_dist_gloo.py).
"""
import hashlib
import hmac
import json
import os
import socket
import struct
import time

import numpy as np

__all__ = ["init", "shard_range", "host_allreduce_callback", "LocalComm", "GlooComm", "SocketComm"]


def shard_range(n_total, rank, world):
    """Contiguous partition of n_total replicates: ranks 0..r-1 get one extra when it does not divide."""
    base, extra = divmod(int(n_total), int(world))
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def host_allreduce_callback(comm):
    """comm.allreduce_sum as the C callback of pyvb_lds_comm_init_host / pyvb_pca_comm_init_host (include/pyvb_hip.h):
    the library's collectives then travel through this process group instead of RCCL -- the rehearsal transport for
    several ranks on one GPU.  The caller keeps the returned object alive as long as the handle."""
    from . import _capi

    def fn(buf, count, user):
        try:
            a = np.ctypeslib.as_array(buf, shape=(int(count),))
            a[:] = comm.allreduce_sum(a)
            return 0
        except Exception:                   # an exception must not unwind through the C frame
            import traceback
            traceback.print_exc()
            return 1
    return _capi.HOST_ALLREDUCE(fn)


class LocalComm(object):
    rank, world = 0, 1

    def barrier(self):
        pass

    def max_float(self, x):
        return float(x)

    def broadcast_bytes(self, b):
        return b

    def allreduce_sum(self, a):
        return np.asarray(a, dtype=np.float64).copy()

    def close(self):
        pass


class GlooComm(object):
    def __init__(self, world, rank):
        import torch
        import torch.distributed as dist
        self._torch, self._dist = torch, dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        if not dist.is_initialized():
            dist.init_process_group("gloo", rank=rank, world_size=world)
        self.rank, self.world = rank, world

    def barrier(self):
        self._dist.barrier()

    def max_float(self, x):
        t = self._torch.tensor([float(x)], dtype=self._torch.float64)
        self._dist.all_reduce(t, op=self._dist.ReduceOp.MAX)
        return float(t[0])

    def broadcast_bytes(self, b):
        box = [b]
        self._dist.broadcast_object_list(box, src=0)
        return box[0]

    def allreduce_sum(self, a):
        t = self._torch.from_numpy(np.ascontiguousarray(a, dtype=np.float64).copy())
        self._dist.all_reduce(t, op=self._dist.ReduceOp.SUM)
        return t.numpy()

    def close(self):
        if self._dist.is_initialized():
            self._dist.destroy_process_group()


# ---- wire format of the TCP rendezvous.  Nothing received from a socket is ever unpickled or evaluated: a frame is
#   1 byte tag | u32 little-endian length | payload,   length <= MAX_FRAME
# with tags  N none | I int64 | F float64 | B bytes | A float64 array (payload = u32 ndim, ndim x u64 shape, raw doubles) |
# J a JSON object (the hello only).  Anything else closes the connection.
MAX_FRAME = 64 << 20


def _pack(obj):
    if obj is None:
        return b"N", b""
    if isinstance(obj, (bool, int, np.integer)):
        return b"I", struct.pack("<q", int(obj))
    if isinstance(obj, (float, np.floating)):
        return b"F", struct.pack("<d", float(obj))
    if isinstance(obj, (bytes, bytearray, memoryview)):
        return b"B", bytes(obj)
    if isinstance(obj, np.ndarray):
        a = np.ascontiguousarray(obj, dtype=np.float64)
        return b"A", struct.pack("<I", a.ndim) + struct.pack("<%dQ" % a.ndim, *a.shape) + a.tobytes()
    if isinstance(obj, dict):
        return b"J", json.dumps(obj, sort_keys=True).encode("utf-8")
    raise TypeError("the rendezvous carries None, int, float, bytes, float64 arrays and the hello object, not %r" % (type(obj),))


def _unpack(tag, data):
    if tag == b"N" and not data:
        return None
    if tag == b"I" and len(data) == 8:
        return struct.unpack("<q", data)[0]
    if tag == b"F" and len(data) == 8:
        return struct.unpack("<d", data)[0]
    if tag == b"B":
        return data
    if tag == b"A" and len(data) >= 4:
        (nd,) = struct.unpack_from("<I", data)
        if nd <= 8 and len(data) >= 4 + 8 * nd:
            shape = struct.unpack_from("<%dQ" % nd, data, 4)
            count = 1
            for s in shape:
                count *= s
            if 4 + 8 * nd + 8 * count == len(data):
                return np.frombuffer(data, dtype=np.float64, offset=4 + 8 * nd).reshape(shape).copy()
    if tag == b"J":
        obj = json.loads(data.decode("utf-8"))
        if isinstance(obj, dict):
            return obj
    raise ConnectionError("malformed rendezvous frame (tag %r, %d bytes)" % (tag, len(data)))


def _send(sock, obj):
    tag, data = _pack(obj)
    if len(data) > MAX_FRAME:
        raise ValueError("rendezvous frame of %d bytes exceeds the cap of %d" % (len(data), MAX_FRAME))
    sock.sendall(tag + struct.pack("<I", len(data)) + data)


HELLO_FRAME = 4096          # cap on a frame before the peer is admitted


def _recv(sock, cap=MAX_FRAME, deadline=None):
    """One frame.  cap: largest payload accepted; deadline: wall-clock time by which the WHOLE frame must have arrived (the
    socket's own time-out applies per recv() call, so a peer that trickles bytes could otherwise hold a connection open)."""
    def exact(n):
        buf = bytearray()
        while len(buf) < n:
            if deadline is not None:
                left = deadline - time.time()
                if left <= 0:
                    raise TimeoutError("rendezvous frame not complete in time")
                sock.settimeout(left)
            chunk = sock.recv(min(n - len(buf), 1 << 20))
            if not chunk:
                raise ConnectionError("peer closed the rendezvous connection")
            buf += chunk
        return bytes(buf)
    head = exact(5)
    (n,) = struct.unpack("<I", head[1:])
    if n > cap:
        raise ConnectionError("rendezvous frame of %d bytes exceeds the cap of %d" % (n, cap))
    return _unpack(head[:1], exact(n))


def _is_loopback(addr):
    return addr in ("localhost", "::1") or addr.startswith("127.")


class SocketComm(object):
    """Rank 0 listens, the others connect (MASTER_ADDR; a port derived from MASTER_PORT, which the launcher's own store
    occupies); every collective is a gather to rank 0 and a broadcast back.  A few bytes per call, a few calls per run.

    Admission is mutual: rank 0 sends a random challenge; the peer answers with a hello object (protocol token, run id, base
    port, world size, its rank), an HMAC-SHA256 of challenge + hello under a shared secret, and a challenge of its own; rank
    0 admits it only on a correct MAC and answers with the verdict and the MAC of the peer's challenge, which the peer checks
    before it trusts the listener (a rogue listener on one of the derived ports cannot admit ranks).  Every handshake has one
    wall-clock deadline (HELLO_TIMEOUT) and frames before admission are capped at 4 KB.  Frames after admission carry no
    MAC: integrity on a network that is not trusted is out of scope (one node, loopback, is what bench.py uses).  On the
    loopback interface the secret defaults to a fixed string (any local process could read the environment anyway); for any
    other MASTER_ADDR it must come from PYVB_RENDEZVOUS_SECRET, identical on every rank, or the constructor refuses to run."""
    TOKEN = "pyvb-rendezvous-3"
    HELLO_TIMEOUT = 5.0          # wall-clock limit of one handshake

    def __init__(self, world, rank, timeout=180.0):
        self.world, self.rank = int(world), int(rank)
        addr = os.environ.get("MASTER_ADDR", "127.0.0.1")
        base = int(os.environ.get("MASTER_PORT", "29500"))
        run = os.environ.get("TORCHELASTIC_RUN_ID", "none")
        secret = os.environ.get("PYVB_RENDEZVOUS_SECRET", "")
        if not secret:
            if not _is_loopback(addr):
                raise RuntimeError("rendezvous on %s (not the loopback interface) needs PYVB_RENDEZVOUS_SECRET set to the same "
                                   "value on every rank" % addr)
            secret = "pyvb-loopback"
        key = secret.encode("utf-8")
        ports = [20000 + (base * 31 + 97 * i) % 30000 for i in range(8)]
        hello = {"token": self.TOKEN, "run": run, "base": base, "world": self.world}

        def mac(challenge, obj):
            return hmac.new(key, challenge + json.dumps(obj, sort_keys=True).encode("utf-8"), hashlib.sha256).digest()

        deadline = time.time() + timeout
        if self.rank == 0:
            srv = None
            for p in ports:
                try:
                    srv = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
                    srv.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
                    srv.bind((addr if addr not in ("localhost",) else "127.0.0.1", p))
                    break
                except OSError:
                    srv.close()
                    srv = None
            if srv is None:
                raise RuntimeError("no free rendezvous port among %r" % (ports,))
            srv.listen(self.world)
            srv.settimeout(1.0)
            self.peers = {}
            while len(self.peers) < self.world - 1:
                if time.time() > deadline:
                    raise TimeoutError("rendezvous: %d of %d ranks connected" % (len(self.peers) + 1, self.world))
                try:
                    c, _ = srv.accept()
                except socket.timeout:
                    continue
                try:
                    c.settimeout(self.HELLO_TIMEOUT)
                    until = time.time() + self.HELLO_TIMEOUT
                    challenge = os.urandom(32)
                    _send(c, challenge)
                    msg, tag, theirs = (_recv(c, HELLO_FRAME, until), _recv(c, HELLO_FRAME, until), _recv(c, HELLO_FRAME, until))
                    r = msg.get("rank") if isinstance(msg, dict) else None
                    good = (isinstance(r, int) and 0 < r < self.world and r not in self.peers
                            and {k: msg.get(k) for k in hello} == hello
                            and isinstance(tag, bytes) and hmac.compare_digest(tag, mac(challenge, msg))
                            and isinstance(theirs, bytes) and len(theirs) == 32)
                    _send(c, 1 if good else 0)
                    if not good:
                        c.close()
                        continue
                    _send(c, mac(theirs, dict(hello, rank=0)))         # proof that the listener knows the secret too
                    c.settimeout(timeout)
                    self.peers[r] = c
                except Exception:               # a stranger, a garbled frame, a stalled peer: drop it, keep listening
                    c.close()
            srv.close()
        else:
            self.sock = None
            mine = dict(hello, rank=self.rank)
            while self.sock is None:
                if time.time() > deadline:
                    raise TimeoutError("rendezvous: rank %d found no rank 0 at %s ports %r" % (self.rank, addr, ports))
                for p in ports:
                    c = None
                    try:
                        c = socket.create_connection((addr, p), timeout=2.0)
                        c.settimeout(self.HELLO_TIMEOUT)
                        until = time.time() + self.HELLO_TIMEOUT
                        challenge = _recv(c, HELLO_FRAME, until)
                        if not isinstance(challenge, bytes) or len(challenge) != 32:
                            raise ConnectionError("not a pyvb rendezvous")
                        own = os.urandom(32)
                        _send(c, mine)
                        _send(c, mac(challenge, mine))
                        _send(c, own)
                        if _recv(c, HELLO_FRAME, until) == 1:
                            proof = _recv(c, HELLO_FRAME, until)
                            if not (isinstance(proof, bytes) and hmac.compare_digest(proof, mac(own, dict(hello, rank=0)))):
                                raise ConnectionError("the listener does not know the rendezvous secret")
                            c.settimeout(timeout)
                            self.sock = c
                            break
                        c.close()
                    except Exception:           # another service on that port, a garbled reply, a time-out: next port
                        if c is not None:
                            c.close()
                else:
                    time.sleep(0.2)

    def _gather_bcast(self, value, combine):
        if self.rank == 0:
            vals = [value] + [_recv(self.peers[r]) for r in sorted(self.peers)]
            out = combine(vals)
            for r in sorted(self.peers):
                _send(self.peers[r], out)
            return out
        _send(self.sock, value)
        return _recv(self.sock)

    def barrier(self):
        self._gather_bcast(0, lambda v: 0)

    def max_float(self, x):
        return float(self._gather_bcast(float(x), max))

    def broadcast_bytes(self, b):
        return self._gather_bcast(b if self.rank == 0 else None, lambda v: v[0])

    def allreduce_sum(self, a):
        return self._gather_bcast(np.asarray(a, dtype=np.float64), lambda v: np.sum(v, axis=0))

    def close(self):
        for c in (list(getattr(self, "peers", {}).values()) + ([self.sock] if getattr(self, "sock", None) else [])):
            try:
                c.close()
            except OSError:
                pass


def init(world=None, rank=None, backend="socket"):
    world = int(os.environ.get("WORLD_SIZE", "1")) if world is None else int(world)
    rank = int(os.environ.get("RANK", "0")) if rank is None else int(rank)
    if world <= 1:
        return LocalComm()
    return GlooComm(world, rank) if backend == "gloo" else SocketComm(world, rank)
