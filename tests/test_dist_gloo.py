"""CPU, world_size > 1: the host-side multi-GPU plumbing (pyvb_amd.dist) -- replicate sharding, unique-id broadcast,
max-over-ranks timing and the one collective of the data path, the sum of the six lower-bound parts -- over both
rendezvous back ends: the standard-library TCP one bench.py uses and torch.distributed's gloo.  The per-rank parts come
from the oracle here; on GPUs the same reduction runs inside libpyvb_hip.so over RCCL (pyvb_lds_iterate)."""
import os
import subprocess
import sys

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r"""
import os, sys
import numpy as np
sys.path.insert(0, %(repo)r)
from pyvb_amd import dist, synth
from oracle import lds_closed_form as O

rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
comm = dist.init(world, rank, backend=os.environ["PYVB_TEST_BACKEND"])
assert type(comm).__name__ == {"socket": "SocketComm", "gloo": "GlooComm"}[os.environ["PYVB_TEST_BACKEND"]]
if os.environ["PYVB_TEST_BACKEND"] == "socket":
    assert "torch" not in sys.modules          # the point of the socket back end
N_total, T, D, K = 5, 30, 3, 4
Y, st0, pri = synth.make_problem(T, D, K, N_total, 77)
lo, hi = dist.shard_range(N_total, rank, world)
st = O.expand_state({k: v[lo:hi] for k, v in st0.items()}, pri, T)
parts = O.iterate(st, pri, Y[lo:hi])
total = comm.allreduce_sum(parts.sum(0))
uid = comm.broadcast_bytes(bytes(range(128)) if rank == 0 else None)
tmax = comm.max_float(1.0 + rank)
comm.barrier()
np.save(os.path.join(%(out)r, "rank%%d.npy" %% rank), np.concatenate([total, [tmax, float(len(uid)), float(lo), float(hi)]]))
comm.close()
"""


def test_shard_range_covers_everything():
    from pyvb_amd.dist import shard_range
    for n, w in [(8192, 8), (5, 2), (7, 3), (3, 8)]:
        parts = [shard_range(n, r, w) for r in range(w)]
        assert parts[0][0] == 0 and parts[-1][1] == n
        assert all(parts[i][1] == parts[i + 1][0] for i in range(w - 1))
        sizes = [b - a for a, b in parts]
        assert max(sizes) - min(sizes) <= 1


@pytest.mark.parametrize("backend,port", [("socket", "29541"), ("gloo", "29543")])
def test_two_ranks(tmp_path, backend, port):
    script = tmp_path / "worker.py"
    script.write_text(WORKER % {"repo": REPO, "out": str(tmp_path)})
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=port, OMP_NUM_THREADS="1", PYVB_TEST_BACKEND=backend)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
           "--master-addr", "127.0.0.1", "--master-port", port, str(script)]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    a, b = np.load(tmp_path / "rank0.npy"), np.load(tmp_path / "rank1.npy")
    assert np.array_equal(a[:6], b[:6])                       # both ranks hold the global sum
    assert a[6] == 2.0 and b[6] == 2.0                        # max over ranks
    assert a[7] == 128 and b[7] == 128                        # the RCCL unique id travels as 128 bytes
    assert (a[8], a[9], b[8], b[9]) == (0, 3, 3, 5)           # contiguous shards
    # the global sum equals the single-process result
    from pyvb_amd import synth
    from oracle import lds_closed_form as O
    Y, st0, pri = synth.make_problem(30, 3, 4, 5, 77)
    st = O.expand_state(st0, pri, 30)
    ref = O.iterate(st, pri, Y).sum(0)
    assert np.allclose(a[:6], ref, rtol=1e-12)


def test_five_ranks_over_sockets(tmp_path):
    """More than one peer per collective, ranks started in any order (no launcher: plain processes with the environment
    the launcher would give them)."""
    script = tmp_path / "worker.py"
    script.write_text(WORKER % {"repo": REPO, "out": str(tmp_path)})
    procs = []
    for rank in (3, 1, 4, 0, 2):
        env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29547", OMP_NUM_THREADS="1", PYVB_TEST_BACKEND="socket",
                   RANK=str(rank), WORLD_SIZE="5", LOCAL_RANK=str(rank))
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    for p in procs:
        out, _ = p.communicate(timeout=300)
        assert p.returncode == 0, out[-2000:]
    res = [np.load(tmp_path / ("rank%d.npy" % r)) for r in range(5)]
    assert all(np.array_equal(r[:6], res[0][:6]) for r in res) and all(r[6] == 5.0 for r in res)
    assert [(int(r[8]), int(r[9])) for r in res] == [(0, 1), (1, 2), (2, 3), (3, 4), (4, 5)]


def test_wire_format_round_trips_and_never_unpickles():
    """The rendezvous frames are typed raw bytes (pyvb_amd/dist.py): every value the collectives carry survives a round trip,
    an oversized or malformed frame raises instead of being interpreted, and nothing on the wire is a pickle."""
    import socket
    import struct
    from pyvb_amd import dist
    a, b = socket.socketpair()
    try:
        vals = [None, 0, -7, 2.5, b"\x00\x01" * 64, np.arange(6.0), np.arange(12.0).reshape(3, 4), {"token": "t", "rank": 1}]
        for v in vals:
            dist._send(a, v)
            got = dist._recv(b)
            if isinstance(v, np.ndarray):
                assert got.dtype == np.float64 and np.array_equal(got, v)
            else:
                assert got == v and type(got) is type(v)
        with pytest.raises(TypeError):
            dist._send(a, [1, 2])                                  # lists, tuples, objects: not carried
        import pickle
        blob = pickle.dumps(("x",), protocol=4)                    # what round 2's wire looked like: an 8-byte length + pickle
        a.sendall(struct.pack("<Q", len(blob)) + blob)
        with pytest.raises((ConnectionError, ValueError)):
            dist._recv(b)
    finally:
        a.close()
        b.close()
    a, b = socket.socketpair()
    try:
        a.sendall(b"B" + struct.pack("<I", dist.MAX_FRAME + 1))    # a length beyond the cap is refused before any allocation
        with pytest.raises(ConnectionError):
            dist._recv(b)
        a.sendall(b"A" + struct.pack("<I", 12) + struct.pack("<I", 1) + struct.pack("<Q", 1 << 40))   # shape disagrees with size
        with pytest.raises(ConnectionError):
            dist._recv(b)
    finally:
        a.close()
        b.close()
    assert "pickle" not in open(dist.__file__).read().replace("unpickled", "").replace("a pickle", "")


def test_rendezvous_refuses_strangers_and_non_loopback_without_secret(monkeypatch):
    """Rank 0 keeps listening after a peer that sends garbage or a wrong MAC; a non-loopback MASTER_ADDR without
    PYVB_RENDEZVOUS_SECRET is refused outright."""
    import socket
    import threading
    from pyvb_amd import dist
    monkeypatch.setenv("MASTER_ADDR", "10.1.2.3")
    monkeypatch.delenv("PYVB_RENDEZVOUS_SECRET", raising=False)
    with pytest.raises(RuntimeError, match="PYVB_RENDEZVOUS_SECRET"):
        dist.SocketComm(2, 1, timeout=1.0)
    monkeypatch.setenv("MASTER_ADDR", "127.0.0.1")
    monkeypatch.setenv("MASTER_PORT", "29549")
    box = {}

    def rank0():
        box["c"] = dist.SocketComm(2, 0, timeout=60.0)
    th = threading.Thread(target=rank0)
    th.start()
    base = 29549
    port = 20000 + (base * 31) % 30000
    import time
    for attempt in range(100):
        try:
            s = socket.create_connection(("127.0.0.1", port), timeout=1.0)
            break
        except OSError:
            time.sleep(0.1)
    ch = dist._recv(s)
    assert isinstance(ch, bytes) and len(ch) == 32
    s.sendall(b"\x80\x04garbage-that-is-not-a-frame" * 3)          # a stranger
    s.close()
    s = socket.create_connection(("127.0.0.1", port), timeout=2.0)
    ch = dist._recv(s)
    dist._send(s, {"token": dist.SocketComm.TOKEN, "run": "none", "base": base, "world": 2, "rank": 1})
    dist._send(s, b"\x00" * 32)                                     # right hello, wrong MAC
    dist._send(s, b"\x01" * 32)                                     # (the peer's own challenge)
    assert dist._recv(s) == 0
    s.close()
    # a stranger that trickles: half a frame header and then nothing -- the handshake has ONE deadline, the listener moves on
    t0 = time.time()
    s = socket.create_connection(("127.0.0.1", port), timeout=2.0)
    dist._recv(s)
    s.sendall(b"J\x10")
    # an oversized frame before admission is refused without being read
    s2 = socket.create_connection(("127.0.0.1", port), timeout=10.0)
    peer = dist.SocketComm(2, 1, timeout=30.0)                      # the real rank 1 still gets in
    assert time.time() - t0 < 3 * dist.SocketComm.HELLO_TIMEOUT + 5
    s.close(); s2.close()
    th.join(60)
    assert not th.is_alive()
    t = threading.Thread(target=lambda: box.__setitem__("m", box["c"].max_float(1.0)))
    t.start()
    assert peer.max_float(3.0) == 3.0
    t.join(30)
    assert box["m"] == 3.0
    peer.close()
    box["c"].close()


def test_a_listener_that_does_not_know_the_secret_admits_nobody(monkeypatch):
    """The handshake is mutual: a rogue listener on one of the derived ports can collect a hello and its MAC, but it cannot
    answer the peer's own challenge, and the peer walks away from it."""
    import socket
    import threading
    import time
    from pyvb_amd import dist
    monkeypatch.setenv("MASTER_ADDR", "127.0.0.1")
    monkeypatch.setenv("MASTER_PORT", "29561")
    monkeypatch.delenv("PYVB_RENDEZVOUS_SECRET", raising=False)
    base = 29561
    port = 20000 + (base * 31) % 30000
    srv = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
    srv.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
    srv.bind(("127.0.0.1", port)); srv.listen(4); srv.settimeout(20.0)
    seen = {}

    def rogue():
        try:
            c, _ = srv.accept()
            c.settimeout(5.0)
            dist._send(c, b"\x07" * 32)
            seen["hello"] = dist._recv(c); dist._recv(c); seen["theirs"] = dist._recv(c)
            dist._send(c, 1)                         # "admitted" ...
            dist._send(c, b"\x00" * 32)              # ... but no proof
            time.sleep(0.5)
            c.close()
        except Exception as e:                       # noqa
            seen["err"] = repr(e)
    th = threading.Thread(target=rogue)
    th.start()
    with pytest.raises(TimeoutError):
        dist.SocketComm(2, 1, timeout=4.0)           # never connected: the only listener failed the proof
    th.join(30)
    srv.close()
    assert isinstance(seen.get("theirs"), bytes) and seen["hello"]["rank"] == 1
