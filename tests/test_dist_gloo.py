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
