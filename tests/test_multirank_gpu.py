"""Several ranks drive libpyvb_hip.so at once on the ONE GPU of the box: rows (PCA) / replicates (LDS) sharded as on a
multi-GPU node, every collective of the library executed -- through the host transport of pyvb_*_comm_init_host, because
RCCL refuses two ranks on one device.  What this pins: the sharded algorithm (row offsets, who owns global row 0, the
order and contents of the collectives) against the single-rank result.  What it cannot pin: RCCL itself over xGMI."""
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
WORKER = os.path.join(HERE, "multirank_worker.py")


def _run(what, world, tmp_path, port, extra=()):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    prefix = str(tmp_path / ("%s_w%d" % (what, world)))
    procs = [subprocess.Popen([sys.executable, WORKER, what, str(r), str(world), prefix] + [str(x) for x in extra], env=env,
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(world)]
    outs = []
    for p in procs:
        try:
            out, _ = p.communicate(timeout=600)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        outs.append(out)
    assert all(p.returncode == 0 for p in procs), "\n".join(outs)[-3000:]
    return [dict(np.load(prefix + "_%d.npz" % r)) for r in range(world)]


def _rel(a, b):
    return float(np.abs(np.asarray(a) - np.asarray(b)).max() / max(np.abs(np.asarray(b)).max(), 1e-300))


@pytest.mark.parametrize("world", [2, 3])
def test_pca_rows_sharded_over_ranks(world, tmp_path):
    one = _run("pca", 1, tmp_path, 29610)[0]
    many = _run("pca", world, tmp_path, 29620 + world)
    assert [int(m["rows"][0]) for m in many] == sorted(int(m["rows"][0]) for m in many) and int(many[0]["rows"][0]) == 0
    for m in many:                                            # global quantities: the same on every rank, equal to one rank's
        for k in ("W_mean", "W_var", "Mu_mean", "Mu_var", "Z_cov", "beta_ab", "elbo"):
            assert _rel(m[k], one[k]) < 1e-10, (k, _rel(m[k], one[k]))
    for k in ("X", "Z", "X_rowvar"):                          # row quantities: the shards stack up to the whole
        assert _rel(np.concatenate([m[k] for m in many]), one[k]) < 1e-10, k


@pytest.mark.parametrize("seed", [3, 4, 5])
def test_pca_random_update_orders_over_three_ranks(seed, tmp_path):
    """Every update call is a collective when a communicator is attached: the same random sequence of calls (row ranges cut
    to each rank's shard, possibly empty; the single-row step through pyvb_pca_update_X0) on three ranks and on one."""
    one = _run("pcafuzz", 1, tmp_path, 29700 + seed, (seed,))[0]
    many = _run("pcafuzz", 3, tmp_path, 29710 + seed, (seed,))
    for m in many:
        for k in ("W_mean", "W_var", "Mu_mean", "Mu_var", "Z_cov", "beta_ab", "elbo", "elbos"):
            assert m[k].shape == one[k].shape and (m[k].size == 0 or _rel(m[k], one[k]) < 1e-9), (k, m[k], one[k])
    for k in ("X", "Z", "X_rowvar"):
        assert _rel(np.concatenate([m[k] for m in many]), one[k]) < 1e-9, k


def test_lds_replicates_sharded_over_ranks(tmp_path):
    one = _run("lds", 1, tmp_path, 29640)[0]
    many = _run("lds", 2, tmp_path, 29650)
    np.testing.assert_array_equal(np.concatenate([m["X"] for m in many]), one["X"])     # replicates are independent: bitwise
    total = sum(m["elbo_local"] for m in many)
    for m in many:
        assert _rel(m["elbo_total"], total) < 1e-13                                      # the all-reduce, on every rank
        assert _rel(m["elbo_total"], one["elbo_total"]) < 1e-12
        assert _rel(m["history"][-1], one["history"][-1]) < 1e-12                       # the side-stream history too


def test_baseline_config5_rows_sharded_over_four_ranks(tmp_path):
    """BASELINE configs[4] at full size (10^6 rows x 256, q = 16, 10 % missing), the rows sharded over four ranks as they
    would be over GPUs -- here four processes on the one GPU -- against the single-rank run of the same data."""
    shape = (1000000, 256, 16)
    one = _run("pcabig", 1, tmp_path, 29660, shape)[0]
    many = _run("pcabig", 4, tmp_path, 29670, shape)
    assert [tuple(m["rows"]) for m in many] == [(0, 250000), (250000, 500000), (500000, 750000), (750000, 1000000)]
    for m in many:
        for k in ("W_mean", "W_var", "Mu_mean", "Mu_var", "Z_cov", "beta_ab", "elbo"):
            assert _rel(m[k], one[k]) < 1e-9, (k, _rel(m[k], one[k]))
    assert _rel(many[0]["Z_head"], one["Z_head"]) < 1e-9 and _rel(many[0]["X_head"], one["X_head"]) < 1e-9
    assert _rel(sum(m["Z_sum"] for m in many), one["Z_sum"]) < 1e-8
    assert _rel(sum(m["X_sum"] for m in many), one["X_sum"]) < 1e-9
