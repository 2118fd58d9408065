"""bench.py itself on the GPU box, at a small shape: the JSON contract, the parity check on the timed batch, and the
multi-GPU code path as far as one GPU allows (RCCL communicator creation, the all-reduce inside the timed loop, the
failure policy when no communicator can be made)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SMALL = ["--replicates", "6", "--T", "400", "--D", "16", "--K", "16", "--steps", "3", "--warmup", "1", "--parity-replicates", "2"]


def _run(cmd, env=None, timeout=600):
    e = dict(os.environ)
    e.update(env or {})
    return subprocess.run(cmd, cwd=REPO, env=e, capture_output=True, text=True, timeout=timeout)


def _line(out):
    lines = [l for l in out.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out
    return json.loads(lines[0])


def test_single_gpu_line_and_parity_of_the_timed_batch():
    r = _run([sys.executable, "bench.py"] + SMALL)
    assert r.returncode == 0, r.stderr[-2000:]
    d = _line(r.stdout)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 3 and d["dtype"] == "f64" and d["vs_baseline"] is None
    assert d["config"]["elbo_rel_err_vs_numpy"] < 1e-8 and d["config"]["state_rel_err_vs_numpy"] < 1e-8
    assert "timed batch" in d["config"]["parity_checked_on"] and d["config"]["degraded"] is False
    rf = d["roofline"]
    assert rf["frac"] <= 1.0 and all(v["frac"] <= 1.0 for v in rf["secondary"].values())
    assert d["cpu_baseline"]["kind"] == "port" and d["cpu_baseline"]["value"] > 0


def test_one_rank_rccl_communicator_through_the_multi_gpu_branch():
    r = _run([sys.executable, "bench.py", "--rccl-single", "--no-cpu-baseline"] + SMALL)
    assert r.returncode == 0, r.stderr[-2000:]
    d = _line(r.stdout)
    assert d["config"]["collective"].startswith("rccl allreduce") and d["config"]["degraded"] is False


def test_two_ranks_on_one_gpu_fail_loudly_or_run_over_rccl():
    """Two processes on the one GPU of this box.  RCCL refuses two ranks on one device -- but only after its bootstrap
    and topology probe have worked, which is what this test is after: with torch in the process (its wheel bundles a second
    HSA runtime) ncclCommInitRank died earlier with 'no ROCm-capable device is detected', on any number of GPUs.  Without
    --allow-host-fallback a failed communicator must end the run with a non-zero status (never a silent host-side number);
    with it the line is marked degraded.  (If this RCCL build does accept the two ranks, the line must say rccl.)"""
    launch = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1"]
    env = {"HSA_ENABLE_IPC_MODE_LEGACY": "0", "NCCL_DEBUG": "WARN"}
    r = _run(launch + ["--master-port", "29611", "bench.py", "--gpus", "2", "--no-cpu-baseline"] + SMALL, env)
    if r.returncode == 0:
        d = _line(r.stdout)
        assert d["n_gpus"] == 2 and d["config"]["collective"].startswith("rccl") and d["config"]["degraded"] is False
        assert d["config"]["replicates_total"] == 12
    else:
        both = r.stderr + r.stdout
        assert "RCCL communicator could not be created" in both
        assert "Duplicate GPU detected" in both, both[-3000:]          # the legitimate reason, reached past bootstrap/topology
        assert "no ROCm-capable device" not in both
        r2 = _run(launch + ["--master-port", "29612", "bench.py", "--gpus", "2", "--no-cpu-baseline", "--allow-host-fallback"] + SMALL, env)
        assert r2.returncode == 0, r2.stderr[-2000:]
        d = _line(r2.stdout)
        assert d["n_gpus"] == 2 and d["config"]["degraded"] is True and "DEGRADED" in d["config"]["collective"]
        assert d["config"]["replicates_total"] == 12


def test_gpus_flag_starts_its_own_ranks():
    """`python bench.py --gpus 2` with no launcher and no WORLD_SIZE: the parent starts two fresh ranks itself and relays
    rank 0's line; the line must say n_gpus == 2 (round 2 ran ONE rank and said n_gpus 1).  On this one-GPU box RCCL
    refuses the second rank on the same device: without --allow-host-fallback the run fails naming the reason, with it the
    line is marked degraded."""
    env = {"HSA_ENABLE_IPC_MODE_LEGACY": "0", "NCCL_DEBUG": "WARN"}
    env_clean = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env_clean.update(env)
    r = subprocess.run([sys.executable, "bench.py", "--gpus", "2", "--no-cpu-baseline"] + SMALL, cwd=REPO, env=env_clean,
                       capture_output=True, text=True, timeout=600)
    if r.returncode == 0:
        d = _line(r.stdout)
        assert d["n_gpus"] == 2 and d["config"]["collective"].startswith("rccl") and d["config"]["degraded"] is False
        return
    both = r.stderr + r.stdout
    assert "Duplicate GPU" in both, both[-3000:]
    assert not [l for l in r.stdout.splitlines() if l.startswith("{")]          # no line from a failed run
    r2 = subprocess.run([sys.executable, "bench.py", "--gpus", "2", "--allow-host-fallback"] + SMALL, cwd=REPO, env=env_clean,
                        capture_output=True, text=True, timeout=600)
    assert r2.returncode == 0, r2.stderr[-2000:]
    d = _line(r2.stdout)
    assert d["n_gpus"] == 2 and d["config"]["replicates_total"] == 2 * 6
    assert d["config"]["degraded"] is True and "DEGRADED" in d["config"]["collective"]
    assert d["config"]["elbo_rel_err_vs_numpy"] < 1e-8


def test_one_rank_cannot_stand_for_several_gpus():
    r = _run([sys.executable, "bench.py", "--gpus", "2", "--no-cpu-baseline"] + SMALL, {"WORLD_SIZE": "1", "RANK": "0"})
    assert r.returncode != 0 and "WORLD_SIZE=1" in (r.stderr + r.stdout)
    assert not [l for l in r.stdout.splitlines() if l.startswith("{")]


def test_pca_workload_with_its_rows_sharded_over_two_ranks():
    """BASELINE configs[4] "batched over 8 GPUs": with --gpus N > 1 every rank joins the VB-PCA workload with its shard of the
    rows and rank 0 appends workloads.pca_config5 with n_gpus, the collective's byte count and `degraded`.  Two ranks on the
    one GPU of this box (host transport: RCCL refuses duplicate devices) against the single-rank run of the same problem."""
    env = {"HSA_ENABLE_IPC_MODE_LEGACY": "0"}
    env_clean = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env_clean.update(env)
    extra = ["--workloads", "pca_config5", "--pca-rows", "60000"]
    r1 = subprocess.run([sys.executable, "bench.py", "--gpus", "1"] + extra + SMALL, cwd=REPO, env=env_clean, capture_output=True, text=True, timeout=600)
    assert r1.returncode == 0, r1.stderr[-2000:]
    one = _line(r1.stdout)["workloads"]["pca_config5"]
    assert one["n_gpus"] == 1 and one["degraded"] is False and one["collective_bytes_per_step"] == 0 and one["rel_err_vs_numpy"] < 1e-8
    r2 = subprocess.run([sys.executable, "bench.py", "--gpus", "2", "--allow-host-fallback"] + extra + SMALL, cwd=REPO, env=env_clean,
                        capture_output=True, text=True, timeout=600)
    assert r2.returncode == 0, r2.stderr[-2000:]
    d = _line(r2.stdout)
    two = d["workloads"]["pca_config5"]
    assert d["n_gpus"] == 2 and two["n_gpus"] == 2 and two["scaling"] == "strong"
    if d["config"]["degraded"]:
        assert two["degraded"] is True and "DEGRADED" in two["collective"]
    else:
        assert two["collective"].startswith("rccl")
    assert two["collective_bytes_per_step"] == 8 * (4632 + 2 * 272)
    assert two["rel_err_vs_numpy"] < 1e-8                               # the sharded copy against the oracle
    assert abs(two["elbo_total"] - one["elbo_total"]) <= 1e-9 * abs(one["elbo_total"])      # the sharded run against the single-rank one
    assert "lds_d128" not in d["workloads"]
