"""bench.py's own launcher, as far as a machine without a GPU can take it: `--gpus N` starts N ranks by itself; a failing
rank (here: every rank, no device) makes the whole run fail without printing a line; one rank never stands for N GPUs."""
import os
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SMALL = ["--replicates", "2", "--T", "50", "--D", "4", "--K", "4", "--steps", "1", "--warmup", "0", "--no-cpu-baseline"]


def _env(**kw):
    e = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR")}
    e.update(kw)
    return e


def _has_gpu():
    return os.path.exists("/dev/kfd")


def test_world_size_mismatch_is_refused_before_any_gpu_call():
    r = subprocess.run([sys.executable, "bench.py", "--gpus", "4"] + SMALL, cwd=REPO, env=_env(WORLD_SIZE="1", RANK="0"),
                       capture_output=True, text=True, timeout=120)
    assert r.returncode != 0 and "--gpus 4 but WORLD_SIZE=1" in r.stderr
    r = subprocess.run([sys.executable, "bench.py", "--gpus", "1"] + SMALL, cwd=REPO, env=_env(WORLD_SIZE="2", RANK="0"),
                       capture_output=True, text=True, timeout=120)
    assert r.returncode != 0 and "--gpus 1 but WORLD_SIZE=2" in r.stderr


def test_self_launch_spawns_ranks_and_propagates_failure():
    if _has_gpu():
        import pytest
        pytest.skip("covered by tests/test_bench_gpu.py on a GPU box")
    r = subprocess.run([sys.executable, "bench.py", "--gpus", "2"] + SMALL, cwd=REPO, env=_env(), capture_output=True, text=True,
                       timeout=300)
    assert r.returncode != 0
    assert "rank" in r.stderr and "stopping the other ranks" in r.stderr            # the parent saw a child die
    assert not [l for l in r.stdout.splitlines() if l.startswith("{")]


def test_launcher_relays_exactly_rank_zero(tmp_path):
    """launch_ranks with a stand-in script: two children, rank 0 prints a line with n_gpus = 2 -> relayed, rc 0; a line with
    the wrong n_gpus -> rc 1; a failing rank 1 -> its status."""
    fake = tmp_path / "fake.py"
    fake.write_text(
        "import os, sys, json, time\n"
        "r = int(os.environ['RANK']); mode = sys.argv[1]\n"
        "assert os.environ['WORLD_SIZE'] == '2' and os.environ['MASTER_ADDR'] == '127.0.0.1' and int(os.environ['MASTER_PORT']) > 0\n"
        "if mode == 'fail' and r == 1: sys.exit(7)\n"
        "if mode == 'fail' and r == 0: time.sleep(60)\n"
        "if r == 0: print('noise'); print(json.dumps({'n_gpus': 1 if mode == 'wrong' else 2, 'value': 1.0}))\n"
        "else: print(json.dumps({'n_gpus': 99}))\n")
    for mode, want in (("ok", 0), ("wrong", 1), ("fail", 7)):
        code = subprocess.run([sys.executable, "-c",
                               "import sys; sys.path.insert(0, %r); import bench; bench.__file__ = %r; "
                               "sys.exit(bench.launch_ranks(2, [%r]))" % (REPO, str(fake), mode)],
                              capture_output=True, text=True, timeout=120)
        assert code.returncode == want, (mode, code.returncode, code.stderr)
        lines = [l for l in code.stdout.splitlines() if l.startswith("{")]
        assert (len(lines) == 1 and '"n_gpus": 2' in lines[0]) if mode == "ok" else not lines
