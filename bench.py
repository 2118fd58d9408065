#!/usr/bin/env python3
"""bench.py -- VB update iterations/sec of the LDS path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W]           (N > 1: starts its own N ranks, one per GPU)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One "step" = one variational iteration of every replicate resident on the rank's GPU:
forward sweep, backward sweep, A columns, C columns, Q, R, lower bound, plus the ELBO
reduction over replicates (and the RCCL all-reduce over ranks when N > 1).  Workload at
N = 1 is BASELINE.json configs[2]: T = 10^4, D = K = 64, 1024 replicates, fp64; each extra
GPU adds another 1024 replicates (weak scaling; N = 8 is configs[3]).  Inputs are synthetic
and already resident in HBM when the timed region starts.

Rank 0 prints one JSON line.  `value` counts iterations of a 1024-replicate block per second
summed over all GPUs, so it equals plain iterations/s at N = 1 and adds up across ranks.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

FP64_MFMA_PEAK_TFLOPS = 78.6      # MI355X vendor figure for FP64 matrix (= FP64 vector); a pure MFMA loop sustains 77 (profiles/r02/mb_power.txt)
HBM_PEAK_GBS = 8000.0
# With real data the sweeps run the chip into its power cap (1343 W, shader clock 2055 of 2400 MHz: profiles/r02/limits.txt);
# the matrix pipe's peak at that clock, reported next to the contract figure:
CAPPED_CLOCK_RATIO = 2055.0 / 2400.0


def make_inputs(T, D, K, N, seed):
    """N replicates: up to 128 distinct simulated systems tiled to N, every replicate with its own
    initial parameter posterior (so all N chains compute different things)."""
    from pyvb_amd import synth
    base = min(N, 128)
    Y, st0, pri = synth.make_problem(T, D, K, base, seed)
    if max(D, K) > 102:
        # the reference's ln det of a column's prior precision goes through np.linalg.det (quirk Q2): det(1e-3 I)
        # underflows from 103 dimensions on and its lower bound is -inf; a prior precision with a representable determinant
        # keeps the parity check of the bound meaningful at these sizes
        pri["A_prior_prec"] = np.full_like(pri["A_prior_prec"], 1e-2)
        pri["C_prior_prec"] = np.full_like(pri["C_prior_prec"], 1e-2)
    rep = (N + base - 1) // base
    if rep > 1:
        Y = np.concatenate([Y] * rep)[:N]
        big = synth.initial_state(1, D, K, N, seed + 1)          # cheap: T = 1
        st = {k: np.concatenate([v] * rep)[:N] for k, v in st0.items()}
        for k in ("A_mean", "C_mean", "A_colvar", "C_colvar", "Q_b", "R_b"):
            st[k] = big[k]
        st0 = st
    return Y, st0, pri


def cpu_baseline_and_parity(sample, pri, iters, got_elbo, got_X):
    """Oracle (numpy port of the reference's loop, oracle/lds_closed_form.py) run on the first replicates OF THE
    TIMED BATCH for the same number of iterations (warm-up + timed), timed on this host's cores: that is both the
    CPU baseline (a bounded sample of the same workload) and the parity check of the numbers just timed."""
    from oracle import lds_closed_form as O
    Y, st0 = sample
    n_s, T, K = Y.shape
    D = st0["A_mean"].shape[1]
    st = O.expand_state(st0, pri, T)
    t0 = time.perf_counter()
    for _ in range(iters):
        ref = O.iterate(st, pri, Y)
    cpu_s = time.perf_counter() - t0
    rel = float(np.max(np.abs(got_elbo[:n_s].sum(1) - ref.sum(1)) / np.abs(ref.sum(1))))
    rel_x = float(np.abs(got_X[:n_s] - st["X"]).max() / np.abs(st["X"]).max())
    try:
        from threadpoolctl import threadpool_info
        cores = max([p.get("num_threads", 1) for p in threadpool_info()] or [1])
    except Exception:
        cores = 1
    base = {
        "value": (n_s / 1024.0) * iters / cpu_s,
        "unit": "VB iterations/s per 1024 replicates",
        "cores": int(cores),
        "kind": "port",
        "sample": "replicates 0..%d of the timed batch x %d iterations at T=%d D=%d K=%d (%.1f s), scaled by replicates/1024"
                  % (n_s - 1, iters, T, D, K, cpu_s),
    }
    # the reference itself cannot run on the GPU box (Python 2 source, never shipped); its rate is measured in the build
    # container by profiles/reference_cpu.py at the same D, K and a short chain, and extrapolated (cost is linear in T
    # and in the number of replicates): reported next to the port, from the committed file
    for tag in ("r04", "r03", "r02"):
        path = os.path.join(REPO, "profiles", tag, "reference_cpu.json")
        if os.path.exists(path):
            base["reference_extrapolated"] = json.load(open(path))
    return base, rel, rel_x


def pca_workload(steps, warmup, with_cpu, comm=None, rank=0, world=1, device=0, transport="none", rows=1000000):
    """BASELINE configs[4], appended to the headline line as workloads.pca_config5: VB-PCA with missing data
    (examples/PCA_missing_data.py:31-45 of the reference), N = 10^6 rows x d = 256, q = 16, 10 % missing.  Measured like the
    headline: inputs resident before the timed region, `steps` iterations after `warmup`, a barrier on both sides, the maximum
    over the ranks.  With world > 1 the ROWS of the one model are sharded over the ranks (strong scaling: the total stays
    10^6; every rank generates its own shard) and every iteration carries three all-reduces -- the statistics vector after
    the sweep, [sum z | delta sum x] after the Z step and after the X_0 step (pyvb_amd/csrc/api_pca.hip) -- over RCCL, or over
    the rendezvous sockets when the caller allowed the degraded transport.  Parity: a 20000-row copy of the problem, sharded
    the same way over the same transport, against oracle/pca_closed_form.py on rank 0, which is also the CPU baseline
    (scaled by rows: the cost is linear in N).  Every rank must call this; rank 0 gets the entry, the others None."""
    from pyvb_amd import synth, dist as pdist
    from pyvb_amd.pca import PCABatch
    from pyvb_amd.lds import LDSBatch
    N, d, q = int(rows), 256, 16
    pri = synth.pca_problem(16, d, q, 33)[1]

    def shard(n_total):
        lo, hi = pdist.shard_range(n_total, rank, world)
        X, obs, Z0, W0 = synth.pca_rows(lo, hi, d, q, 33)
        b = PCABatch(hi - lo, d, q, device, N_total=n_total, row_offset=lo)
        if world > 1:
            if transport == "rccl":
                uid = comm.broadcast_bytes(LDSBatch.comm_unique_id() if rank == 0 else None)
                b.comm_init(uid, rank, world)
            else:
                b.comm_init_host(comm, rank, world)
        b.set_priors(pri)
        b.set_data(np.where(obs, X, np.nan))
        b.set_state(X_missing=np.where(obs, X, 0.0), W_mean=W0, Z=Z0, Z_cov=np.eye(q), Mu_mean=np.zeros(d), beta_b=1.0)
        return b, float((~obs).sum())

    b, nmiss = shard(N)
    # an iteration takes a millisecond: at the headline's 3 + 20 steps the timed region would be 20 ms, taken right after seconds of
    # host-side data generation with the GPU idle (measured: 1.00 ms per step that way against 0.94 over 100 steps).  The timed loop
    # of this workload is therefore at least 10 + 100 steps; the parity copy below keeps the headline's counts (the oracle runs them too).
    t_warm, t_steps = max(warmup, 10), max(steps, 100)
    b.iterate(t_warm); b.sync()
    if comm is not None:
        comm.barrier()
    t0 = time.perf_counter()
    b.iterate(t_steps); b.sync()
    if comm is not None:
        comm.barrier()
    dt = time.perf_counter() - t0
    if comm is not None:
        dt = comm.max_float(dt)
        nmiss = float(comm.allreduce_sum(np.array([nmiss]))[0])
    elbo = b.elbo()                         # global: the statistics are all-reduced
    b.close()
    n_s = min(20000, N)
    sb, _ = shard(n_s)
    sb.iterate(warmup + steps)
    got = sb.get_state(); ge = sb.elbo(); sb.close()
    if rank != 0:
        return None
    parity, cpu = None, None
    if with_cpu:
        from oracle import pca_closed_form as P        # checker and CPU baseline only
        sinit, _ = synth.pca_problem(n_s, d, q, 33)
        sst = P.make_state(sinit, pri, n_s, d, q)
        t1 = time.perf_counter()
        for _ in range(warmup + steps):
            ref = P.iterate(sst, pri)
        cpu_dt = time.perf_counter() - t1
        rel = lambda x, y: float(np.abs(x - y).max() / np.abs(y).max())
        m = got["Z"].shape[0]                           # rank 0's rows of the copy
        parity = max(rel(got["W_mean"], sst["W_mean"]), rel(got["Z"], sst["Z"][:m]), rel(got["X"], sst["X"][:m]),
                     float(np.abs(ge - ref).max() / np.abs(ref).sum()))
        cpu = {"value": (warmup + steps) / cpu_dt * n_s / N, "unit": "VB iterations/s at N=%d (scaled from the sample)" % N,
               "cores": os.cpu_count(), "kind": "port",
               "sample": "%d rows, %d iterations of oracle/pca_closed_form.py in %.1f s" % (n_s, warmup + steps, cpu_dt)}
    # algorithmic bytes of an iteration: X and the byte mask are read once (the Z and the X updates share one sweep over the rows,
    # k_pca_pass12), the previous Z is read and the new one written; the imputed entries are not stored (round 4: the next sweep
    # recomputes them from Z and the parameters).  (Two sweeps -- the reference's order taken literally, and this path before
    # round 3 -- move X twice and Z three times and write the missing entries back: `algorithmic_bytes_two_sweeps`.)
    alg = 1.0 * N * d * 8 + N * d + 2.0 * N * q * 8
    alg2 = 2.0 * N * d * 8 + N * d + 2.0 * N * q * 8 + nmiss * 8
    traffic, tsrc = None, None
    for tag in ("r04", "r03"):
        tpath = os.path.join(REPO, "profiles", tag, "traffic_pca_pmc.json")
        if world == 1 and N == 1000000 and os.path.exists(tpath):
            tj = json.load(open(tpath))
            hit = [(k, v["hbm_bytes_per_launch"]) for k, v in tj.items() if "k_pca_pairs" in k or "k_pca_pass12" in k]
            if hit:
                traffic = hit[0][1]
                tsrc = "profiles/%s/traffic_pca_pmc.json: %s, rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (committed; not measured in this run)" % (tag, hit[0][0].split("(")[0].replace("void ", ""))
                break
    step_s = dt / t_steps
    QP, DP = 16, 256
    stats_doubles = (QP * QP + DP * QP + DP + QP + 4 + 7) // 8 * 8         # pyvb_amd/csrc/pca.h: pca_stats_layout
    coll_bytes = 8 * (stats_doubles + 2 * (QP + DP))
    collective = "none (1 GPU)" if world == 1 else \
        ("%s all-reduce, 3 per iteration: statistics %d B + 2 x [sum z | delta sum x] %d B = %d B per iteration"
         % ("rccl" if transport == "rccl" else "host (over TCP: DEGRADED)", 8 * stats_doubles, 8 * (QP + DP), coll_bytes))
    return {"workload": "VB-PCA N=%d d=%d q=%d, 10%% missing (%s), rows sharded over %d GPU(s)"
                        % (N, d, q, "BASELINE configs[4]" if N == 1000000 else "not a BASELINE configuration", world),
            "metric": "VB-PCA iterations/sec", "value": t_steps / dt, "unit": "VB iterations/s", "n_gpus": world, "scaling": "strong",
            "steps": t_steps, "warmup": t_warm, "ms_per_step": step_s * 1e3, "dtype": "f64", "rel_err_vs_numpy": parity,
            "parity_checked_on": "a %d-row copy of the problem sharded the same way, %d iterations, same kernels and collectives" % (n_s, warmup + steps),
            "elbo_total": float(elbo.sum()), "collective": collective, "collective_bytes_per_step": 0 if world == 1 else coll_bytes,
            "degraded": world > 1 and transport != "rccl",
            "roofline": {"bound": "hbm", "achieved": alg / step_s / 1e9, "peak": HBM_PEAK_GBS * world, "unit": "GB/s",
                         "frac": alg / step_s / 1e9 / (HBM_PEAK_GBS * world), "algorithmic_bytes": alg, "algorithmic_bytes_two_sweeps": alg2,
                         "traffic": traffic, "traffic_source": tsrc,
                         "kernel": "whole iteration (the sweep k_pca_pairs + reductions + small kernels)",
                         # the sweep is as much a matrix-core kernel as a streaming one: per entry of X three 16-deep products of the
                         # algorithm (z = Gz x, the prediction W z, the statistic x z^T: 6 q flops) and a fourth that recomputes the
                         # entries the sweep before did not store; at fp64 the machine balance is 9.8 flop/byte, this is 9.6 / 12.8
                         "mfma": {"algorithmic_flops": 6.0 * q * N * d, "executed_flops": 8.0 * q * N * d, "peak": FP64_MFMA_PEAK_TFLOPS * world, "unit": "TFLOP/s",
                                  "achieved": 6.0 * q * N * d / step_s / 1e12, "frac": 6.0 * q * N * d / step_s / 1e12 / (FP64_MFMA_PEAK_TFLOPS * world),
                                  "executed_frac": 8.0 * q * N * d / step_s / 1e12 / (FP64_MFMA_PEAK_TFLOPS * world)}},
            "cpu_baseline": cpu}


def d128_workload(steps, warmup, with_cpu):
    """The headline's loop in the second shape class of the fused kernels (64 < max(D, K) <= 128; the reference has no size limit,
    gaussian.py:43-46), appended to the headline line as workloads.lds_d128: T = 10^4, D = K = 128, 1024 replicates, measured like
    the headline (inputs resident before the timed region, `steps` iterations after `warmup`).  Parity and CPU baseline on a
    short copy of the problem (T = 40, two replicates, three iterations through the same kernels and through the oracle): an
    oracle iteration at T = 10^4 takes 5 s per replicate at this size."""
    from pyvb_amd.lds import LDSBatch
    T, D, K, N = 10000, 128, 128, 1024
    Y, st0, pri = make_inputs(T, D, K, N, seed=777)
    b = LDSBatch.from_problem(Y, st0, pri)
    del Y
    b.iterate(warmup); b.sync()
    t0 = time.perf_counter()
    b.iterate(steps); b.sync()
    dt = time.perf_counter() - t0
    elbo = b.elbo()
    b.close()
    Ts, Ns, its = 40, 2, 3
    Ys, ss, ps = make_inputs(Ts, D, K, Ns, seed=778)
    sb = LDSBatch.from_problem(Ys, ss, ps); sb.iterate(its); gX = sb.get_state(("X",))["X"]; ge = sb.elbo(); sb.close()
    parity, cpu = None, None
    if with_cpu:
        from oracle import lds_closed_form as O        # checker and CPU baseline only
        st = O.expand_state(ss, ps, Ts)
        t1 = time.perf_counter()
        for _ in range(its):
            ref = O.iterate(st, ps, Ys)
        t_short = (time.perf_counter() - t1) / its / Ns        # seconds per replicate and iteration at T = Ts
        parity = max(float(np.abs(gX - st["X"]).max() / np.abs(st["X"]).max()), float(np.max(np.abs(ge.sum(1) - ref.sum(1)) / np.abs(ref.sum(1)))))
        # the oracle's iteration is a + b T (the parameter updates do not depend on T, and dominate at this size): a second,
        # longer sample fixes b; the baseline is the model at T = 10^4
        Tl = 400
        Yl, sl, pl = make_inputs(Tl, D, K, 1, seed=779)
        stl = O.expand_state(sl, pl, Tl)
        t1 = time.perf_counter()
        O.iterate(stl, pl, Yl)
        t_long = time.perf_counter() - t1
        bT = max(t_long - t_short, 0.0) / (Tl - Ts)
        t_full = t_short + bT * (T - Ts)
        cpu = {"value": 1.0 / (t_full * N), "unit": "VB iterations/s per 1024 replicates (model a + b T fitted to two samples, linear in replicates)",
               "cores": os.cpu_count(), "kind": "port",
               "sample": "oracle/lds_closed_form.py: %.2f s per replicate-iteration at T=%d (%d replicates x %d iterations), %.2f s at T=%d (1 x 1) -> %.2f s at T=%d" % (t_short, Ts, Ns, its, t_long, Tl, t_full, T)}
    nt = float(N) * T
    alg = nt * (8 * D * D + 4 * D * K) + nt * (4 * D * D + 2 * D * K + 2 * K)      # two sweeps + all statistics, as the headline's roofline.iteration
    step_s = dt / steps
    # HBM bytes of an iteration: the PMC passes over this workload committed with the round's profiles (NOT measured in this run)
    traffic, tsrc, tparts = None, None, None
    tpath = os.path.join(REPO, "profiles", "r04", "traffic_d128_pmc.json")
    if os.path.exists(tpath):
        tj = json.load(open(tpath))
        tparts = {k.split("(")[0].replace("void ", ""): v["hbm_bytes_per_launch"] for k, v in tj.items()}
        traffic = float(sum(tparts.values()))
        tsrc = "profiles/r04/traffic_d128_pmc.json: one launch each of k_gy_big, k_sweep_big<3,2>, k_sweep_big<2,2>, k_stats_big, k_prep_big, k_cols_big (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes, committed; not measured in this run)"
    return {"workload": "LDS T=%d D=%d K=%d, %d replicates (second shape class of the fused kernels)" % (T, D, K, N),
            "metric": "VB iterations/sec per 1024 replicates", "value": steps / dt, "unit": "VB iterations/s", "steps": steps, "warmup": warmup,
            "ms_per_step": step_s * 1e3, "dtype": "f64", "rel_err_vs_numpy": parity,
            "parity_checked_on": "a T=%d, %d-replicate problem of the same shape class, %d iterations, same kernels" % (Ts, Ns, its),
            "elbo_total": float(elbo.sum()),
            "roofline": {"bound": "mfma", "achieved": alg / step_s / 1e12, "peak": FP64_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": alg / step_s / 1e12 / FP64_MFMA_PEAK_TFLOPS, "algorithmic_flops": alg, "traffic": traffic, "traffic_source": tsrc,
                         "traffic_by_kernel": tparts, "algorithmic_bytes": nt * 16 * (K + 2 * D),
                         "hbm_GBs_at_measured_traffic": (traffic / step_s / 1e9) if traffic else None,
                         "kernel": "whole iteration (k_gy_big, k_sweep_big, k_stats_big, k_prep_big, k_cols_big)"},
            "cpu_baseline": cpu}


def launch_ranks(n, argv):
    """`python bench.py --gpus N` without a launcher: this process -- which has not touched the GPU (pyvb_amd is imported
    only further down, in the ranks) -- starts N fresh child processes of this script, one per GPU, with the environment a
    launcher would give them (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT), relays rank 0's JSON line and
    returns non-zero if any rank fails or the line does not say n_gpus == N.  Children are started with subprocess, never
    by replacing this process."""
    import socket
    import subprocess
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), PYVB_BENCH_CHILD="1")
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + list(argv), env=env,
                                      stdout=(subprocess.PIPE if r == 0 else subprocess.DEVNULL), text=True))
    import threading
    out0 = []
    reader = threading.Thread(target=lambda: out0.extend(procs[0].stdout.readlines()))
    reader.start()
    rc = 0
    live = set(range(n))
    kill_at = None              # after one rank has failed: when the others, already asked to terminate, are killed
    while live:
        for r in sorted(live):
            code = procs[r].poll()
            if code is None:
                continue
            live.discard(r)
            if code != 0 and rc == 0:
                rc = code if code > 0 else 1
                sys.stderr.write("bench.py: rank %d exited with status %d; stopping the other ranks\n" % (r, code))
                for o in live:
                    procs[o].terminate()        # exactly the children started above
                kill_at = time.time() + 15.0
        if kill_at is not None and live and time.time() > kill_at:
            for o in live:                      # a rank stuck in a GPU call ignores SIGTERM: do not wait for it for ever
                procs[o].kill()
            kill_at = time.time() + 15.0
        time.sleep(0.05)
    reader.join()
    lines = [l for l in out0 if l.startswith("{")]
    for l in out0:
        if not l.startswith("{"):
            sys.stderr.write(l)
    if rc:
        return rc
    if len(lines) != 1:
        sys.stderr.write("bench.py: expected one JSON line from rank 0, got %d\n" % len(lines))
        return 1
    d = json.loads(lines[0])
    if d.get("n_gpus") != n:
        sys.stderr.write("bench.py: asked for %d GPUs, the line reports n_gpus = %r\n" % (n, d.get("n_gpus")))
        return 1
    sys.stdout.write(lines[0])
    sys.stdout.flush()
    return 0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--replicates", type=int, default=1024, help="replicates per GPU")
    ap.add_argument("--T", type=int, default=10000)
    ap.add_argument("--D", type=int, default=64)
    ap.add_argument("--K", type=int, default=64)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-workloads", action="store_true", help="skip the second workload (VB-PCA, BASELINE configs[4]) after the headline")
    ap.add_argument("--workloads", default="auto", help="comma list of pca_config5, lds_d128; auto: both after the headline shape on one "
                                                        "GPU, pca_config5 (rows sharded over the ranks) with several")
    ap.add_argument("--pca-rows", type=int, default=1000000, help="rows of the VB-PCA workload (BASELINE configs[4]: 10^6)")
    ap.add_argument("--parity-replicates", type=int, default=4, help="replicates of the timed batch re-run in the oracle")
    ap.add_argument("--rccl-single", action="store_true",
                    help="with one process: still create a (one-rank) RCCL communicator and all-reduce the lower bound through it, "
                         "i.e. run the multi-GPU code path as far as one GPU allows")
    ap.add_argument("--allow-host-fallback", "--allow-gloo-fallback", dest="allow_host_fallback", action="store_true",
                    help="if the RCCL communicator cannot be created, reduce the lower bound through host memory over the rendezvous "
                         "sockets (line marked degraded) instead of failing")
    args = ap.parse_args()

    if args.gpus < 1:
        raise SystemExit("--gpus must be at least 1")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # no launcher: start the N ranks ourselves (nothing has touched the GPU in this process yet)
        raise SystemExit(launch_ranks(args.gpus, sys.argv[1:]))
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        # one rank asked to stand for N GPUs (or the reverse) would put a wrong n_gpus beside a real number: refuse
        raise SystemExit("--gpus %d but WORLD_SIZE=%d: start one rank per GPU (python bench.py --gpus N does so itself)" % (args.gpus, world))

    from pyvb_amd.lds import LDSBatch      # raises if libpyvb_hip.so is missing: no fallback
    from pyvb_amd import dist as pdist

    comm = pdist.init(world, rank)         # TCP rendezvous (standard library) for barriers / max-reduce; RCCL inside the library
    T, D, K, N = args.T, args.D, args.K, args.replicates
    Y, st0, pri = make_inputs(T, D, K, N, seed=20240 + 1000 * rank)
    from pyvb_amd import _capi
    ndev = _capi.ctypes.c_int(0)
    _capi.check(_capi.lib.pyvb_device_count(_capi.ctypes.byref(ndev)))
    device = local_rank % max(ndev.value, 1)          # a launcher may already have narrowed the visible devices
    b = LDSBatch.from_problem(Y, st0, pri, device=device)
    n_par = max(1, min(args.parity_replicates, N))
    sample = (Y[:n_par].copy(), {k: v[:n_par].copy() for k, v in st0.items()})
    del Y
    collective = "none (1 GPU)"
    degraded = False
    use_rccl = False
    if world > 1 or args.rccl_single:
        # the one collective of the data path: all-reduce of the 6 lower-bound parts, over RCCL inside the library
        uid, ok, why = None, 0.0, ""
        # one node: RCCL's bootstrap (sockets, not the data path) over loopback, like the rendezvous above --
        # the container's hostname / outward interface may not be usable
        if os.environ.get("MASTER_ADDR", "127.0.0.1") in ("127.0.0.1", "localhost"):
            os.environ.setdefault("NCCL_SOCKET_IFNAME", "lo")
        if rank == 0:
            try:
                uid = LDSBatch.comm_unique_id()
            except Exception as e:
                why = "rank 0: no RCCL unique id (%s)" % (e,)
        uid = comm.broadcast_bytes(uid)
        if uid is not None:
            try:
                b.comm_init(uid, rank, world)
                ok = 1.0
            except Exception as e:
                why = "rank %d: RCCL communicator failed (%s)" % (rank, e)
        ok = -comm.max_float(-ok)       # min over ranks
        use_rccl = ok == 1.0
        if not use_rccl:
            # a number taken with the lower bound going through host memory is not this design's data path:
            # fail unless the caller asked for the fallback, and mark the line if so
            if why:
                sys.stderr.write(why + "\n")
            if not args.allow_host_fallback:
                b.close()
                comm.close()
                raise SystemExit("RCCL communicator could not be created on every rank (see stderr); "
                                 "re-run with --allow-host-fallback for a degraded measurement (lower bound reduced through host memory)")
            degraded = True
            b.comm_init_host(comm, rank, world)     # the library's own collective calls, carried by the rendezvous sockets
        collective = ("rccl allreduce(6 x f64) per step, %d rank(s)" % world) if use_rccl else "host allreduce(6 x f64) per step over TCP (RCCL init failed: DEGRADED)"

    def step():
        # one variational iteration of every replicate on this GPU: forward sweep, backward sweep, A, C, Q, R on the handle's
        # main stream; the lower bound, its reduction over the replicates and the RCCL all-reduce over the ranks on the
        # side stream (they feed nothing in the next iteration).  Nothing synchronises with the host inside a step; the
        # per-step lower bounds are read from the library's history ring after the timed region.
        b.iterate(1)

    for _ in range(args.warmup):
        step()
    b.sync()
    b.reset_elbo_history()
    b.timing(True)
    comm.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    b.sync()
    comm.barrier()
    dt = time.perf_counter() - t0
    dt = comm.max_float(dt)
    hist = b.elbo_history(args.steps)
    assert hist.shape[0] == min(args.steps, 4096) and np.all(np.isfinite(hist)), "lower-bound history incomplete"
    elbo = hist[-1]
    kt = b.kernel_times()
    b.timing(False)

    # ---- roofline, per kernel instantiation.  Work per launch (N replicates, T nodes; unit u = 2 D^2 flops per node):
    #   forward sweep   k_sweep<..,1,..>: F mu_{t-1} + B mu_{t+1} + G y_t          executed = algorithmic = T (4 D^2 + 2 D K)
    #   backward sweep  k_sweep<..,2,..>: B mu_{t+1} + c_t (c_t cached by the forward sweep) + the 10 upper tiles of Sxx
    #                   executed = T (2 D^2 + Sxx share); algorithmic (a full sweep)   = T (4 D^2 + 2 D K)
    #   statistics      k_stats<..,false>: Sx1x + Syx                              executed = T (2 D^2 + 2 D K)
    #                   algorithmic (all statistics, a third of them done in the backward launch) = T (4 D^2 + 2 D K + 2 K)
    # `achieved`/`frac` are EXECUTED flops over the mean launch time from HIP events on the handle's stream: they can
    # not exceed the peak.  The contract figure for the whole iteration (SURVEY's algorithmic flops over the wall
    # time of a step, whatever kernel a piece of work is fused into) is roofline.iteration.
    DT = (D + 15) // 16
    sxx_share = (DT * (DT + 1) / 2.0) / (DT * DT)           # symmetric: upper tiles only
    nt = float(N) * T
    if D > 64 or K > 64:
        # the second shape class (pyvb_amd/csrc/k_big.hip): a workgroup per replicate, both dimensions padded to 128.  G y_t is a
        # batched product of its own (k_gy_big), the forward sweep adds the two recurrence products, the backward sweep behind
        # it reads c_t and runs one; the statistics are three full 128-wide products
        P = 128
        work = {
            "sweep_fwd": {"kernel": "k_sweep_big<3, 2>(BigSweepArgs)", "executed_flops": nt * 4 * P * P,
                          "algorithmic_flops": nt * 4 * D * D, "algorithmic_bytes": nt * 8 * (4 * P)},
            "gy": {"kernel": "k_gy_big<true>(BigGyArgs)", "executed_flops": nt * 2 * P * P,
                   "algorithmic_flops": nt * 2 * D * K, "algorithmic_bytes": nt * 8 * (K + P)},
            "sweep_bwd": {"kernel": "k_sweep_big<2, 2>(BigSweepArgs)", "executed_flops": nt * 2 * P * P,
                          "algorithmic_flops": nt * (4 * D * D + 2 * D * K), "algorithmic_bytes": nt * 8 * (2 * P)},
            "stats": {"kernel": "k_stats_big(BigStatsArgs)", "executed_flops": nt * 6 * P * P,
                      "algorithmic_flops": nt * (4 * D * D + 2 * D * K + 2 * K), "algorithmic_bytes": 0.0},
        }
    else:
        work = {
            "sweep_fwd": {"kernel": "k_sweep<%d, %d, %s, 1, false>" % (DT, (K + 15) // 16, "true" if (D % 16 == 0 and K % 16 == 0) else "false"),
                          "executed_flops": nt * (4 * D * D + 2 * D * K), "algorithmic_flops": nt * (4 * D * D + 2 * D * K),
                          "algorithmic_bytes": nt * 8 * (K + 2 * D)},
            "sweep_bwd": {"kernel": "k_sweep<%d, %d, %s, 2, false>" % (DT, (K + 15) // 16, "true" if (D % 16 == 0 and K % 16 == 0) else "false"),
                          "executed_flops": nt * (2 * D * D + sxx_share * 2 * D * D), "algorithmic_flops": nt * (4 * D * D + 2 * D * K),
                          "algorithmic_bytes": nt * 8 * (K + 2 * D)},
            "stats": {"kernel": "k_stats<%d, %d, false>" % (DT, (K + 15) // 16),
                      "executed_flops": nt * (2 * D * D + 2 * D * K), "algorithmic_flops": nt * (4 * D * D + 2 * D * K + 2 * K),
                      "algorithmic_bytes": 0.0},
        }
    # HBM bytes per launch: PMC passes over this workload, committed with the round's profiles (NOT measured in this run)
    traffic, traffic_source = {}, None
    if (N, T, D, K) == (1024, 10000, 64, 64):
        for tag in ("r04", "r03", "r02", "r01"):
            tpath = os.path.join(REPO, "profiles", tag, "traffic_pmc.json")
            if os.path.exists(tpath):
                tj = json.load(open(tpath))
                for key, w in work.items():
                    hit = [v["hbm_bytes_per_launch"] for k, v in tj.items() if w["kernel"].replace(" ", "") in k.replace(" ", "")]
                    if hit:
                        traffic[key] = hit[0]
                traffic_source = "profiles/%s/traffic_pmc.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes, committed; not measured in this run)" % tag
                break
    entries = {}
    for key, w in work.items():
        ms, cnt = kt[key]
        if not cnt:
            continue
        mean_ms = ms / cnt
        ex = w["executed_flops"] / (mean_ms * 1e-3) / 1e12
        e = {"kernel": w["kernel"], "launches": cnt, "mean_launch_ms": mean_ms,
             "executed_flops": w["executed_flops"], "algorithmic_flops": w["algorithmic_flops"],
             "achieved": ex, "frac": ex / FP64_MFMA_PEAK_TFLOPS,
             "algorithmic_TFLOPs": w["algorithmic_flops"] / (mean_ms * 1e-3) / 1e12,
             "traffic": traffic.get(key)}
        if traffic.get(key):
            e["hbm_GBs_measured_traffic"] = traffic[key] / (mean_ms * 1e-3) / 1e9
            e["hbm_frac"] = e["hbm_GBs_measured_traffic"] / HBM_PEAK_GBS
        entries[key] = e
    dom = entries.get("sweep_fwd", {})
    roofline = {"bound": "mfma", "achieved": dom.get("achieved", 0.0), "peak": FP64_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                "frac": dom.get("frac", 0.0), "traffic": dom.get("traffic"), "traffic_source": traffic_source,
                "kernel": dom.get("kernel"), "launches": dom.get("launches"), "mean_launch_ms": dom.get("mean_launch_ms"),
                "algorithmic_flops_per_launch": work["sweep_fwd"]["algorithmic_flops"],
                "algorithmic_bytes_per_launch": work["sweep_fwd"]["algorithmic_bytes"],
                "secondary": {k: v for k, v in entries.items() if k != "sweep_fwd"},
                "power_cap": {"peak_at_capped_clock": FP64_MFMA_PEAK_TFLOPS * CAPPED_CLOCK_RATIO,
                              "frac_of_capped_peak": dom.get("achieved", 0.0) / (FP64_MFMA_PEAK_TFLOPS * CAPPED_CLOCK_RATIO),
                              "source": "profiles/r02/limits.txt (rocm-smi during the sweeps: 1343 W, sclk 2055 MHz; not measured in this run)"}}
    # the whole iteration against the same peak: SURVEY.md section 8(d) work model (sweeps + statistics) over the wall time
    # of a step -- independent of which kernel a piece of the work is fused into -- and the flops actually executed
    iter_flops = float(N) * T * (12 * D * D + 6 * D * K + 2 * K)
    iter_exec = sum(w["executed_flops"] for w in work.values())
    step_s = dt / args.steps
    roofline["iteration"] = {"algorithmic_flops": iter_flops, "achieved": iter_flops / step_s / 1e12,
                             "frac": iter_flops / step_s / 1e12 / FP64_MFMA_PEAK_TFLOPS,
                             "executed_flops_big_kernels": iter_exec, "executed_TFLOPs": iter_exec / step_s / 1e12,
                             "executed_frac": iter_exec / step_s / 1e12 / FP64_MFMA_PEAK_TFLOPS,
                             "algorithmic_bytes": float(N) * 16 * T * (K + 2 * D),
                             "traffic": (sum(traffic.values()) if len(traffic) == len(work) else None)}

    if rank == 0:
        cpu, rel, rel_x = (None, None, None)
        if not args.no_cpu_baseline:
            got = b.elbo()
            gx = b.get_state(("X",))["X"][:n_par].copy()
            cpu, rel, rel_x = cpu_baseline_and_parity(sample, pri, args.warmup + args.steps, got, gx)
        total_rep = N * world
        value = (total_rep / 1024.0) * args.steps / dt
        out = {
            "metric": "VB update iterations/sec (T=10k, D=64, N=1024 LDS); ELBO rel-err vs NumPy",
            "value": value, "unit": "VB iterations/s per 1024 replicates (whole job)",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f64",
            "data": "synthetic LDS (simulated x_t = A x_{t-1} + w, y_t = C x_t + v; up to 128 distinct systems tiled, distinct initial posteriors)",
            "config": {"workload": ("LDS T=%d D=%d K=%d, %d replicates per GPU (BASELINE configs[%d])" % (T, D, K, N, 2 if world == 1 else 3))
                                   if (T, D, K, N) == (10000, 64, 64, 1024) else "LDS T=%d D=%d K=%d, %d replicates per GPU (not a BASELINE configuration)" % (T, D, K, N),
                       "replicates_total": total_rep, "parallelism": "replicates sharded over %d GPU(s)" % world, "collective": collective,
                       "elbo_rel_err_vs_numpy": rel, "state_rel_err_vs_numpy": rel_x,
                       "parity_checked_on": "replicates 0..%d of the timed batch after %d iterations" % (n_par - 1, args.warmup + args.steps),
                       "degraded": degraded, "elbo_total": float(np.sum(elbo)),
                       "elbo_total_first_timed_step": float(hist[0].sum()),
                       "kernel_ms_per_step": {k: v[0] / args.steps for k, v in kt.items() if v[1]}},
            "roofline": roofline,
            "cpu_baseline": cpu,
        }
    else:
        out = None
    b.close()
    # the other GPU-sized configurations of BASELINE.json, after the headline's timed region and with its memory released.
    # VB-PCA (configs[4]) is a job of ALL ranks -- its rows are sharded over them; the 128-wide LDS class runs on one GPU only
    if args.workloads == "auto":
        want = [] if args.no_workloads or (N, T, D, K) != (1024, 10000, 64, 64) else (["pca_config5", "lds_d128"] if world == 1 else ["pca_config5"])
    else:
        want = [w for w in args.workloads.split(",") if w]
    wl = {}
    if "pca_config5" in want:
        transport = "none" if world == 1 else ("rccl" if use_rccl else "host")
        wl["pca_config5"] = pca_workload(args.steps, args.warmup, not args.no_cpu_baseline, comm, rank, world, device, transport, args.pca_rows)
    if "lds_d128" in want and world == 1:
        wl["lds_d128"] = d128_workload(args.steps, args.warmup, not args.no_cpu_baseline)
    if rank == 0:
        if wl:
            out["workloads"] = wl
        print(json.dumps(out), flush=True)
    comm.close()


if __name__ == "__main__":
    main()
