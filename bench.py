#!/usr/bin/env python3
"""bench.py -- VB update iterations/sec of the LDS path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W]
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

FP64_MFMA_PEAK_TFLOPS = 78.6      # MI355X vendor figure for FP64 matrix (= FP64 vector); profiles/r01/microbench_f64.txt measures 73.8 sustained
HBM_PEAK_GBS = 8000.0


def make_inputs(T, D, K, N, seed):
    """N replicates: up to 128 distinct simulated systems tiled to N, every replicate with its own
    initial parameter posterior (so all N chains compute different things)."""
    from pyvb_amd import synth
    base = min(N, 128)
    Y, st0, pri = synth.make_problem(T, D, K, base, seed)
    rep = (N + base - 1) // base
    if rep > 1:
        Y = np.concatenate([Y] * rep)[:N]
        big = synth.initial_state(1, D, K, N, seed + 1)          # cheap: T = 1
        st = {k: np.concatenate([v] * rep)[:N] for k, v in st0.items()}
        for k in ("A_mean", "C_mean", "A_colvar", "C_colvar", "Q_b", "R_b"):
            st[k] = big[k]
        st0 = st
    return Y, st0, pri


def cpu_baseline_and_parity(T, D, K, device):
    """Oracle (numpy port of the reference's loop, oracle/lds_closed_form.py) timed on this host on
    a bounded sample of the same workload, and the ELBO of the HIP path checked against it."""
    from oracle import lds_closed_form as O
    from pyvb_amd import synth
    from pyvb_amd.lds import LDSBatch
    n_s, iters = 4, 8        # about 12 s of numpy on the GPU box's host cores
    Y, st0, pri = synth.make_problem(T, D, K, n_s, seed=99)
    st = O.expand_state(st0, pri, T)
    t0 = time.perf_counter()
    for _ in range(iters):
        ref = O.iterate(st, pri, Y)
    cpu_s = time.perf_counter() - t0
    b = LDSBatch.from_problem(Y, st0, pri, device=device)
    b.iterate(iters)
    got = b.elbo()
    b.close()
    rel = float(np.max(np.abs(got.sum(1) - ref.sum(1)) / np.abs(ref.sum(1))))
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
        "sample": "%d replicates x %d iterations at T=%d D=%d K=%d (%.1f s), scaled by replicates/1024" % (n_s, iters, T, D, K, cpu_s),
    }
    return base, rel


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
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))

    from pyvb_amd.lds import LDSBatch      # raises if libpyvb_hip.so is missing: no fallback
    from pyvb_amd import dist as pdist

    comm = pdist.init(world, rank)         # gloo rendezvous for barriers / max-reduce; RCCL inside the library
    T, D, K, N = args.T, args.D, args.K, args.replicates
    Y, st0, pri = make_inputs(T, D, K, N, seed=20240 + 1000 * rank)
    from pyvb_amd import _capi
    ndev = _capi.ctypes.c_int(0)
    _capi.check(_capi.lib.pyvb_device_count(_capi.ctypes.byref(ndev)))
    device = local_rank % max(ndev.value, 1)          # a launcher may already have narrowed the visible devices
    b = LDSBatch.from_problem(Y, st0, pri, device=device)
    del Y
    collective = "none (1 GPU)"
    if world > 1:
        # the one collective of the data path: all-reduce of the 6 lower-bound parts, over RCCL inside the library
        uid, ok = None, 0.0
        # one node: RCCL's bootstrap (sockets, not the data path) over loopback, like the rendezvous above --
        # the container's hostname / outward interface may not be usable
        if os.environ.get("MASTER_ADDR", "127.0.0.1") in ("127.0.0.1", "localhost"):
            os.environ.setdefault("NCCL_SOCKET_IFNAME", "lo")
        if rank == 0:
            try:
                uid = LDSBatch.comm_unique_id()
            except Exception as e:
                sys.stderr.write("rank 0: no RCCL unique id (%s)\n" % (e,))
        uid = comm.broadcast_bytes(uid)
        if uid is not None:
            try:
                b.comm_init(uid, rank, world)
                ok = 1.0
            except Exception as e:      # keep the run alive: the 48-byte reduction then goes over gloo, and the line says so
                sys.stderr.write("rank %d: RCCL communicator failed (%s); reducing the lower bound over gloo\n" % (rank, e))
        ok = -comm.max_float(-ok)       # min over ranks
        collective = "rccl allreduce(6 x f64) per step" if ok == 1.0 else "gloo allreduce(6 x f64) per step (RCCL init failed)"
        use_rccl = ok == 1.0
    else:
        use_rccl = False

    def step():
        b.iterate(1)
        if world > 1 and not use_rccl:
            return comm.allreduce_sum(b.elbo().sum(0))
        return b.elbo_total()           # device reduction over replicates (+ one ncclAllReduce of 6 doubles)

    for _ in range(args.warmup):
        step()
    b.sync()
    b.timing(True)
    comm.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        elbo = step()
    b.sync()
    comm.barrier()
    dt = time.perf_counter() - t0
    dt = comm.max_float(dt)
    kt = b.kernel_times()
    b.timing(False)

    # roofline of the dominant kernel (the sweep): algorithmic fp64 flops per launch / mean launch time
    sweep_ms, sweep_n = kt["sweep"]
    flops_per_launch = float(N) * T * (4 * D * D + 2 * D * K)            # one sweep of N replicates
    bytes_per_launch = float(N) * 8 * T * (K + 2 * D)
    mean_ms = sweep_ms / max(sweep_n, 1)
    achieved = flops_per_launch / (mean_ms * 1e-3) / 1e12 if sweep_n else 0.0
    # HBM bytes per k_sweep launch from the PMC passes committed with this round's profiles
    # (profiles/collect_traffic.sh: FETCH_SIZE x2 per the gfx950 correction, + WRITE_SIZE; mean of the two directions)
    traffic = None
    tpath = os.path.join(REPO, "profiles", "r01", "traffic_pmc.json")
    if os.path.exists(tpath) and (N, T, D, K) == (1024, 10000, 64, 64):
        tj = json.load(open(tpath))
        vals = [v["hbm_bytes_per_launch"] for k, v in tj.items() if "k_sweep" in k]
        traffic = sum(vals) / len(vals) if vals else None
    roofline = {"bound": "mfma", "achieved": achieved, "peak": FP64_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                "frac": achieved / FP64_MFMA_PEAK_TFLOPS, "traffic": traffic,
                "algorithmic_bytes_per_launch": bytes_per_launch,
                "kernel": "k_sweep", "launches": sweep_n, "mean_launch_ms": mean_ms,
                # the backward launch reuses c_t = F mu_{t-1} + G y_t of the forward one and executes a third of
                # its algorithmic FLOPs: executed / algorithmic over a forward+backward pair
                "executed_over_algorithmic_flops": (3.0 + 1.0) / 6.0,
                # ... and uses its idle matrix pipe for Sxx = sum_t mu_t mu_t^T, 10 of the 42 statistics tiles, which are
                # NOT counted in `achieved` (algorithmic FLOPs of the fused part per backward launch, for reference):
                "fused_statistics_flops_per_backward_launch": float(N) * T * 2 * D * D,
                "hbm_algorithmic_GBs": bytes_per_launch / (mean_ms * 1e-3) / 1e9 if sweep_n else 0.0}
    # the whole iteration against the same peak: SURVEY.md section 8(d) work model (sweeps + statistics) over the wall time
    # of a step -- independent of which kernel a piece of the work is fused into
    iter_flops = float(N) * T * (12 * D * D + 6 * D * K + 2 * K)
    roofline["iteration"] = {"algorithmic_flops": iter_flops, "achieved": iter_flops / (dt / args.steps) / 1e12,
                             "frac": iter_flops / (dt / args.steps) / 1e12 / FP64_MFMA_PEAK_TFLOPS,
                             "algorithmic_bytes": float(N) * 16 * T * (K + 2 * D)}

    if rank == 0:
        cpu, rel = (None, None)
        if not args.no_cpu_baseline:
            cpu, rel = cpu_baseline_and_parity(T, D, K, local_rank)
        total_rep = N * world
        value = (total_rep / 1024.0) * args.steps / dt
        out = {
            "metric": "VB update iterations/sec (T=10k, D=64, N=1024 LDS); ELBO rel-err vs NumPy",
            "value": value, "unit": "VB iterations/s per 1024 replicates (whole job)",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f64",
            "data": "synthetic LDS (simulated x_t = A x_{t-1} + w, y_t = C x_t + v; up to 128 distinct systems tiled, distinct initial posteriors)",
            "config": {"workload": "LDS T=%d D=%d K=%d, %d replicates per GPU (BASELINE configs[%d])" % (T, D, K, N, 2 if world == 1 else 3),
                       "replicates_total": total_rep, "parallelism": "replicates sharded over %d GPU(s)" % world, "collective": collective,
                       "elbo_rel_err_vs_numpy": rel, "elbo_total": float(np.sum(elbo)),
                       "kernel_ms_per_step": {k: v[0] / args.steps for k, v in kt.items() if v[1]}},
            "roofline": roofline,
            "cpu_baseline": cpu,
        }
        print(json.dumps(out), flush=True)
    b.close()
    comm.close()


if __name__ == "__main__":
    main()
