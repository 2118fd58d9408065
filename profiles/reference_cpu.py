#!/usr/bin/env python3
"""Times the REFERENCE's own NumPy path (jameshensman/pyvb, through the lib2to3-translated scratch copy that
tests/golden/make_golden.py sets up) on the LDS example's loop at the headline's D and K and a short chain, in the build
container, and extrapolates to the headline workload (cost is linear in T -- BASELINE.md section 2 -- and replicates are
independent).  Writes profiles/<tag>/reference_cpu.json, which bench.py reports as cpu_baseline.reference_extrapolated.
The reference cannot travel to the GPU box (Python 2 source, never shipped), so this is the only place it can be timed.

    python profiles/reference_cpu.py [tag] [T_measured]
"""
import datetime
import json
import os
import platform
import sys
import time
import warnings

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests", "golden"))
import make_golden as MG  # noqa: E402
from pyvb_amd import synth  # noqa: E402

tag = sys.argv[1] if len(sys.argv) > 1 else "r02"
Tm = int(sys.argv[2]) if len(sys.argv) > 2 else 4
D = K = 64
warnings.simplefilter("ignore")
ref = MG.load_reference()
Y, st0, pri = synth.make_problem(Tm, D, K, 1, 4242)
g = MG.build_graph(ref.nodes, Y[0], pri, st0)
Xs = g["Xs"]


def iteration():
    [x.update() for x in Xs]                    # examples/Linear_Dynamic_System.py:69-77
    Xs.reverse(); [x.update() for x in Xs]; Xs.reverse()
    [a.update() for a in g["As"]]
    [c.update() for c in g["Cs"]]
    g["Q"].update(); g["R"].update()
    return sum(float(n.log_lower_bound()) for grp in (Xs, g["Ys"], g["As"], g["Cs"], [g["Q"], g["R"]]) for n in grp)   # network.py:49


t0 = time.perf_counter()
iteration()
t1 = time.perf_counter()
iteration()
t2 = time.perf_counter()
sec = min(t1 - t0, t2 - t1)
try:
    from threadpoolctl import threadpool_info
    threads = max([p.get("num_threads", 1) for p in threadpool_info()] or [1])
except Exception:
    threads = os.cpu_count()
T_head, N_head = 10000, 1024
per_iter_head = sec * (T_head / Tm) * N_head
out = {
    "value": 1.0 / per_iter_head, "unit": "VB iterations/s per 1024 replicates",
    "kind": "reference (jameshensman/pyvb NumPy path), extrapolated",
    "measured": {"seconds_per_iteration": sec, "T": Tm, "D": D, "K": K, "replicates": 1, "iterations_timed": 2},
    "extrapolation": "x (10000 / %d) in T (cost linear in T: BASELINE.md section 2) x 1024 replicates (independent)" % Tm,
    "seconds_per_headline_iteration": per_iter_head,
    "host": "%s, %d logical CPUs, BLAS threads %d, numpy %s" % (platform.processor() or platform.machine(), os.cpu_count(), threads, np.__version__),
    "where": "build container (the reference's source never travels to the GPU box)",
    "date": datetime.date.today().isoformat(),
    "script": "profiles/reference_cpu.py",
}
os.makedirs(os.path.join(REPO, "profiles", tag), exist_ok=True)
json.dump(out, open(os.path.join(REPO, "profiles", tag, "reference_cpu.json"), "w"), indent=1)
print(json.dumps(out, indent=1))
