#!/bin/bash
# HBM traffic of the two VB-PCA passes at BASELINE configs[4]'s size, as collect_traffic.sh does it for the LDS kernels:
# one --pmc pass per counter.  Run on the GPU box from the repo root:  bash profiles/collect_traffic_pca.sh r02
set -e
TAG=${1:-r02}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/traffic_pca_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $C -d $OUT/$C -o t --output-format csv -- python3 $ROOT/profiles/pca_probe.py 1000000 256 16 4 > $OUT/$C.log 2>&1
done
python3 - "$OUT" <<'PY'
import csv, sys, json, collections
out = sys.argv[1]
res = collections.defaultdict(dict)
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    rows = list(csv.DictReader(open("%s/%s/t_counter_collection.csv" % (out, c))))
    acc = collections.defaultdict(list)
    for r in rows:
        if r["Counter_Name"] == c:
            acc[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    for k, v in acc.items():
        res[k][c] = sum(v) / len(v)
summary = {}
for k, v in res.items():
    if "k_pca_pass" in k:
        # units: KiB.  gfx950: FETCH_SIZE counts half the bytes of wide (>= 16 B/lane) coalesced reads -> x2 (MI355X_MICROARCH.md,
        # HBM section); both passes read X that way (32 and 16 B per lane); the 8-byte Z and 2-byte mask loads of pass 2 are
        # over-corrected by it, so its read figure is an upper bound.
        rd = 2.0 * v.get("FETCH_SIZE", 0.0) * 1024
        wr = v.get("WRITE_SIZE", 0.0) * 1024
        summary[k] = {"fetch_bytes_corrected": rd, "write_bytes": wr, "hbm_bytes_per_launch": rd + wr,
                      "raw_FETCH_SIZE_KiB": v.get("FETCH_SIZE"), "raw_WRITE_SIZE_KiB": v.get("WRITE_SIZE")}
json.dump(summary, open(out + "/traffic.json", "w"), indent=1)
print(json.dumps(summary, indent=1))
PY
