#!/bin/bash
# Wishart noise at the headline shape: time per iteration, rocprofv3 kernel stats, and HBM traffic per kernel (one --pmc pass
# per counter, kernel trace only).  On the GPU box, from the repo root:  bash profiles/profile_wishart.sh r03
set -e
TAG=${1:-r03}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/wishart_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
python3 $ROOT/profiles/wishart_probe.py > $OUT/wishart_probe.txt 2>&1
rocprofv3 --kernel-trace --stats -d $OUT/stats -o w --output-format csv -- python3 $ROOT/profiles/wishart_probe.py > $OUT/stats.log 2>&1
cp $OUT/stats/w_kernel_stats.csv $OUT/wishart_kernel_stats.csv
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $C -d $OUT/$C -o t --output-format csv -- python3 $ROOT/profiles/wishart_probe.py 1024 3 > $OUT/$C.log 2>&1
done
python3 $ROOT/profiles/traffic_summary.py $OUT k_colcov k_colmean k_wresid k_dense_pre k_wexpect k_cols_wishart k_prep k_elbo_dense > $OUT/wishart_traffic_pmc.json
cat $OUT/wishart_probe.txt
head -12 $OUT/wishart_kernel_stats.csv
cat $OUT/wishart_traffic_pmc.json
