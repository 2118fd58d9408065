#!/bin/bash
# The artefacts besides the headline's (profiles/run_all.sh), on a GPU box from the repo root:  bash profiles/run_extra.sh r03
#   Wishart iteration (kernel stats + PMC traffic), the 128-wide shape class (bench line + kernel stats), VB-PCA (kernel stats +
#   PMC traffic + the per-launch timeline of an iteration), the generic path's probe, SQ counters of the small headline kernels.
set -e
TAG=${1:-r03}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/extra_$TAG
mkdir -p $OUT
cd $ROOT
bash profiles/profile_wishart.sh $TAG > $OUT/wishart.log 2>&1
cp $ROOT/gpurun_out/wishart_$TAG/wishart_probe.txt $ROOT/gpurun_out/wishart_$TAG/wishart_kernel_stats.csv $ROOT/gpurun_out/wishart_$TAG/wishart_traffic_pmc.json $OUT/
cd /tmp && export TMPDIR=/tmp
python3 $ROOT/bench.py --D 128 --K 128 --steps 10 --warmup 2 --parity-replicates 1 > $OUT/bench_d128.json 2> $OUT/bench_d128.err
rocprofv3 --kernel-trace --stats -d $OUT/d128_stats -o b --output-format csv -- python3 $ROOT/bench.py --D 128 --K 128 --steps 5 --warmup 1 --no-cpu-baseline > $OUT/d128_stats.log 2>&1
cp $OUT/d128_stats/b_kernel_stats.csv $OUT/d128_kernel_stats.csv
rocprofv3 --kernel-trace --stats -d $OUT/pca_stats -o p --output-format csv -- python3 $ROOT/profiles/pca_probe.py > $OUT/pca_stats.log 2>&1
cp $OUT/pca_stats/p_kernel_stats.csv $OUT/pca_kernel_stats.csv
python3 $ROOT/profiles/pca_trace_summary.py $OUT/pca_stats/p_kernel_trace.csv > $OUT/pca_iteration_timeline.txt
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $C -d $OUT/pca_$C -o t --output-format csv -- python3 $ROOT/profiles/pca_probe.py 1000000 256 16 3 > $OUT/pca_$C.log 2>&1
  mkdir -p $OUT/pcatr/$C && cp $OUT/pca_$C/t_counter_collection.csv $OUT/pcatr/$C/
done
python3 $ROOT/profiles/traffic_summary.py $OUT/pcatr k_pca_pass12 k_pca_pass1 k_pca_pass2 k_pca_small k_pca_reduce > $OUT/traffic_pca_pmc.json
python3 $ROOT/profiles/generic_probe.py > $OUT/generic_probe.txt 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS -d $OUT/small_pmc -o s --output-format csv -- python3 $ROOT/profiles/headline_probe.py 3 > $OUT/small_pmc.log 2>&1
python3 $ROOT/profiles/pmc_summary.py $OUT/small_pmc/s_counter_collection.csv $OUT/small_pmc/s_kernel_trace.csv > $OUT/small_kernels_sq.txt
ls -l $OUT
