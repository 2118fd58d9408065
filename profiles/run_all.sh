#!/bin/bash
# Everything under profiles/<tag>/ in one go, on a GPU box, from the repo root:
#   bash profiles/run_all.sh r02
# bench line, rocprofv3 kernel stats of the same command, HBM traffic (one --pmc pass per counter),
# clock / MFMA-busy counters, and the fp64 microbenchmarks.  Outputs land in gpurun_out/<tag>/ (scratch);
# copy what is to be kept into profiles/<tag>/.
set -e
TAG=${1:-r01}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp

python3 $ROOT/bench.py > $OUT/bench.json 2> $OUT/bench.err
rocprofv3 --kernel-trace --stats -d $OUT/stats -o k --output-format csv -- python3 $ROOT/bench.py --steps 10 --warmup 2 --no-cpu-baseline > $OUT/stats.log 2>&1
cp $OUT/stats/k_kernel_stats.csv $OUT/bench_kernel_stats.csv

# counters: their own runs, kernel trace only (never together with sys/hip/hsa traces)
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY -d $OUT/pmc -o p --output-format csv -- python3 $ROOT/profiles/headline_probe.py 3 > $OUT/pmc.log 2>&1
python3 $ROOT/profiles/pmc_summary.py $OUT/pmc/p_counter_collection.csv $OUT/pmc/p_kernel_trace.csv > $OUT/pmc_clock_mfma.txt

cd $ROOT && bash profiles/collect_traffic.sh $TAG > $OUT/traffic.log 2>&1
cp $ROOT/gpurun_out/traffic_$TAG/traffic.json $OUT/traffic_pmc.json

for mb in mb_f64 mb_power mb_copy; do if [ -x $ROOT/profiles/microbench/$mb ]; then $ROOT/profiles/microbench/$mb > $OUT/$mb.txt 2>&1; fi; done
ls -l $OUT
