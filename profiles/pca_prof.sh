set -e
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pca_prof
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
python3 $ROOT/profiles/bench_pca.py --steps 20 --warmup 3 --no-cpu-baseline > $OUT/bench.json 2>&1
rocprofv3 --kernel-trace --stats -d $OUT/pca_stats -o p --output-format csv -- python3 $ROOT/profiles/pca_probe.py > $OUT/pca_stats.log 2>&1
cp $OUT/pca_stats/p_kernel_stats.csv $OUT/pca_kernel_stats.csv
python3 $ROOT/profiles/pca_trace_summary.py $OUT/pca_stats/p_kernel_trace.csv > $OUT/pca_iteration_timeline.txt
python3 $ROOT/profiles/fuzz_ops_pca.py > $OUT/fuzz_ops_pca.txt 2>&1 || true
python3 $ROOT/profiles/fuzz_batch_pca.py > $OUT/fuzz_batch_pca.txt 2>&1 || true
