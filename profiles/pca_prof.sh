# VB-PCA and generic-path artefacts of a round, on a GPU box from the repo root:  bash profiles/pca_prof.sh   (writes gpurun_out/pca_prof/)
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
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $C -d $OUT/pca_$C -o t --output-format csv -- python3 $ROOT/profiles/pca_probe.py 1000000 256 16 3 > $OUT/pca_$C.log 2>&1
  mkdir -p $OUT/pcatr/$C && cp $OUT/pca_$C/t_counter_collection.csv $OUT/pcatr/$C/
done
python3 $ROOT/profiles/traffic_summary.py $OUT/pcatr k_pca_pass12 k_pca_pass1 k_pca_pass2 k_pca_small k_pca_reduce > $OUT/traffic_pca_pmc.json
python3 $ROOT/profiles/generic_probe.py > $OUT/generic_probe.txt 2>&1
PYVB_TAPE_STATS=1 python3 $ROOT/profiles/generic_lds_probe.py 2>&1 | grep "19550\|ms per\|queued run" > $OUT/generic_lds_probe.txt
python3 $ROOT/profiles/tape_record_cost.py > $OUT/tape_record_cost.txt 2>&1
python3 $ROOT/profiles/fuzz_generic.py > $OUT/fuzz_generic.txt 2>&1 || true
