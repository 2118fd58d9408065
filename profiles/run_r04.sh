#!/bin/bash
# Round 4's measurements on a GPU box, from the repo root:  bash profiles/run_r04.sh A|B|C   (writes gpurun_out/r04/; copy what is
# to be kept into profiles/r04/).  A: the bench line, rocprofv3 kernel statistics and HBM traffic (one --pmc pass per counter)
# of the same command -- headline, VB-PCA and the 128-wide class in one run.  B: the probes.  C: the fuzzers.
set -e
PART=${1:-A}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r04
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
if [ "$PART" = A ]; then
  python3 $ROOT/bench.py > $OUT/bench_final.json 2> $OUT/bench.err
  rocprofv3 --kernel-trace --stats -d $OUT/stats -o k --output-format csv -- python3 $ROOT/bench.py --steps 10 --warmup 2 --no-cpu-baseline > $OUT/stats.log 2>&1
  cp $OUT/stats/k_kernel_stats.csv $OUT/bench_kernel_stats.csv
  for C in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --kernel-trace --pmc $C -d $OUT/tr/$C -o t --output-format csv -- python3 $ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OUT/tr_$C.log 2>&1
  done
  python3 $ROOT/profiles/traffic_summary.py $OUT/tr k_sweep k_stats k_prep k_cols k_gy_big k_pca_pass12 k_pca_pairs k_pca_small k_pca_reduce k_pca_rowvar > $OUT/traffic_pmc.json
  python3 - $OUT <<'PY'
import json, sys
out = sys.argv[1]
t = json.load(open(out + "/traffic_pmc.json"))
json.dump({k: v for k, v in t.items() if "_big" not in k and "k_pca" not in k}, open(out + "/traffic_headline.json", "w"), indent=1)
json.dump({k: v for k, v in t.items() if "k_pca" in k and "k_pca_pass12" not in k}, open(out + "/traffic_pca_pmc.json", "w"), indent=1)    # (k_pca_pass12: the small parity copy's sweep)
json.dump({k: v for k, v in t.items() if "_big" in k}, open(out + "/traffic_d128_pmc.json", "w"), indent=1)
PY
  rm -rf $OUT/stats/*.db $OUT/tr/*/*.db 2>/dev/null || true
  ls -l $OUT
fi
if [ "$PART" = B ]; then
  cd $ROOT
  python3 profiles/pca_probe.py 1000000 256 16 20 > $OUT/pca_probe.txt 2>&1
  python3 profiles/example_probe.py > $OUT/example_probe.txt 2>&1
  python3 profiles/big_single_chain_probe.py > $OUT/big_single_chain.txt 2>&1
  python3 profiles/wishart_probe.py > $OUT/wishart_probe.txt 2>&1 || true
  python3 profiles/small_configs_probe.py > $OUT/small_configs.txt 2>&1 || true
  cd /tmp
  rocprofv3 --kernel-trace --stats -d $OUT/pca_stats -o p --output-format csv -- python3 $ROOT/profiles/pca_probe.py > $OUT/pca_stats.log 2>&1
  cp $OUT/pca_stats/p_kernel_stats.csv $OUT/pca_kernel_stats.csv
  python3 $ROOT/profiles/pca_trace_summary.py $OUT/pca_stats/p_kernel_trace.csv > $OUT/pca_iteration_timeline.txt || true
  rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY -d $OUT/pmc -o p --output-format csv -- python3 $ROOT/profiles/headline_probe.py 3 > $OUT/pmc.log 2>&1
  python3 $ROOT/profiles/pmc_summary.py $OUT/pmc/p_counter_collection.csv $OUT/pmc/p_kernel_trace.csv > $OUT/pmc_clock_mfma.txt || true
  rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY -d $OUT/pmc_pca -o p --output-format csv -- python3 $ROOT/profiles/pca_probe.py 1000000 256 16 5 > $OUT/pmc_pca.log 2>&1
  python3 $ROOT/profiles/pmc_summary.py $OUT/pmc_pca/p_counter_collection.csv $OUT/pmc_pca/p_kernel_trace.csv > $OUT/pmc_clock_mfma_pca.txt || true
  rm -rf $OUT/pca_stats/*.db $OUT/pmc/*.db $OUT/pmc_pca/*.db 2>/dev/null || true
  cd $ROOT
  for s in columns rows pairs; do echo -n "$s: " >> $OUT/pca_sweeps.txt; PYVB_PCA_SWEEP=$s python3 profiles/bench_pca.py --no-cpu-baseline --steps 100 | grep -o '"ms_per_step": [0-9.]*' >> $OUT/pca_sweeps.txt; done
  (PYVB_PCA_WRITEBACK=1 python3 profiles/bench_pca.py --no-cpu-baseline --steps 100 | grep -o '"ms_per_step": [0-9.]*' | sed 's/^/write-back: /') >> $OUT/pca_sweeps.txt
  $ROOT/build/hbm_read > $OUT/hbm_read_microbench.txt 2>&1 || true
fi
if [ "$PART" = C ]; then
  cd $ROOT
  mkdir -p $OUT/fuzz
  for f in fuzz_batch fuzz_batch_pca fuzz_generic fuzz_more soak; do
    timeout -k 10 180 python3 profiles/$f.py > $OUT/fuzz/$f.txt 2>&1; echo "$f rc=$? : $(tail -1 $OUT/fuzz/$f.txt)"
  done
  for f in fuzz_batch_pca fuzz_ops_pca; do          # the VB-PCA fuzzers again through the pair-owning sweep (small problems default to the column-owning one)
    PYVB_PCA_SWEEP=pairs timeout -k 10 180 python3 profiles/$f.py > $OUT/fuzz/${f}_pairs.txt 2>&1; echo "$f (pairs) rc=$? : $(tail -1 $OUT/fuzz/${f}_pairs.txt)"
  done
  timeout -k 10 200 python3 profiles/fuzz_shapes.py 30 3 > $OUT/fuzz/fuzz_shapes.txt 2>&1; echo "fuzz_shapes rc=$? : $(tail -1 $OUT/fuzz/fuzz_shapes.txt)"
fi
