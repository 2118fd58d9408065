#!/bin/bash
# The differential fuzzers and the soak on a GPU box, from the repo root:  bash profiles/fuzz_all.sh  (writes gpurun_out/fuzz_all/)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/fuzz_all
mkdir -p $OUT
cd $ROOT
for f in fuzz_ops fuzz_ops_pca fuzz_batch fuzz_batch_pca fuzz_generic fuzz_more soak; do
  timeout -k 10 500 python3 profiles/$f.py > $OUT/$f.txt 2>&1; echo "$f rc=$? : $(tail -1 $OUT/$f.txt)"
done
timeout -k 10 500 python3 profiles/fuzz_shapes.py 30 3 > $OUT/fuzz_shapes.txt 2>&1; echo "fuzz_shapes rc=$? : $(tail -1 $OUT/fuzz_shapes.txt)"
