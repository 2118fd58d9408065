ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/fuzz_long2
mkdir -p $OUT
cd $ROOT
for s in 21 22 23; do timeout -k 10 200 python3 profiles/fuzz_ops.py 120 $s > $OUT/fuzz_ops_s$s.txt 2>&1; echo "fuzz_ops s$s rc=$? : $(tail -1 $OUT/fuzz_ops_s$s.txt)"; done
for s in 31 32 33; do timeout -k 10 200 python3 profiles/fuzz_ops_pca.py 200 $s > $OUT/fuzz_ops_pca_s$s.txt 2>&1; echo "fuzz_ops_pca s$s rc=$? : $(tail -1 $OUT/fuzz_ops_pca_s$s.txt)"; done
for s in 41 42; do timeout -k 10 200 python3 profiles/fuzz_batch.py 100 $s > $OUT/fuzz_batch_s$s.txt 2>&1; echo "fuzz_batch s$s rc=$? : $(tail -1 $OUT/fuzz_batch_s$s.txt)"; done
for s in 51 52; do timeout -k 10 200 python3 profiles/fuzz_batch_pca.py 300 $s > $OUT/fuzz_batch_pca_s$s.txt 2>&1; echo "fuzz_batch_pca s$s rc=$? : $(tail -1 $OUT/fuzz_batch_pca_s$s.txt)"; done
timeout -k 10 300 python3 profiles/fuzz_generic.py 250 9000 > $OUT/fuzz_generic_9000.txt 2>&1; echo "fuzz_generic rc=$? : $(tail -1 $OUT/fuzz_generic_9000.txt)"
timeout -k 10 300 python3 profiles/fuzz_shapes.py 30 61 big > $OUT/fuzz_shapes_big_s61.txt 2>&1; echo "fuzz_shapes big s61 rc=$? : $(tail -1 $OUT/fuzz_shapes_big_s61.txt)"
