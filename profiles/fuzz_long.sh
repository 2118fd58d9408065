ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/fuzz_long
mkdir -p $OUT
cd $ROOT
timeout -k 10 240 python3 profiles/fuzz_ops.py 150 11 > $OUT/fuzz_ops_s11.txt 2>&1; echo "fuzz_ops s11 rc=$? : $(tail -1 $OUT/fuzz_ops_s11.txt)"
timeout -k 10 240 python3 profiles/fuzz_ops_pca.py 200 12 > $OUT/fuzz_ops_pca_s12.txt 2>&1; echo "fuzz_ops_pca s12 rc=$? : $(tail -1 $OUT/fuzz_ops_pca_s12.txt)"
timeout -k 10 240 python3 profiles/fuzz_batch.py 120 13 > $OUT/fuzz_batch_s13.txt 2>&1; echo "fuzz_batch s13 rc=$? : $(tail -1 $OUT/fuzz_batch_s13.txt)"
timeout -k 10 240 python3 profiles/fuzz_batch_pca.py 300 14 > $OUT/fuzz_batch_pca_s14.txt 2>&1; echo "fuzz_batch_pca s14 rc=$? : $(tail -1 $OUT/fuzz_batch_pca_s14.txt)"
timeout -k 10 300 python3 profiles/fuzz_generic.py 250 5000 > $OUT/fuzz_generic_5000.txt 2>&1; echo "fuzz_generic rc=$? : $(tail -1 $OUT/fuzz_generic_5000.txt)"
timeout -k 10 240 python3 profiles/fuzz_more.py 60 15 > $OUT/fuzz_more_s15.txt 2>&1; echo "fuzz_more s15 rc=$? : $(tail -1 $OUT/fuzz_more_s15.txt)"
timeout -k 10 300 python3 profiles/fuzz_shapes.py 40 16 > $OUT/fuzz_shapes_s16.txt 2>&1; echo "fuzz_shapes s16 rc=$? : $(tail -1 $OUT/fuzz_shapes_s16.txt)"
