cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_hip_ops.py -m gpu -x -q -k "attn or window or win or mha" > gpurun_out/rows4_tests.log 2>&1; echo "tests exit $?" >> gpurun_out/rows4_tests.log
for i in 1 2; do
  echo "== new (permlane swaps) run $i"; timeout 300 python tools/op_bench.py 512 attn
  echo "== old (ds_bpermute) run $i"; FIBER_HIP_LIB=$GRAFT_REPO_ROOT/tools/bin/libfiber_shfl.so timeout 300 python tools/op_bench.py 512 attn
done > gpurun_out/r04_rows4_ab.log 2>&1
tail -5 gpurun_out/rows4_tests.log
cat gpurun_out/r04_rows4_ab.log
