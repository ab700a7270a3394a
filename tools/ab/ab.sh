cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for v in new base; do
  if [ $v = base ]; then export FIBER_HIP_LIB=$R/tools/ab/libfiber_hip_base.so; fi
  rocprofv3 --kernel-trace --stats -d $R/gpurun_out/ab_$v --output-format csv -- python $R/tools/op_bench.py 512 attn > $R/gpurun_out/ab_$v.log 2>&1
  echo == $v; tail -3 $R/gpurun_out/ab_$v.log; python - <<PY
import csv,glob
f=glob.glob("$R/gpurun_out/ab_$v/**/*kernel_stats.csv",recursive=True)
for r in csv.DictReader(open(f[0])):
    if 'win_' in r['Name']: print(r['Name'][:70], r['Calls'], r['AverageNs'], r['MinNs'], r['MaxNs'])
PY
done
