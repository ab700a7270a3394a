run() { (cd $1 && env $2 timeout 400 python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-extras 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1 $2', d['ms_per_step'])"); }
for i in 1 2; do
run .ab/r05 X=1
run . X=1
run . FIBER_LN_MLP=0
run . FIBER_TN_ROWMAP=0
run . FIBER_HM_REFRESH=0
done
