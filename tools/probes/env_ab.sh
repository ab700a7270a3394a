#!/bin/bash
# The bench step with one environment switch at two values, interleaved.   tools/probes/env_ab.sh NAME a b [reps] [steps]
NAME=$1; A=$2; B=$3; REPS=${4:-2}; STEPS=${5:-30}
for i in $(seq $REPS); do for v in $A $B; do
  env $NAME=$v timeout 400 python bench.py --steps $STEPS --warmup 8 --no-cpu-baseline --no-extras 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$NAME=$v', d['ms_per_step'], d['loss'])"
done; done
