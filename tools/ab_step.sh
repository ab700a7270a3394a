#!/bin/bash
# Same-box A/B of the whole training step: the tree at .ab/<tag> (an older commit exported with `git archive <commit> bench.py fiber_amd fiber oracle
# include | tar -x -C .ab/<tag>` and built there) against the working tree, interleaved, `reps` times.   tools/ab_step.sh [tag] [reps] [steps]
TAG=${1:-r05}; REPS=${2:-2}; STEPS=${3:-30}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
for i in $(seq $REPS); do
  for t in ".ab/$TAG" "."; do
    (cd "$ROOT/$t" && timeout 400 python bench.py --steps $STEPS --warmup 8 --no-cpu-baseline --no-extras 2>/dev/null | tail -1 |
      python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$t', d['ms_per_step'], d['value'], d.get('roofline',{}).get('frac'))")
  done
done
