#!/bin/bash
# Per-kernel time of one training step in two trees on the same box: rocprofv3 --kernel-trace --stats of bench.py (7 steps) in .ab/<tag> and in the
# working tree, then tools/probes/kernel_diff.py prints the kernels whose time per step differs.   tools/probes/kernel_diff.sh [tag]
TAG=${1:-r05}
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
export TMPDIR=/tmp
for t in ".ab/$TAG" "."; do
  name=$(echo -n "$t" | tr -c "a-zA-Z0-9" "_")
  rm -rf "$ROOT/gpurun_out/kd_$name"
  (cd "$ROOT/$t" && FIBER_NO_OVERLAP=1 rocprofv3 --kernel-trace --stats -d "$ROOT/gpurun_out/kd_$name" --output-format csv -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extras > "$ROOT/gpurun_out/kd_$name.log" 2>&1)
done
python "$ROOT/tools/probes/kernel_diff.py" "$ROOT/gpurun_out/kd__ab_${TAG}" "$ROOT/gpurun_out/kd__" 7
