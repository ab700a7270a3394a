#!/bin/bash
# Kernel durations + HBM bytes of the window-attention kernels (tools/op_bench.py 512 attn, head-major layout).
# usage (GPU box): bash tools/pmc_win_hbm.sh [tag]  -> gpurun_out/win_hbm_<tag>.txt      env FIBER_WIN_FUSED=0|1 selects the backward
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
tag=${1:-run}
out=gpurun_out/win_hbm_$tag.txt
: > $out
rocprofv3 --kernel-trace --stats -d gpurun_out/win_kt_$tag --output-format csv -- python tools/op_bench.py 512 attn > gpurun_out/win_kt_$tag.log 2>&1
f=$(find gpurun_out/win_kt_$tag -name "*kernel_stats.csv" | head -1)
echo "== kernel stats" >> $out
grep -E "win_|Name" $f | cut -c1-200 >> $out
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c -d gpurun_out/win_pmc_${tag}_$c --output-format csv -- python tools/op_bench.py 512 attn > gpurun_out/win_pmc_${tag}_$c.log 2>&1
  f=$(find gpurun_out/win_pmc_${tag}_$c -name "*counter_collection.csv" | head -1)
  python - "$f" $c >> $out <<'PY'
import csv, sys, collections
acc = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"]
    if "win_" in k:
        acc[(k.split("::")[-1].split("(")[0][:40], r.get("Grid_Size", ""))].append(float(r["Counter_Value"]))
print("==", sys.argv[2], "(raw counter, KB units per the guide; x2 correction for reads not applied)")
for k, v in sorted(acc.items()):
    print(k, "avg", round(sum(v) / len(v)), "n", len(v))
PY
done
cat $out
