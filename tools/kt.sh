#!/bin/bash
# Kernel-trace summary of one command on the GPU box:  bash tools/kt.sh <tag> <regex> <command ...>
# -> gpurun_out/kt_<tag>/ (rocprofv3 csv); prints, per kernel matching the regex AND per grid size, calls / average / min / max us
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
tag=$1; pat=$2; shift 2
timeout 300 rocprofv3 --kernel-trace --stats -d gpurun_out/kt_$tag --output-format csv -- "$@" > gpurun_out/kt_$tag.log 2>&1
f=$(find gpurun_out/kt_$tag -name "*kernel_trace.csv" | head -1)
[ -z "$f" ] && { echo "no trace file"; tail -5 gpurun_out/kt_$tag.log; exit 1; }
python - "$f" "$pat" <<'PY'
import csv, re, sys, collections
acc = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Kernel_Name"].replace("(anonymous namespace)::", "")
    if re.search(sys.argv[2], n):
        acc[(re.sub(r"\(.*", "", n)[:60], r.get("Grid_Size_X", "") + "x" + r.get("Grid_Size_Y", "") + "x" + r.get("Grid_Size_Z", ""))].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for (n, g), v in sorted(acc.items()):
    v2 = sorted(v)[len(v) // 4:]                      # drop the warm-up quarter from the average
    print(f"{n:60s} grid {g:>16s} calls {len(v):4d}  avg {sum(v2) / len(v2):9.1f} us  min {min(v):9.1f}  max {max(v):9.1f}")
PY
