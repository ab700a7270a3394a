#!/bin/bash
# usage: tools/pmc_pass.sh "<counters>" <tag> -- <cmd...> ; prints per-kernel counter averages for gemm kernels
ctr="$1"; tag="$2"; shift 3
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --pmc $ctr -d gpurun_out/pmc_$tag --output-format csv -- "$@" > gpurun_out/pmc_$tag.log 2>&1
f=$(find gpurun_out/pmc_$tag -name "*counter_collection.csv" | head -1)
python - "$f" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    if "gemm_nt" in r["Kernel_Name"]:
        acc[r["Kernel_Name"][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in acc.items():
    print(k, {c: round(sum(v) / len(v), 1) for c, v in d.items()}, "n=", len(next(iter(d.values()))))
PY
