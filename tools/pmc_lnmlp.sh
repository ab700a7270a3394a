#!/bin/bash
# usage (on the GPU box): tools/pmc_lnmlp.sh <tag> [env...]  -- SQ counters of the fused LN-MLP kernels (tools/lnmlp_probe.py)
tag="$1"; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
env "$@" rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE \
  -d gpurun_out/pmc_$tag --output-format csv -- python tools/lnmlp_probe.py 256 > gpurun_out/pmc_$tag.log 2>&1
f=$(find gpurun_out/pmc_$tag -name "*counter_collection.csv" | head -1)
python - "$f" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    if "ln_mlp" in r["Kernel_Name"]:
        acc[r["Kernel_Name"][:70] + " grid " + r.get("Grid_Size", "?")][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in acc.items():
    print(k, {c: round(sum(v) / len(v)) for c, v in d.items()}, "n=", len(next(iter(d.values()))))
PY
