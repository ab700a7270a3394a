#!/bin/bash
# PMC passes over the window-attention kernels (tools/op_bench.py 512 attn): per-kernel averages over the stage mix.
# usage (GPU box): bash tools/pmc_win.sh  -> gpurun_out/pmc_win_*.txt
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
pass() {
  tag=$1; shift
  rocprofv3 --pmc "$@" -d gpurun_out/pmc_win_$tag --output-format csv -- python tools/op_bench.py 512 attn > gpurun_out/pmc_win_$tag.log 2>&1
  f=$(find gpurun_out/pmc_win_$tag -name "*counter_collection.csv" | head -1)
  [ -z "$f" ] && { echo "pass $tag: no counters (see gpurun_out/pmc_win_$tag.log)"; tail -3 gpurun_out/pmc_win_$tag.log; return; }
  python - "$f" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"]
    if "win_" in k and "kernel<" in k:
        acc[k.split("::")[1].split("(")[0][:34]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in sorted(acc.items()):
    print(k, {c: round(sum(v) / len(v)) for c, v in d.items()}, "n=", len(next(iter(d.values()))))
PY
}
pass a SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAVES
pass b SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_MFMA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
pass c SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_WAIT_INST_LDS SQ_INST_CYCLES_SALU SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE
