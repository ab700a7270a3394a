#!/bin/bash
# LDS / issue counters of the TN weight-gradient kernel and, for comparison, the NT q8 kernel (tools/pmc_tn.py, tools/pmc_dominant.py shapes).
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
pass() {
  tag=$1; shift
  rocprofv3 --pmc "$@" -d gpurun_out/pmc_tnl_$tag --output-format csv -- python tools/pmc_tn.py 256 > gpurun_out/pmc_tnl_$tag.log 2>&1
  f=$(find gpurun_out/pmc_tnl_$tag -name "*counter_collection.csv" | head -1)
  python - "$f" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"]
    if "gemm_tn_kernel" in k:
        acc["gemm_tn"][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in acc.items():
    print(k, {c: round(sum(v) / len(v)) for c, v in d.items()})
PY
}
pass a SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_WAVE_CYCLES
pass b SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_SALU
