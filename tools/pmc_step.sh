#!/bin/bash
# Per-kernel PMC summary of a short training run: separate rocprofv3 --pmc passes (kernel trace in its own pass).
# usage: tools/pmc_step.sh   -> gpurun_out/pmc_step_{busy,mfma,fetch,write}/..., gpurun_out/pmc_step_summary.csv
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export FIBER_NO_OVERLAP=1    # one stream: per-kernel counters and durations without a concurrent neighbour
CMD="python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras"
rocprofv3 --kernel-trace --stats -d gpurun_out/pmc_step_trace --output-format csv -- $CMD > gpurun_out/pmc_step_trace.log 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES -d gpurun_out/pmc_step_mfma --output-format csv -- $CMD > gpurun_out/pmc_step_mfma.log 2>&1
rocprofv3 --pmc FETCH_SIZE -d gpurun_out/pmc_step_fetch --output-format csv -- $CMD > gpurun_out/pmc_step_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d gpurun_out/pmc_step_write --output-format csv -- $CMD > gpurun_out/pmc_step_write.log 2>&1
python - <<'PY'
import csv, glob, collections
def short(n):
    n = n.replace("void (anonymous namespace)::", "").replace("(anonymous namespace)::", "")
    return n.split("(")[0][:70]
dur = {}
for r in csv.DictReader(open(glob.glob("gpurun_out/pmc_step_trace/*/*kernel_stats.csv")[0])):
    dur[short(r["Name"])] = (int(r["Calls"]), float(r["AverageNs"]), float(r["TotalDurationNs"]))
ctr = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.defaultdict(lambda: collections.defaultdict(int))
for d in ("mfma", "fetch", "write"):
    for r in csv.DictReader(open(glob.glob(f"gpurun_out/pmc_step_{d}/*/*counter_collection.csv")[0])):
        k = short(r["Kernel_Name"])
        ctr[k][r["Counter_Name"]] += float(r["Counter_Value"])
        cnt[k][r["Counter_Name"]] += 1
tot = sum(v[2] for v in dur.values())
rows = sorted(dur.items(), key=lambda kv: -kv[1][2])[:28]
with open("gpurun_out/pmc_step_summary.csv", "w") as f:
    f.write("kernel,calls,avg_us,share_pct,mfma_busy_pct,hbm_read_GB_per_call(x2 corrected),hbm_write_GB_per_call,hbm_TBps\n")
    for k, (calls, avg, total) in rows:
        c = ctr.get(k, {})
        n = max(1, cnt[k].get("GRBM_GUI_ACTIVE", 1))
        gui = c.get("GRBM_GUI_ACTIVE", 0.0)
        # SQ_VALU_MFMA_BUSY_CYCLES sums over SIMDs (1024); GRBM_GUI_ACTIVE sums over XCDs (8)
        mfma = 100.0 * (c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / 1024.0) / (gui / 8.0) if gui else float("nan")
        rd = c.get("FETCH_SIZE", 0.0) * 1024 * 2 / max(1, cnt[k].get("FETCH_SIZE", 1)) / 1e9
        wr = c.get("WRITE_SIZE", 0.0) * 1024 / max(1, cnt[k].get("WRITE_SIZE", 1)) / 1e9
        bw = (rd + wr) / (avg * 1e-9) / 1e3 if avg else 0.0
        line = f"{k},{calls},{avg / 1e3:.1f},{100 * total / tot:.1f},{mfma:.1f},{rd:.3f},{wr:.3f},{bw:.2f}"
        f.write(line + "\n")
        print(line)
PY
