#!/bin/bash
# Evidence collection for one round on one MI355X (tools/evidence.sh r04; the round-3 run is kept as tools/pmc_r03.sh): per-stage and per-shape breakdowns, kernel-trace stats of the bench step, per-kernel
# MFMA-busy / HBM summary (tools/pmc_step.sh), FETCH_SIZE / WRITE_SIZE + kernel-trace passes on the dominant kernel.
TAG=${1:-r06}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python tools/stage_breakdown.py 256 5 > gpurun_out/${TAG}_stage_breakdown.md 2> gpurun_out/${TAG}_stage_breakdown.err
cp gpurun_out/stage_breakdown.json gpurun_out/${TAG}_stage_breakdown.json
python tools/shape_breakdown.py 256 3 > gpurun_out/${TAG}_shape_breakdown.log 2>&1
cp gpurun_out/shape_breakdown.json gpurun_out/${TAG}_shape_breakdown.json
cp gpurun_out/shape_breakdown.json profiles/${TAG}_shape_breakdown.json      # the passes below select the dominant kernel from it
# per-kernel durations with ONE stream (FIBER_NO_OVERLAP=1): with the text stack on its own stream kernels of the two streams share
# the CUs and every duration in the trace is inflated by its neighbours
FIBER_NO_OVERLAP=1 rocprofv3 --kernel-trace --stats -d gpurun_out/${TAG}_trace --output-format csv -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extras > gpurun_out/${TAG}_trace.log 2>&1
cp gpurun_out/${TAG}_trace/*/*kernel_stats.csv gpurun_out/${TAG}_kernel_stats_b256.csv
bash tools/pmc_step.sh > gpurun_out/${TAG}_pmc_step.log 2>&1
cp gpurun_out/pmc_step_summary.csv gpurun_out/${TAG}_pmc_step_summary.csv
rocprofv3 --kernel-trace --stats -d gpurun_out/${TAG}_dom_trace --output-format csv -- python tools/pmc_dominant.py 256 > gpurun_out/${TAG}_dom_trace.log 2>&1
rocprofv3 --pmc FETCH_SIZE -d gpurun_out/${TAG}_dom_fetch --output-format csv -- python tools/pmc_dominant.py 256 > gpurun_out/${TAG}_dom_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d gpurun_out/${TAG}_dom_write --output-format csv -- python tools/pmc_dominant.py 256 > gpurun_out/${TAG}_dom_write.log 2>&1
cp gpurun_out/${TAG}_dom_trace/*/*kernel_stats.csv gpurun_out/${TAG}_pmc_dominant_kernel_stats.csv
TAG=$TAG python - <<'PY'
import csv, glob, json, os
TAG = os.environ["TAG"]
info = json.loads([l for l in open(f"gpurun_out/{TAG}_dom_trace.log") if l.startswith("{")][-1])
kern = "gemm_nt_q8_kernel" if "q8" in info["kernel"] else ("gemm_tn_kernel<256" if "gemm_tn" in info["kernel"] else info["kernel"].split("<")[0])
def ctr(d, name):
    rows = [r for r in csv.DictReader(open(glob.glob(f"gpurun_out/{d}/*/*counter_collection.csv")[0])) if r["Counter_Name"] == name and kern in r["Kernel_Name"]]
    vals = [float(r["Counter_Value"]) for r in rows]
    return sum(vals) / max(1, len(vals)), len(vals)
def dur():
    best = None
    for r in csv.DictReader(open(f"gpurun_out/{TAG}_pmc_dominant_kernel_stats.csv")):
        if kern in r["Name"] and (best is None or float(r["TotalDurationNs"]) > float(best["TotalDurationNs"])):
            best = r
    return (float(best["AverageNs"]) / 1e3, float(best["MinNs"]) / 1e3, int(best["Calls"]), best["Name"][:100]) if best else (None, None, 0, None)
f, nf = ctr(f"{TAG}_dom_fetch", "FETCH_SIZE")
w, nw = ctr(f"{TAG}_dom_write", "WRITE_SIZE")
avg, mn, calls, name = dur()
traffic = f * 1024 * 2 + w * 1024
out = {"dominant": {"kernel": name, "selected": info["kernel"], "kind": info["kind"], "shape": info["shape"], "launches": calls,
                    "FETCH_SIZE_KB_raw": f, "WRITE_SIZE_KB_raw": w, "fetch_bytes_corrected_x2": f * 2048, "write_bytes": w * 1024,
                    "traffic_bytes_per_launch": traffic, "algorithmic_bytes_per_launch": info["algorithmic_bytes"],
                    "traffic_over_algorithmic": traffic / info["algorithmic_bytes"], "avg_us_under_rocprof": avg, "min_us_under_rocprof": mn,
                    "live_us_in_the_same_process": info["live_us"]}}
json.dump(out, open(f"gpurun_out/{TAG}_pmc_kernels.json", "w"), indent=1)
print(json.dumps(out, indent=1))
PY
# the bench line itself (50 steps, CPU baseline included), the same step with the round-3 tile boundary, and the small-batch sweep
python bench.py --steps 50 --warmup 10 > gpurun_out/${TAG}_bench_b256.log 2>&1
FIBER_GEMM_STAGGER=1 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras > gpurun_out/${TAG}_bench_b256_stagger_kept.log 2>&1
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras > gpurun_out/${TAG}_bench_b256_again.log 2>&1
for b in 8 16 32 64; do
  python bench.py --batch $b --steps 30 --warmup 5 --no-cpu-baseline --no-extras 2>/dev/null | python -c "import json,sys; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print($b, d['value'], d['ms_per_step'])"
done > gpurun_out/${TAG}_batch_sweep.log 2>&1
# round 6: the fused LayerNorm-Mlp kernels (csrc/mlp_rows.hip) against the separate kernels they replace, and their SQ counters
python tools/lnmlp_bench.py 512 > gpurun_out/${TAG}_lnmlp_bench.log 2>&1
bash tools/pmc_lnmlp.sh ${TAG}_lnmlp > gpurun_out/${TAG}_pmc_lnmlp.log 2>&1
