#!/bin/bash
# Round-2 evidence collection on one MI355X: kernel-trace stats of the bench step, per-kernel MFMA-busy / HBM summary
# (tools/pmc_step.sh), FETCH_SIZE / WRITE_SIZE + kernel-trace passes on the two dominant hand-written kernels.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
# per-kernel durations are taken with ONE stream (FIBER_NO_OVERLAP=1): with the text stack on its own stream kernels of the two
# streams share the CUs and every duration in the trace is inflated by its neighbours
FIBER_NO_OVERLAP=1 rocprofv3 --kernel-trace --stats -d gpurun_out/r02_trace --output-format csv -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extras > gpurun_out/r02_trace.log 2>&1
bash tools/pmc_step.sh > gpurun_out/r02_pmc_step.log 2>&1
for k in gemm tn; do
  rocprofv3 --kernel-trace --stats -d gpurun_out/r02_${k}_trace --output-format csv -- python tools/pmc_$k.py 256 > gpurun_out/r02_${k}_trace.log 2>&1
  rocprofv3 --pmc FETCH_SIZE -d gpurun_out/r02_${k}_fetch --output-format csv -- python tools/pmc_$k.py 256 > gpurun_out/r02_${k}_fetch.log 2>&1
  rocprofv3 --pmc WRITE_SIZE -d gpurun_out/r02_${k}_write --output-format csv -- python tools/pmc_$k.py 256 > gpurun_out/r02_${k}_write.log 2>&1
done
python - <<'PY'
import csv, glob, json
def ctr(d, name, kern):
    vals = [float(r["Counter_Value"]) for r in csv.DictReader(open(glob.glob(f"gpurun_out/{d}/*/*counter_collection.csv")[0]))
            if r["Counter_Name"] == name and kern in r["Kernel_Name"]]
    return sum(vals) / max(1, len(vals)), len(vals)
def dur(d, kern):
    for r in csv.DictReader(open(glob.glob(f"gpurun_out/{d}/*/*kernel_stats.csv")[0])):
        if kern in r["Name"]:
            return float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3, int(r["Calls"])
    return None, None, 0
out = {}
for tag, kern, alg in (("gemm", "gemm_nt_wide_persist2_kernel<2, 4, 1", 2 * (294912 * 512 + 2048 * 512 + 2 * 294912 * 2048)),
                       ("tn", "gemm_tn_kernel<256", 2 * (294912 * 2048 + 294912 * 512) + 4 * 2048 * 512)):
    f, nf = ctr(f"r02_{tag}_fetch", "FETCH_SIZE", kern)
    w, nw = ctr(f"r02_{tag}_write", "WRITE_SIZE", kern)
    avg, mn, calls = dur(f"r02_{tag}_trace", kern)
    traffic = f * 1024 * 2 + w * 1024
    out[tag] = {"kernel": kern, "launches": calls, "FETCH_SIZE_KB_raw": f, "WRITE_SIZE_KB_raw": w, "fetch_bytes_corrected_x2": f * 2048,
                "write_bytes": w * 1024, "traffic_bytes_per_launch": traffic, "algorithmic_bytes_per_launch": alg,
                "traffic_over_algorithmic": traffic / alg, "avg_us_under_rocprof": avg, "min_us_under_rocprof": mn}
json.dump(out, open("gpurun_out/r02_pmc_kernels.json", "w"), indent=1)
print(json.dumps(out, indent=1))
PY
