"""Launch pattern for the PMC passes on the dominant kernel of the bench step: the (entry point, shape) that
the latest profiles/rNN_shape_breakdown.json ranks first, run 20 times.  Used under `rocprofv3 --pmc FETCH_SIZE`, `--pmc WRITE_SIZE` and
`--kernel-trace --stats` (separate passes, tools/evidence.sh)."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from fiber_amd import lib
lib.load()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
dom = bench._dominant_from_profile(B)
assert dom is not None, "profiles/rNN_shape_breakdown.json missing or taken at another batch"
row, _ = dom
# bench.time_dominant_kernel builds the launch; here the same closure is run a fixed 20 + 5 times
out = bench.time_dominant_kernel(B, torch.device("cuda", 0))
print(json.dumps({"kernel": out["kernel"], "shape": out["shape"], "kind": out["kind"], "algorithmic_bytes": out["algorithmic_bytes"], "live_us": out["us"]}))
