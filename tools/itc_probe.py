"""Where does the MLM+ITC+ITM step spend its time?  (debug helper)"""
import faulthandler, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
faulthandler.dump_traceback_later(int(os.environ.get("WATCHDOG", "90")), exit=True)
import torch
from fiber_amd import lib, ops
from fiber_amd.config import named_config
from fiber_amd.modules import FIBERTransformerSS, fiber_utils
from bench import synth_batch
lib.load()
import threading


def report():
    tr = lib._TRACE or []
    seen = {}
    for st, name, sc, ev in tr:
        if not ev.query() and st not in seen:
            seen[st] = (name, sc)
    print("FIRST UNFINISHED LAUNCH PER STREAM:", seen, "of", len(tr), "traced", flush=True)
    last = {}
    for st, name, sc, ev in tr:
        last.setdefault(st, []).append(name)
    for st, names in last.items():
        idx = next((i for i, (s2, n2, sc2, e2) in enumerate([t for t in tr if t[0] == st]) if not e2.query()), None)
        print("stream", st, "launches", len(names), "first unfinished index", idx, "context", names[max(0, (idx or 0) - 3):(idx or 0) + 4], flush=True)


threading.Timer(int(os.environ.get("REPORT_AT", "40")), report).start()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
qs = int(sys.argv[2]) if len(sys.argv) > 2 else 256
t = time.time()
cfg = named_config("task_pretrain_mlm_itm_itc", per_gpu_batchsize=B, itc_queue_size=qs)
m = FIBERTransformerSS(cfg)
print("construct", round(time.time() - t, 1), flush=True); t = time.time()
m.to("cuda").train()
torch.cuda.synchronize(); print("to device", round(time.time() - t, 1), flush=True)
fiber_utils.set_task(m)
batch = synth_batch(B, 384, 40, 50265, torch.device("cuda"), 0)
for i in range(3):
    t = time.time()
    out = m(batch)
    torch.cuda.synchronize(); t1 = time.time()
    sum(v for k, v in out.items() if "loss" in k).backward()
    torch.cuda.synchronize()
    print(f"step {i}: fwd {t1 - t:.2f}s bwd {time.time() - t1:.2f}s", {k: round(float(v), 4) for k, v in out.items() if 'loss' in k}, flush=True)
    m.zero_grad(set_to_none=True)
