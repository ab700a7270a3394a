import sys, os
sys.path.insert(0, "/root/repo")
import torch
from fiber_amd import lib, ops, parallel
from fiber_amd.config import make_config
from fiber_amd.modules import FIBERTransformerSS, fiber_utils
from oracle import cases, detgen
lib.load()
dev = torch.device("cuda", 0)
torch.manual_seed(0)
model = FIBERTransformerSS(make_config(**dict(cases.TINY, text_dropout=0.1, drop_path_rate=0.1)))
parallel.freeze_unused(model, model.unused_parameter_names())
model.to(dev).train()
fiber_utils.set_task(model)
b = detgen.synth_batch(4, 96, 12, 1000, seed=40, min_len=6)
bd = {k: (v.to(dev) if isinstance(v, torch.Tensor) else [t.to(dev) for t in v] if isinstance(v, list) and isinstance(v[0], torch.Tensor) else v) for k, v in b.items()}
bd["itm_labels_override"] = bd["itm_labels"]
out = model(bd); sum(v for k, v in out.items() if "loss" in k).backward()
import collections
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU], record_shapes=True, with_stack=True) as prof:
    out = model(bd)
    sum(v for k, v in out.items() if "loss" in k).backward()
cnt = collections.Counter()
for e in prof.events():
    if e.name in ("aten::add", "aten::add_") and e.input_shapes and len(e.input_shapes[0]) >= 2:
        st = [s for s in (e.stack or []) if "fiber_amd" in s or "autograd" in s][:2]
        cnt[(e.name, str(e.input_shapes[:2]), tuple(st))] += 1
for k, v in cnt.most_common(25):
    print(v, k)
