"""Which Python lines of fiber_amd launch the step's ATen kernels?  One profiled training step (torch.profiler, with_stack), ATen ops that
launched a device kernel grouped by the innermost fiber_amd / bench frame.   python tools/probes/aten_sites.py [batch]"""
import collections
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import bench
from fiber_amd import lib, ops, parallel
from fiber_amd.config import named_config
from fiber_amd.modules import FIBERTransformerSS, fiber_utils

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
dev = torch.device("cuda", 0)
lib.load()
torch.manual_seed(0)
ops.manual_seed(0)
cfg = named_config("task_pretrain_mlm_itm", per_gpu_batchsize=B, num_gpus=1, max_steps=100000, warmup_steps=10000, draw_false_image=1)
model = FIBERTransformerSS(cfg)
for n, p in model.named_parameters():
    if "alpha_" in n:
        p.data.fill_(0.5)
parallel.freeze_unused(model, model.unused_parameter_names())
model.to(dev).train()
fiber_utils.set_task(model)
(opt,), (sched,) = model.configure_optimizers()
batch = bench.synth_batch(B, cfg["image_size"], cfg["max_text_len"], cfg["vocab_size"], dev, seed=0)


def step():
    ops.set_rng_step(model.global_step)
    out = model(batch)
    sum(v for k, v in out.items() if "loss" in k).backward()
    opt.step()
    sched["scheduler"].step()
    opt.zero_grad(set_to_none=True)
    model.global_step += 1


for _ in range(3):
    step()
torch.cuda.synchronize()
from torch.profiler import ProfilerActivity, profile
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    step()
    torch.cuda.synchronize()
sites = collections.Counter()
kern = collections.Counter()
for ev in prof.events():
    if not ev.name.startswith("aten::") or ev.device_time_total <= 0 or not getattr(ev, "kernels", None):
        continue
    if not ev.kernels:
        continue
    frame = "?"
    for fr in (ev.stack or []):
        if "fiber_amd" in fr or "bench.py" in fr or "aten_sites" in fr:
            frame = fr.split("/root/repo/")[-1] if "/root/repo/" in fr else fr
            break
    sites[(ev.name, frame)] += len(ev.kernels)
    kern[ev.name] += len(ev.kernels)
print("ATen ops with device kernels in one step:", sum(kern.values()))
for (name, frame), c in sites.most_common(60):
    print(f"{c:5d}  {name:28s} {frame[:150]}")
