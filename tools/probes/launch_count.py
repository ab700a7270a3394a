"""Device launches of ONE steady-state training step, counted with torch.profiler (device activity only): kernels (ours + ATen + library)
and copy / fill commands.  bench.py runs this as a child process with a deadline and puts the JSON line into `extra.launches_per_step`.
    python tools/probes/launch_count.py [batch]"""
import json
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import bench
from fiber_amd import lib, ops, parallel
from fiber_amd.config import named_config
from fiber_amd.modules import FIBERTransformerSS, fiber_utils

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
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


for _ in range(4):                                            # (the first steps zero-fill the optimizer state and build its tables)
    step()
torch.cuda.synchronize()
from torch.profiler import ProfilerActivity, profile
STEPS = 2
with profile(activities=[ProfilerActivity.CUDA]) as prof:
    for _ in range(STEPS):
        step()
    torch.cuda.synchronize()
kern = copies = ours = 0
for e in prof.events():
    if e.device_type != torch.autograd.DeviceType.CUDA:
        continue
    name = e.name or ""
    if "Memcpy" in name or "Memset" in name or "copyBuffer" in name or "fillBuffer" in name:
        copies += 1
    else:
        kern += 1
        ours += "anonymous namespace" in name or "_GLOBAL__N_1" in name
print(json.dumps({"batch": B, "steps_profiled": STEPS, "kernels": round(kern / STEPS, 1), "of_which_fiber_hip": round(ours / STEPS, 1),
                  "copy_or_fill_commands": round(copies / STEPS, 1), "counted_with": "torch.profiler, device activity, steady-state eager steps"}), flush=True)
