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

# Which Python line issues each ATen call?  torch.profiler's with_stack comes back empty on this build, so the calls are counted by a
# TorchDispatchMode (it follows autograd into its worker thread) with the innermost fiber_amd / bench frame of the Python stack; ops that
# launch no kernel (views, metadata) are dropped by name.
import traceback
from torch.utils._python_dispatch import TorchDispatchMode
NO_KERNEL = ("view", "reshape", "expand", "permute", "transpose", "t.default", "slice", "select", "unsqueeze", "squeeze", "as_strided", "detach", "alias",
             "empty", "_unsafe_view", "unbind", "split", "narrow", "size", "stride", "numel", "is_", "_local_scalar", "lift_fresh", "resize", "set_", "chunk",
             "_to_copy.default@same", "new_empty", "record_stream", "storage_offset", "sym_", "prim", "unfold", "diagonal", "movedim", "flatten", "contiguous")
sites = collections.Counter()


class Census(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = str(func)
        if not any(k in name for k in NO_KERNEL):
            frame = "?"
            for fr in reversed(traceback.extract_stack(limit=40)):
                if ("fiber_amd" in fr.filename or fr.filename.endswith("bench.py")) and "probes" not in fr.filename:
                    frame = f"{fr.filename.split('/root/repo/')[-1].split('repo/')[-1]}:{fr.lineno} {fr.name}"
                    break
            sites[(name, frame)] += 1
        return func(*args, **(kwargs or {}))


with Census():
    step()
torch.cuda.synchronize()
print(f"ATen calls (views / metadata dropped) in one step at B = {B}: {sum(sites.values())}")
for (name, frame), n in sites.most_common(70):
    print(f"{n:5d}  {name:36s} {frame}")
