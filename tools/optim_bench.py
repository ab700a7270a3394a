"""GPU time of one optimizer step on the FIBER-Base parameter set: FiberAdamW (one kernel per group, bf16 copies refreshed)
vs torch.optim.AdamW (foreach) + the bf16 re-casts it leaves to the next forward."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fiber_amd import lib, ops
from fiber_amd.config import make_config
from fiber_amd.modules import FIBERTransformerSS, fiber_utils
lib.load()
m = FIBERTransformerSS(make_config()).to("cuda")
(opt,), _ = m.configure_optimizers()
ps = [p for g in opt.param_groups for p in g["params"]]
for p in ps:
    p.grad = torch.randn_like(p) * 1e-3
    if p.dim() == 2:
        ops.bf16_weight(p)
ref = torch.optim.AdamW([{"params": g["params"], "lr": g["lr"], "weight_decay": g["weight_decay"]} for g in opt.param_groups],
                        lr=1e-5, eps=1e-8, betas=(0.9, 0.98))


def timed(fn, n=5):
    fn(); torch.cuda.synchronize()
    t = time.perf_counter()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n, (time.perf_counter() - t) / n * 1e3


def torch_step():
    ref.step()
    for p in ps:
        if p.dim() == 2:
            p.detach().to(torch.bfloat16)


print(type(opt).__name__, "gpu ms, wall ms:", timed(opt.step))
print("torch AdamW + casts gpu ms, wall ms:", timed(torch_step))
print("params", sum(p.numel() for p in ps) / 1e6, "M in", len(ps), "tensors")
