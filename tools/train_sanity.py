"""Longer training sanity on the real configuration: N optimizer steps on ONE fixed synthetic batch (so the loss must fall),
prints the loss every 10 steps and fails on a non-finite value.  python tools/train_sanity.py [batch=64] [steps=80]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fiber_amd import lib, ops
from fiber_amd.config import named_config
from fiber_amd.modules import FIBERTransformerSS, fiber_utils
from bench import synth_batch
lib.load()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 80
torch.manual_seed(0)
ops.manual_seed(0)
cfg = named_config("task_pretrain_mlm_itm", per_gpu_batchsize=B, max_steps=steps, warmup_steps=0.1, learning_rate=5e-5, draw_false_image=1)
m = FIBERTransformerSS(cfg)
for n, p in m.named_parameters():
    if "alpha_" in n:
        p.data.fill_(0.5)
m.to("cuda").train()
fiber_utils.set_task(m)
(opt,), (sch,) = m.configure_optimizers()
batch = synth_batch(B, 384, 40, 50265, torch.device("cuda"), 0)
first = None
for i in range(steps):
    out = m(batch)
    loss = sum(v for k, v in out.items() if "loss" in k)
    loss.backward()
    opt.step(); sch["scheduler"].step(); opt.zero_grad(set_to_none=True)
    if i % 10 == 0 or i == steps - 1:
        lv = {k: round(float(v.detach()), 4) for k, v in out.items() if "loss" in k}
        assert all(map(lambda x: x == x and abs(x) < 1e4, lv.values())), lv
        first = first or lv
        print(f"step {i}: {lv}", flush=True)
print("first", first, "last", lv)
assert lv["mlm_loss"] < first["mlm_loss"] - 0.5, "MLM loss did not fall on a fixed batch"
print("train sanity ok")
