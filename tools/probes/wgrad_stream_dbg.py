"""Which parameter gradients differ when the weight-gradient GEMMs run on their own stream (ops.set_wgrad_stream)?  Debug helper."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from fiber_amd import lib, ops
from fiber_amd.config import make_config
from fiber_amd.modules import FIBERTransformerSS, fiber_utils
from oracle import cases, detgen
lib.load()
torch.manual_seed(0)
model = FIBERTransformerSS(make_config(**dict(cases.SWIN_B))).eval()
for n, p in model.named_parameters():
    if "alpha_" in n:
        p.data.fill_(0.5)
model.to("cuda")
fiber_utils.set_task(model)
b = detgen.synth_batch(8, 384, 40, 50265, seed=5, min_len=8)
bd = {k: (v.to("cuda") if isinstance(v, torch.Tensor) else [t.to("cuda") for t in v] if isinstance(v, list) and isinstance(v[0], torch.Tensor) else v) for k, v in b.items()}
bd["itm_labels_override"] = bd["itm_labels"]


def grads(on):
    ops.set_wgrad_stream(on)
    model.zero_grad(set_to_none=True)
    out = model(bd)
    sum(v for k, v in out.items() if "loss" in k).backward()
    torch.cuda.synchronize()
    ops.set_wgrad_stream(False)
    return {n: p.grad.detach().clone() for n, p in model.named_parameters() if p.grad is not None}


ref = grads(False)
ref2 = grads(False)
print("run-to-run (off, off) differing:", [n for n in ref if not torch.equal(ref[n], ref2[n])][:10])
got = grads(True)
bad = []
for n in ref:
    d = (ref[n].float() - got[n].float()).abs().max().item()
    if not (d <= 1e-3 * (ref[n].float().abs().max().item() + 1e-6)):
        bad.append((n, d, ref[n].float().abs().max().item(), bool(torch.isnan(got[n]).any())))
print(len(bad), "of", len(ref), "differ")
for r in bad[:60]:
    print(r)
