import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from fiber_amd import lib, ops
lib.load()
torch.manual_seed(0)
M = 8 * 9216
cols = torch.randn(M, 64, device="cuda").to(torch.bfloat16)
dy2 = torch.randn(M, 128, device="cuda").to(torch.bfloat16)
ref = (dy2.float().t() @ cols.float())
side = ops.wgrad_stream(cols.device)


class F(torch.autograd.Function):
    @staticmethod
    def forward(ctx, w, mode):
        ctx.mode = mode
        return w.sum() * 0 + 1

    @staticmethod
    def backward(ctx, g):
        mode = ctx.mode
        if mode == "raw":
            return ops.wgrad(dy2, cols), None
        if mode == "clone":
            return ops.wgrad(dy2, cols, post=lambda w_, b_: w_.clone()), None
        if mode == "slice":
            return ops.wgrad(dy2, cols, post=lambda w_, b_: torch.cat([w_[:, :48].reshape(128, 3, 4, 4).reshape(128, 48), w_[:, 48:]], 1)), None
        if mode == "index":
            inv = torch.arange(127, -1, -1, device="cuda")
            return ops.wgrad(dy2, cols, post=lambda w_, b_: w_[inv][inv]), None


for mode in ("raw", "clone", "slice", "index"):
    bad = 0
    for i in range(20):
        w = torch.nn.Parameter(torch.zeros(128, 64, device="cuda"))
        ops.set_wgrad_stream(True)
        F.apply(w, mode).backward()
        torch.cuda.synchronize()
        ops.set_wgrad_stream(False)
        err = (w.grad - ref).abs().max().item() / ref.abs().max().item()
        bad += err > 1e-2
    print(mode, "bad runs", bad, "of 20", flush=True)
