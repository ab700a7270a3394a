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
PTR = {}
PIN = len(sys.argv) > 1 and sys.argv[1] == "pin"


class F(torch.autograd.Function):
    @staticmethod
    def forward(ctx, w, b, mode):
        ctx.mode = mode
        return w.sum() * 0 + (b.sum() * 0 if b is not None else 0) + 1

    @staticmethod
    def backward(ctx, g):
        mode = ctx.mode
        if mode == "4d":
            dw = ops.wgrad(dy2, cols, post=lambda w_, b_: w_[:, :48].reshape(128, 3, 4, 4))
            PTR["dw"] = dw.data_ptr()
            return dw, None, None
        if mode == "2d_bias":
            dw, db = ops.wgrad(dy2, cols, want_bias=True, post=lambda w_, b_: (w_[:, :48].reshape(128, 48), b_))
            PTR["dw"] = dw.data_ptr()
            return dw, db, None
        if mode == "4d_bias":
            dw, db = ops.wgrad(dy2, cols, want_bias=True, post=lambda w_, b_: (w_[:, :48].reshape(128, 3, 4, 4), b_))
            PTR["dw"] = dw.data_ptr()
            return dw, db, None
        if mode == "4d_bias_clone":
            dw, db = ops.wgrad(dy2, cols, want_bias=True, post=lambda w_, b_: (w_[:, :48].reshape(128, 3, 4, 4), b_.clone()))
            PTR["dw"] = dw.data_ptr()
            return dw, db, None


for mode in ("4d", "2d_bias", "4d_bias", "4d_bias_clone"):
    bad = stolen = 0
    for i in range(60):
        shape = (128, 48) if mode == "2d_bias" else (128, 3, 4, 4)
        w = torch.nn.Parameter(torch.zeros(*shape, device="cuda"))
        b = torch.nn.Parameter(torch.zeros(128, device="cuda")) if "bias" in mode else None
        holder = torch.nn.ParameterList([q for q in (w, b) if q is not None])
        if PIN:
            ops.enable_wgrad_stream(holder)
        ops.set_wgrad_stream(True)
        F.apply(w, b, mode).backward()
        torch.cuda.synchronize()
        ops.set_wgrad_stream(False)
        stolen += w.grad.data_ptr() == PTR["dw"]
        err = (w.grad.reshape(128, 48) - ref[:, :48]).abs().max().item() / ref.abs().max().item()
        bad += err > 1e-2
    print(mode, "bad", bad, "stolen", stolen, "of 60", flush=True)
