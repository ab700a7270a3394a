import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from fiber_amd import lib, ops
lib.load()
torch.manual_seed(0)
B = 8
img = torch.randn(B, 3, 384, 384, device="cuda")
dy = torch.randn(B, 9216, 128, device="cuda").to(torch.bfloat16)
mode = sys.argv[1] if len(sys.argv) > 1 else "base"


class PE(torch.autograd.Function):
    @staticmethod
    def forward(ctx, img, weight, bias):
        Bn, _, H, W = img.shape
        rows = Bn * (H // 4) * (W // 4)
        cols = torch.empty((rows, 64), dtype=torch.bfloat16, device=img.device)
        lib.call("fiber_im2col_patch4", lib.ptr(img), lib.ptr(cols), Bn, H, W)
        wp = torch.zeros((128, 64), dtype=torch.bfloat16, device=img.device)
        wp[:, :48] = weight.detach().reshape(128, 48).to(torch.bfloat16)
        y, _ = ops.gemm_nt(cols, wp, bias)
        ctx.save_for_backward(cols)
        return y.view(Bn, rows // Bn, 128)

    @staticmethod
    def backward(ctx, dyy):
        (cols,) = ctx.saved_tensors
        dy2 = dyy.view(-1, 128)
        if mode == "base":
            dw, db = ops.wgrad(dy2, cols, want_bias=True, post=lambda w_, b_: (w_[:, :48].reshape(128, 3, 4, 4), b_))
        elif mode == "nobias":
            dw, db = ops.wgrad(dy2, cols, post=lambda w_, b_: w_[:, :48].reshape(128, 3, 4, 4)), None
        elif mode == "nopost":
            dw, db = ops.wgrad(dy2, cols, want_bias=True)
            dw = None
        elif mode == "clonecols":
            cols = cols.clone()
            dw, db = ops.wgrad(dy2, cols, want_bias=True, post=lambda w_, b_: (w_[:, :48].reshape(128, 3, 4, 4), b_))
        return None, dw, db


def run(on):
    w = torch.nn.Parameter(torch.randn(128, 3, 4, 4, device="cuda") * 0.1) if False else W
    W.grad = None; Bi.grad = None
    ops.set_wgrad_stream(on)
    PE.apply(img, W, Bi).backward(dy)
    torch.cuda.synchronize()
    ops.set_wgrad_stream(False)
    return (W.grad.clone() if W.grad is not None else None), Bi.grad.clone()


W = torch.nn.Parameter(torch.randn(128, 3, 4, 4, device="cuda") * 0.1)
Bi = torch.nn.Parameter(torch.zeros(128, device="cuda"))
a = run(False)
bad_w = bad_b = 0
for i in range(30):
    b = run(True)
    if a[0] is not None:
        bad_w += (a[0] - b[0]).abs().max().item() > 1e-3 * a[0].abs().max().item()
    bad_b += (a[1] - b[1]).abs().max().item() > 1e-3 * a[1].abs().max().item()
print(mode, "bad dw", bad_w, "bad db", bad_b, "of 30")
