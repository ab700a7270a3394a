import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from fiber_amd import lib, ops
lib.load()
torch.manual_seed(0)
B = 8
img = torch.randn(B, 3, 384, 384, device="cuda")
w = torch.nn.Parameter(torch.randn(128, 3, 4, 4, device="cuda") * 0.1)
bias = torch.nn.Parameter(torch.zeros(128, device="cuda"))
dy = torch.randn(B, 9216, 128, device="cuda").to(torch.bfloat16)


def run(on, big_first=False):
    ops.set_wgrad_stream(on)
    w.grad = None; bias.grad = None
    y = ops.patch_embed_proj(img, w, bias)
    if big_first:   # something long on the main stream right before backward returns
        pass
    y.backward(dy)
    torch.cuda.synchronize()
    ops.set_wgrad_stream(False)
    return w.grad.clone(), bias.grad.clone()


a = run(False)
for i in range(3):
    b = run(True)
    print("stream on run", i, "dw max diff", (a[0] - b[0]).abs().max().item(), "db", (a[1] - b[1]).abs().max().item(), "ref max", a[0].abs().max().item())
# raw wgrad + post outside autograd (inside no_grad, so the side stream is taken)
cols = torch.randn(B * 9216, 64, device="cuda").to(torch.bfloat16)
dy2 = dy.view(-1, 128)
with torch.no_grad():
    ops.set_wgrad_stream(False)
    r0 = ops.wgrad(dy2, cols, want_bias=True, post=lambda w_, b_: (w_[:, :48].reshape(128, 3, 4, 4), b_))
    torch.cuda.synchronize()
print("split count", lib.plain("fiber_gemm_tn_splits", B * 9216, 128, 64))
