"""Achieved HBM GB/s of the memory-bound kernels at bench-size tensors (direct C-ABI calls, no autograd)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fiber_amd import lib, ops
lib.load()
dev, BF = "cuda", torch.bfloat16
def t(fn, reps=20):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps
for M, C in ((147456, 512), (2359296 // 2, 128), (36864, 1024)):
    x = torch.randn(M, C, device=dev).to(BF); dy = torch.randn(M, C, device=dev).to(BF); dres = torch.randn(M, C, device=dev).to(BF)
    g = torch.ones(C, device=dev); b = torch.zeros(C, device=dev)
    y = torch.empty_like(x); mean = torch.empty(M, device=dev); rstd = torch.empty(M, device=dev); dx = torch.empty_like(x)
    dg = torch.empty(C, device=dev); db = torch.empty(C, device=dev)
    ws = torch.empty(lib.plain("fiber_layernorm_bwd_grid", M) * 8 * C, device=dev)
    P = lib.ptr
    by = M * C * 2
    us = t(lambda: lib.call("fiber_layernorm_fwd_bf16", P(x), P(g), P(b), P(y), P(mean), P(rstd), M, C, 1e-5))
    print(f"M={M} C={C}  ln_fwd {us:7.1f}us {2 * by / us / 1e3:6.0f} GB/s", end="")
    us = t(lambda: lib.call("fiber_layernorm_bwd_bf16", P(dy), P(x), P(g), P(mean), P(rstd), P(dres), P(dx), P(dg), P(db), P(ws), M, C))
    print(f" | ln_bwd(+dres) {us:7.1f}us {4 * by / us / 1e3:6.0f} GB/s", end="")
    h = torch.randn(M, 4 * C, device=dev).to(BF); dgl = torch.randn(M, 4 * C, device=dev).to(BF)
    us = t(lambda: ops.gelu_bwd_colsum(dgl, h))
    print(f" | gelu_bwd_colsum {us:7.1f}us {3 * 4 * by / us / 1e3:6.0f} GB/s", end="")
    us = t(lambda: ops.colsum(h))
    print(f" | colsum(4C) {us:7.1f}us {4 * by / us / 1e3:6.0f} GB/s", end="")
    sc = torch.ones(256, device=dev)
    out = torch.empty_like(x)
    us = t(lambda: lib.call("fiber_rowscale_add_bf16", None, P(x), P(sc), P(out), x.numel(), x.numel() // 256))
    print(f" | rowscale {us:6.1f}us {2 * by / us / 1e3:6.0f} GB/s")
