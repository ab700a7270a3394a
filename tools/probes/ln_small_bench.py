"""LayerNorm forward / backward through the C ABI at the text-stack, stage-3 and stage-2 shapes: us per call and TB/s of the algorithmic bytes
(how the small-tensor launch geometry was chosen: profiles/r06_summary.md section 8).   python tools/probes/ln_small_bench.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from fiber_amd import lib, ops
lib.load()
for rows, C in ((20480, 768), (10240, 768), (73728, 1024), (294912, 512), (1179648, 256)):
    x = torch.randn(rows, C, device="cuda").to(torch.bfloat16).requires_grad_(True)
    g = torch.randn(C, device="cuda", requires_grad=True)
    b = torch.randn(C, device="cuda", requires_grad=True)
    dy = torch.randn(rows, C, device="cuda").to(torch.bfloat16)

    def fwd():
        return ops.layernorm(x, g, b, 1e-5)

    def time(fn, n=30):
        for _ in range(5):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) * 1e3 / n
    with torch.no_grad():
        tf = time(fwd)
    y = fwd()

    def bwd():
        x.grad = g.grad = b.grad = None
        y.backward(dy, retain_graph=True)
    tb = time(bwd)
    print(f"[{rows:8d}, {C:5d}]  fwd {tf:7.1f} us {4 * rows * C / tf / 1e6:5.2f} TB/s   bwd (+ reduce) {tb:7.1f} us {6 * rows * C / tb / 1e6:5.2f} TB/s")
