"""ops.stream_add forward / backward at the fused stage-2 block's shape ([B*576, 512]: shortcut + DropPath * (a + alpha * b)), us and TB/s of the
algorithmic bytes (forward: 3 reads + 1 write; backward: dy and b read, da and db written)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from fiber_amd import lib, ops
lib.load()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
def timeit(fn, reps=10, warm=3):
    for _ in range(warm): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps
for L, C in ((576, 512), (144, 1024)):
    res = torch.randn(B, L, C, device="cuda").to(torch.bfloat16)
    a = torch.randn(B, L, C, device="cuda").to(torch.bfloat16).requires_grad_(True)
    b = torch.randn(B, L, C, device="cuda").to(torch.bfloat16).requires_grad_(True)
    alpha = torch.full((1,), 0.5, device="cuda", requires_grad=True)
    rs = (torch.rand(B, device="cuda") > 0.1).float() / 0.9
    dy = torch.randn(B, L, C, device="cuda").to(torch.bfloat16)
    with torch.no_grad():
        tf = timeit(lambda: ops.stream_add(res, a, b=b, alpha=alpha, rowscale=rs))
    def fb():
        ops.stream_add(res, a, b=b, alpha=alpha, rowscale=rs).backward(dy)
        a.grad = b.grad = alpha.grad = None
    tfb = timeit(fb)
    n = B * L * C * 2
    print(f"[{B * L}, {C}]: fwd {tf:7.1f} us {4 * n / tf / 1e6:5.2f} TB/s   bwd ~{tfb - tf:7.1f} us {4 * n / (tfb - tf) / 1e6:5.2f} TB/s")
