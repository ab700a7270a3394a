import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fiber_amd import lib, ops
lib.load()
B = 512
def timeit(fn, reps=10, warm=3):
    for _ in range(warm): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps
for hm in (0, 1, 2):
    out = []
    for name, H, C, heads in (("s0", 96, 128, 4), ("s2", 24, 512, 16)):
        qkv = torch.randn(B, H * H, 3 * C, device="cuda").to(torch.bfloat16).requires_grad_(True)
        tab = (torch.randn(529, heads, device="cuda") * 0.5).requires_grad_(True)
        do = torch.randn(B, H * H, C, device="cuda").to(torch.bfloat16)
        for shift in (0, 6):
            with torch.no_grad():
                tf = timeit(lambda: ops.window_attention(qkv, tab, B, H, H, heads, 12, shift, head_major=hm))
            def fb():
                ops.window_attention(qkv, tab, B, H, H, heads, 12, shift, head_major=hm).backward(do)
            tfb = timeit(fb)
            out.append(f"{name}/{shift}: fwd {tf:7.1f} bwd {tfb - tf:7.1f}")
    print("layout", hm, "  ".join(out))
