import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fiber_amd import lib, ops
lib.load()
B, H, C, heads = 32, 96, 128, 4
qkv = torch.randn(B, H * H, 3 * C, device="cuda").to(torch.bfloat16)
tab = torch.randn(529, heads, device="cuda") * 0.5
def t(fn, reps=20):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps
with torch.no_grad():
    print(os.environ.get("FIBER_WIN_DBG", "0"), "fwd us:", t(lambda: ops.window_attention(qkv, tab, B, H, H, heads, 12, 0, head_major=True)))
