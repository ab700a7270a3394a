import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fiber_amd import lib, ops
lib.load()
def t(fn, reps=20):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps
for M, N, K in ((147456, 2048, 512), (589824 * 2, 512, 128)):
    x = torch.randn(M, K, device="cuda").to(torch.bfloat16)
    w = (torch.randn(N, K, device="cuda") * K ** -0.5).to(torch.bfloat16)
    b = torch.randn(N, device="cuda")
    r = torch.randn(M, N, device="cuda").to(torch.bfloat16)
    print(f"M={M} N={N} K={K}: plain {t(lambda: ops.gemm_nt(x, w, None, None, 0, False)):.0f}us | bias {t(lambda: ops.gemm_nt(x, w, b, None, 0, False)):.0f} | "
          f"bias+gelu (no pre store) {t(lambda: ops.gemm_nt(x, w, b, None, 1, False)):.0f} | bias+gelu+pre {t(lambda: ops.gemm_nt(x, w, b, None, 1, True)):.0f} | "
          f"bias+residual {t(lambda: ops.gemm_nt(x, w, b, r, 0, False)):.0f}")
