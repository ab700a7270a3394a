"""ops.wgrad (library TN GEMM split over M chunks + fp32 fold) at the step's shapes: us per call for the current FIBER_WGRAD_TARGET."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fiber_amd import lib, ops
lib.load()
def t(fn, n=10):
    for _ in range(2): fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
B = 512
tot = 0.0
for name, L, C, cnt in (("s0", 9216, 128, 2), ("s1", 2304, 256, 2), ("s2", 576, 512, 18), ("s3", 144, 1024, 2)):
    M = B * L
    for nm, N, K in (("qkv", 3 * C, C), ("proj", C, C), ("fc1", 4 * C, C), ("fc2", C, 4 * C)):
        dh = torch.randn(M, N, device="cuda").to(torch.bfloat16); x = torch.randn(M, K, device="cuda").to(torch.bfloat16)
        us = t(lambda: ops.wgrad(dh, x)); tot += us * cnt
        print(f"{name}.{nm:4s} M={M:8d} N={N:5d} K={K:5d}: {us:8.1f} us  {2.0 * M * N * K / us / 1e6:7.1f} TF")
        del dh, x
print(f"sum over the Swin blocks of one step: {tot / 1e3:.2f} ms (target {ops._WGRAD_TARGET})")
