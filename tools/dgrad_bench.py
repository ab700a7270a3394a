"""Library dgrad dX = dY . W as NN (W [N,K] as stored) vs NT (F.linear with the transposed copy W^T [K,N]) at the step's shapes."""
import sys, os, torch
import torch.nn.functional as F
def t(fn, n=10):
    for _ in range(2): fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
B = 512
tn = tt = 0.0
for name, L, C, cnt in (("s0", 9216, 128, 2), ("s1", 2304, 256, 2), ("s2", 576, 512, 18), ("s3", 144, 1024, 2)):
    M = B * L
    for nm, N, K in (("qkv", 3 * C, C), ("proj", C, C), ("fc1", 4 * C, C), ("fc2", C, 4 * C)):
        dy = torch.randn(M, N, device="cuda").to(torch.bfloat16)
        w = torch.randn(N, K, device="cuda").to(torch.bfloat16); wt = w.t().contiguous()
        a = t(lambda: torch.matmul(dy, w)); b = t(lambda: F.linear(dy, wt))
        tn += a * cnt; tt += b * cnt
        print(f"{name}.{nm:4s} M={M:8d} N={N:5d} K={K:5d}: NN {a:8.1f} us | NT {b:8.1f} us  ({100 * (b / a - 1):+5.1f} %)")
        del dy, w, wt
print(f"per step: NN {tn / 1e3:.2f} ms, NT {tt / 1e3:.2f} ms")
