import sys, os, torch
sys.path.insert(0, os.getcwd())
from fiber_amd import lib, ops
def t(fn, n=20):
    for _ in range(3): fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
for M, N in ((294912, 512), (1179648, 256), (4718592, 128)):
    dy = torch.randn(M, N, device="cuda").to(torch.bfloat16)
    rs = torch.rand(M // 576 if M % 576 == 0 else 512, device="cuda")
    us = t(lambda: ops.rowscale_colsum(dy, rs))
    print(f"rowscale_colsum M={M} N={N}: {us:8.1f} us  {2 * M * N * 2 / us / 1e3:7.0f} GB/s")
for M, N in ((1179648, 1024), (4718592, 512)):
    dg = torch.randn(M, N, device="cuda").to(torch.bfloat16); h = torch.randn(M, N, device="cuda").to(torch.bfloat16)
    us = t(lambda: ops.gelu_bwd_colsum(dg, h))
    print(f"gelu_bwd_colsum M={M} N={N}: {us:8.1f} us  {3 * M * N * 2 / us / 1e3:7.0f} GB/s")
