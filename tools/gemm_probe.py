"""Ablation of the GEMM main loop: act bit 0x100 skips the in-loop DMA (operands stay whatever the prologue staged), which
separates the MFMA + LDS-read ceiling from the L2->LDS DMA cost.  (Earlier builds also had no-MFMA / no-barrier switches;
their results are recorded in profiles/r01_summary.md.)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fiber_amd import lib, ops
dev = "cuda"
def timeit(fn, reps=20):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps
lib.load()
for M, N, K in ((18432, 512, 2048), (18432, 2048, 512), (4608, 4096, 1024), (73728, 2048, 512), (8192, 8192, 8192)):
    x = torch.randn(M, K, device=dev).to(torch.bfloat16)
    w = (torch.randn(N, K, device=dev) * K ** -0.5).to(torch.bfloat16)
    fl = 2.0 * M * N * K
    row = f"M={M} N={N} K={K}:"
    for name, flag in (("full", 0), ("noDMA", 0x100)):
        t = timeit(lambda: ops.gemm_nt(x, w, None, None, flag, False))
        row += f"  {name} {t:7.1f}us ({fl / t / 1e6:6.0f}TF)"
    t = timeit(lambda: torch.nn.functional.linear(x, w))
    row += f"  lib {t:7.1f}us ({fl / t / 1e6:6.0f}TF)"
    print(row)
