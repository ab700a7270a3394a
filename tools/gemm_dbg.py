"""Ablation timing of the wide GEMM main loop: FIBER_GEMM_DBG=0|1|2|4|8|5|13 (bits: 1 no DMA, 2 no MFMA, 4 no ds_read, 8 no barrier)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fiber_amd import lib, ops
from tools.gemm_ab import timeit
lib.load()
BF = torch.bfloat16
out = [f"DBG={os.environ.get('FIBER_GEMM_DBG', '0')}"]
for M, N, K in ((294912, 2048, 512), (294912, 512, 2048), (73728, 1024, 4096)):
    x = torch.randn(M, K, device="cuda").to(BF)
    w = (torch.randn(N, K, device="cuda") * K ** -0.5).to(BF)
    t = timeit(lambda: ops.gemm_nt(x, w))
    out.append(f"{N}x{K}: {t:6.0f}us {2.0 * M * N * K / t / 1e6:5.0f}TF")
print(" | ".join(out))
