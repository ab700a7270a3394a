"""FETCH_SIZE calibration for the GEMM's DMA access pattern: N = 128 (one N tile) => X is read exactly once."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fiber_amd import lib, ops
lib.load()
for M, N, K in ((589824, 128, 512), (73728, 2048, 512)):
    x = torch.randn(M, K, device="cuda").to(torch.bfloat16)
    w = (torch.randn(N, K, device="cuda") * K ** -0.5).to(torch.bfloat16)
    for _ in range(5):
        ops.gemm_nt(x, w, None, None, 0, False)
    torch.cuda.synchronize()
    print(M, N, K, "X bytes", M * K * 2, "W bytes", N * K * 2, "Y bytes", M * N * 2)
