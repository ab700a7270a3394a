"""One shape of the hand-written GEMM, 10 launches, for `rocprofv3 --pmc ...` passes.  argv: M N K mode(plain|gelu)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fiber_amd import lib, ops
lib.load()
M, N, K = (int(v) for v in sys.argv[1:4])
mode = sys.argv[4] if len(sys.argv) > 4 else "plain"
x = torch.randn(M, K, device="cuda").to(torch.bfloat16)
w = (torch.randn(N, K, device="cuda") * K ** -0.5).to(torch.bfloat16)
b = torch.randn(N, device="cuda")
for _ in range(10):
    if mode == "gelu":
        ops.gemm_nt(x, w, b, None, 1, True)
    else:
        ops.gemm_nt(x, w)
torch.cuda.synchronize()
