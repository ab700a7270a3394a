"""One launch pattern for the PMC pass: the dominant hand-written kernel (256x128 MFMA GEMM, stage-2 fc1 shape at the
bench batch) run 20 times.  Used under `rocprofv3 --pmc FETCH_SIZE` and `--pmc WRITE_SIZE` (separate passes)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fiber_amd import lib, ops
lib.load()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
M, N, K = 2 * B * 576, 2048, 512          # one fused 2B-sample pass
x = torch.randn(M, K, device="cuda").to(torch.bfloat16)
w = (torch.randn(N, K, device="cuda") * K ** -0.5).to(torch.bfloat16)
bias = torch.randn(N, device="cuda")
for _ in range(20):
    ops.gemm_nt(x, w, bias, None, 1, True)
torch.cuda.synchronize()
print("algorithmic bytes per launch:", 2 * (M * K + N * K + 2 * M * N), "flops:", 2 * M * N * K)
