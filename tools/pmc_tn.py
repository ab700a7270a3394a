"""Launch pattern for the PMC passes on the weight-gradient kernel (csrc/gemm_tn.hip) at the stage-2 fc1 shape of the bench
batch: dW[2048,512] = dH[M,2048]^T . X[M,512] + bias sums, M = 2*B*576, 20 launches."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fiber_amd import lib, ops
lib.load()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
M, N, K = 2 * B * 576, 2048, 512
dh = torch.randn(M, N, device="cuda").to(torch.bfloat16)
x = torch.randn(M, K, device="cuda").to(torch.bfloat16)
for _ in range(20):
    ops.wgrad(dh, x, want_bias=True)
torch.cuda.synchronize()
S = lib.plain("fiber_gemm_tn_splits", M, N, K)
print("algorithmic bytes per launch:", 2 * (M * N + M * K) + 4 * N * K, "slab bytes (written + folded):", 2 * 4 * S * N * K, "flops:", 2 * M * N * K)
