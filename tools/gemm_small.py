"""NT GEMM timings at the shapes with 200..511 output tiles (text stack at B=256, stage 3, small batches): which kernel serves them
is FIBER_GEMM_PERSIST_MIN (tiles from which the persistent q8 kernel is used; default 512)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fiber_amd import lib, ops
from tools.gemm_ab import timeit
lib.load()
print("PERSIST_MIN", os.environ.get("FIBER_GEMM_PERSIST_MIN", "512"))
for M, N, K in [(20480, 768, 768), (20480, 3072, 768), (20480, 768, 3072), (20480, 2304, 768), (73728, 1024, 1024), (73728, 3072, 1024),
                (36864, 512, 512), (36864, 2048, 512), (36864, 1536, 512), (147456, 256, 256)]:
    x = torch.randn(M, K, device="cuda").to(torch.bfloat16)
    w = (torch.randn(N, K, device="cuda") * K ** -0.5).to(torch.bfloat16)
    b = torch.randn(N, device="cuda")
    t = timeit(lambda: ops.gemm_nt(x, w, b), reps=30)
    tl = timeit(lambda: torch.nn.functional.linear(x, w), reps=30)
    tiles = -(-M // 256) * -(-N // 256)
    print(f"M={M:7d} N={N:5d} K={K:5d} tiles={tiles:5d} | own {t:7.1f}us {2.0 * M * N * K / t / 1e6:6.0f}TF | lib {tl:7.1f}us", flush=True)
