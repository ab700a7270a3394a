import os, sys
sys.path.insert(0, "/root/repo")
import torch
from fiber_amd import lib, ops
from tools.gemm_ab import timeit
lib.load()
for M, N, K in [(294912, 2048, 512), (1179648, 1024, 256), (73728, 4096, 1024)]:
    x = torch.randn(M, K, device="cuda").to(torch.bfloat16)
    w = (torch.randn(N, K, device="cuda") * K ** -0.5).to(torch.bfloat16)
    b = torch.randn(N, device="cuda")
    aux = torch.randn(M, N, device="cuda").to(torch.bfloat16)
    r = torch.randn(M, N, device="cuda").to(torch.bfloat16)
    t = {"bias": timeit(lambda: ops.gemm_nt(x, w, b), reps=20),
         "gelu only": timeit(lambda: ops.gemm_nt(x, w, b, None, 1, False), reps=20),
         "gelu+pre": timeit(lambda: ops.gemm_nt(x, w, b, None, 1, True), reps=20),
         "gelu'*aux": timeit(lambda: ops.gemm_nt(x, w, None, None, 2, False, None, 0, aux), reps=20),
         "residual": timeit(lambda: ops.gemm_nt(x, w, b, r), reps=20)}
    print(M, N, K, " | ".join(f"{k} {v:7.0f}us" for k, v in t.items()), flush=True)
