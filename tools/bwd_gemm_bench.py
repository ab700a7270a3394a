"""Backward GEMMs of the step's linears: hand-written TN weight gradient (csrc/gemm_tn.hip, + bias sums) and NT dgrad vs the
library forms they replaced (TN GEMM split into a batch of M chunks + fp32 fold; F.linear with the transposed weight)."""
import sys, os, torch
import torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fiber_amd import lib, ops
lib.load()
def t(fn, n=10):
    for _ in range(2): fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
def lib_wgrad(dh, x2, target=768):
    M, N = dh.shape; K = x2.shape[1]
    tiles = ((N + 127) // 128) * ((K + 127) // 128); S = 1
    while S < 64 and tiles * S < target and M % (2 * S) == 0 and M // (2 * S) >= 1024: S *= 2
    if S == 1: return torch.matmul(dh.t(), x2).float()
    return torch.bmm(dh.view(S, M // S, N).transpose(1, 2), x2.view(S, M // S, K), out_dtype=torch.float32).sum(0)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
tot = {"own_w": 0.0, "lib_w": 0.0, "own_d": 0.0, "lib_d": 0.0}
shapes = []
for name, L, C, cnt in (("s0", 9216, 128, 2), ("s1", 2304, 256, 2), ("s2", 576, 512, 18), ("s3", 144, 1024, 2)):
    for nm, N, K in (("qkv", 3 * C, C), ("proj", C, C), ("fc1", 4 * C, C), ("fc2", C, 4 * C)):
        shapes.append((f"{name}.{nm}", B * L, N, K, cnt))
shapes += [("txt.qkv", B * 40, 768, 768, 48), ("txt.fc1", B * 40, 3072, 768, 12), ("txt.fc2", B * 40, 768, 3072, 12)]
for name, M, N, K, cnt in shapes:
    dh = torch.randn(M, N, device="cuda").to(torch.bfloat16); x = torch.randn(M, K, device="cuda").to(torch.bfloat16)
    w = torch.randn(N, K, device="cuda").to(torch.bfloat16); wt = w.t().contiguous()
    a = t(lambda: ops.wgrad(dh, x, want_bias=True)); b = t(lambda: lib_wgrad(dh, x))
    c = t(lambda: ops.gemm_nt(dh, wt)); d = t(lambda: F.linear(dh, wt))
    S = lib.plain("fiber_gemm_tn_splits", M, N, K)
    tf = lambda us: 2.0 * M * N * K / us / 1e6
    print(f"{name:8s} M={M:8d} N={N:5d} K={K:5d} S={S:3d}: wgrad own {a:8.1f} us {tf(a):6.0f} TF | lib {b:8.1f} us {tf(b):6.0f} TF || dgrad own {c:8.1f} us {tf(c):6.0f} TF | lib {d:8.1f} us {tf(d):6.0f} TF", flush=True)
    for k, v in zip(("own_w", "lib_w", "own_d", "lib_d"), (a, b, c, d)): tot[k] += v * cnt
    del dh, x, w, wt
print("per step (ms):", {k: round(v / 1e3, 2) for k, v in tot.items()})
