import os, sys
sys.path.insert(0, os.getcwd())
import torch
from fiber_amd import lib, ops
lib.load()
def t(fn, reps=20):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps
for M, N, K in ((294912, 2048, 512), (294912, 2048, 512), (294912, 512, 2048), (294912, 1536, 512), (73728, 4096, 1024)):
    dy = torch.randn(M, N, device="cuda").to(torch.bfloat16); x = torch.randn(M, K, device="cuda").to(torch.bfloat16)
    for rep in range(2):
        a = t(lambda: ops.wgrad(dy, x, want_bias=True)); b = t(lambda: ops.wgrad(dy, x, want_bias=False))
        print(f"M={M} N={N} K={K}: with bias sums {a:7.1f} us {2.0*M*N*K/a/1e6:5.0f} TF | without (no VALU code) {b:7.1f} us {2.0*M*N*K/b/1e6:5.0f} TF", flush=True)
