"""Ablation timing of csrc/mlp_rows.hip: the same forward / backward kernel call under builds with one ingredient compiled out
(tools/bin/libfiber_mlp_<variant>.so, built with -DMLP_PROBE_<variant>; results are wrong by construction, only the clock counts).
    FIBER_HIP_LIB=tools/bin/libfiber_mlp_NOGELU.so python tools/lnmlp_probe.py [images]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fiber_amd import lib, ops

dev, BF = "cuda", torch.bfloat16


def timeit(fn, reps=10, warm=3):
    for _ in range(warm):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
lib.load()
out = [os.path.basename(lib.LIB_PATH)]
for name, L, C in (("s0", 9216, 128), ("s1", 2304, 256)):
    H, M = 4 * C, B * L
    x = torch.randn(M, C, device=dev).to(BF)
    dy = torch.randn(M, C, device=dev).to(BF)
    w1 = (torch.randn(H, C, device=dev) * C ** -0.5).to(BF)
    w2 = (torch.randn(C, H, device=dev) * H ** -0.5).to(BF)
    w1t, w2t = w1.t().contiguous(), w2.t().contiguous()
    b1, b2 = torch.zeros(H, device=dev), torch.zeros(C, device=dev)
    y, dx, xh = torch.empty_like(x), torch.empty_like(x), torch.empty_like(x)
    dh = torch.empty(M, H, device=dev, dtype=BF)
    g = torch.empty_like(dh)
    tf = timeit(lambda: lib.call("fiber_ln_mlp_fwd_bf16", lib.ptr(x), lib.ptr(w1), lib.ptr(b1), lib.ptr(w2), lib.ptr(b2), None, lib.ptr(y), lib.ptr(g), M, C, 0, 1e-5))
    tb = timeit(lambda: lib.call("fiber_ln_mlp_bwd_bf16", lib.ptr(x), lib.ptr(dy), lib.ptr(w1), lib.ptr(b1), lib.ptr(w2t), lib.ptr(w1t), None,
                                 lib.ptr(dx), lib.ptr(dh), lib.ptr(xh), M, C, 0, 1e-5))
    out.append(f"{name}: fwd {tf:7.1f} us  bwd kernel {tb:7.1f} us")
print(" | ".join(out), flush=True)
