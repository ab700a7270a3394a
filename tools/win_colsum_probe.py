"""Window-attention backward with and without the dqkv column sums (bias gradient of the qkv linear) produced inside it."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fiber_amd import lib, ops
lib.load()
dev, bf = "cuda", torch.bfloat16
def t(fn, reps=20):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps
B = 512
for name, H, C, heads in (("s2", 24, 512, 16), ("s0", 96, 128, 4), ("s1", 48, 256, 8), ("s3", 12, 1024, 32)):
    for shift in (0, 6 if H > 12 else 0):
        qkv = torch.randn(B, H * H, 3 * C, device=dev).to(bf)
        tab = torch.randn(529, heads, device=dev)
        with torch.no_grad():
            o = ops.window_attention(qkv, tab, B, H, H, heads, 12, shift, head_major=True)
        # forward again through the raw entry to get lse
        rows, N = B * H * H, 144
        lse = torch.empty((rows, heads), dtype=torch.float32, device=dev); o2 = torch.empty_like(o)
        lib.call("fiber_window_attn_fwd_bf16", lib.ptr(qkv), lib.ptr(tab), lib.ptr(o2), lib.ptr(lse), B, H, H, C, heads, 12, shift, 1)
        do = torch.randn_like(o2)
        dqkv = torch.empty_like(qkv); dtab = torch.empty_like(tab); delta = torch.empty((rows, heads), dtype=torch.float32, device=dev)
        nz = lib.plain("fiber_window_attn_bwd_slices", rows // N, heads)
        part = torch.empty(nz * heads * N * N, dtype=torch.float32, device=dev)
        cs_rows = lib.plain("fiber_window_attn_colsum_rows", rows // N, heads, 12)
        csum = torch.empty(3 * C, dtype=torch.float32, device=dev); cs_ws = torch.empty(cs_rows * 3 * C, dtype=torch.float32, device=dev)
        def run(with_cs):
            lib.call("fiber_window_attn_bwd_bf16", lib.ptr(qkv), lib.ptr(tab), lib.ptr(o2), lib.ptr(do), lib.ptr(lse), lib.ptr(dqkv), lib.ptr(dtab), lib.ptr(delta),
                     lib.ptr(part), lib.ptr(csum) if with_cs else None, lib.ptr(cs_ws) if with_cs else None, B, H, H, C, heads, 12, shift, 1)
        for rep in range(2):
            print(f"{name} shift {shift}: backward with column sums {t(lambda: run(True)):8.1f} us | without {t(lambda: run(False)):8.1f} us", flush=True)
        del qkv, o, o2, do, dqkv, part
