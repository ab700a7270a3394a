"""TN weight-gradient kernel with and without the DropPath row mask at the step's shapes (us, TFLOP/s on the FULL flop count), and a check of
the masked result against the unmasked kernel on pre-scaled rows.  A/B: FIBER_HIP_LIB=<other build> python tools/tn_mask_bench.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fiber_amd import lib, ops
lib.load()
dev = "cuda"
bf = torch.bfloat16


def t(fn, reps=20):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


torch.manual_seed(0)
for name, M, N, K, rps in (("s2.fc2", 294912, 512, 2048, 576), ("s2.proj", 294912, 512, 512, 576), ("s1.fc2", 1179648, 256, 1024, 2304), ("s0.fc2", 4718592, 128, 512, 9216),
                           ("many samples", 64 * 300, 512, 512, 64)):
    dy = torch.randn(M, N, device=dev).to(bf)
    x = torch.randn(M, K, device=dev).to(bf)
    keep = 0.95
    mask = (torch.rand(M // rps, device=dev) < keep).float() / keep
    us0 = t(lambda: ops.wgrad(dy, x, want_bias=True))
    us1 = t(lambda: ops.wgrad(dy, x, want_bias=True, row_mask=mask, scale=1 / keep))
    got, gb = ops.wgrad(dy, x, want_bias=True, row_mask=mask, scale=1 / keep)
    dys = (dy.float() * mask.repeat_interleave(rps)[:, None]).to(bf)      # exact: factors are 0 or 1/keep... rounding of dy/keep differs, so compare loosely
    want, wb = ops.wgrad(dys, x, want_bias=True)
    err = ((got - want).abs().max() / want.abs().max()).item()
    errb = ((gb - wb).abs().max() / wb.abs().max()).item()
    fl = 2.0 * M * N * K
    print(f"{name:14s} M={M:8d} N={N:5d} K={K:5d}: plain {us0:8.1f} us {fl / us0 / 1e6:6.0f} TF | masked {us1:8.1f} us {fl / us1 / 1e6:6.0f} TF | rel err vs pre-scaled rows {err:.2e} (bias {errb:.2e})", flush=True)
