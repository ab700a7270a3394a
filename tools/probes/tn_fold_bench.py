"""The fold of a split weight-gradient GEMM alone (csrc/gemm_tn.hip tn_fold_item through fiber_tn_fold_multi with one descriptor): us per call and
TB/s over the slabs, cold (a 512-MB buffer written in between) and warm (slabs resident in the Infinity Cache), at the step's shapes.
    python tools/probes/tn_fold_bench.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from fiber_amd import lib
lib.load()
dev = "cuda"
SHAPES = [(2048, 512, 16), (512, 2048, 16), (1536, 512, 21), (512, 512, 64), (1024, 256, 64), (4096, 1024, 4), (384, 128, 170)]
flush = torch.empty(512 << 20, dtype=torch.uint8, device=dev)
for N, K, S in SHAPES:
    ws = torch.randn(S * (N * K + N), device=dev)
    out = torch.empty(N * K, device=dev)
    cs = torch.empty(N, device=dev)
    nk4 = N * K // 4
    blocks = lib.plain("fiber_tn_fold_blocks", S, N, K, 1)
    table = torch.tensor([(ws.data_ptr(), out.data_ptr(), cs.data_ptr(), S | (N << 32), nk4 | (0 << 32))], dtype=torch.int64).to(dev)
    ref = ws[:S * N * K].view(S, -1).sum(0)
    lib.call("fiber_tn_fold_multi", lib.ptr(table), 1, blocks)
    torch.cuda.synchronize()
    err = max((out - ref).abs().max().item(), (cs - ws[S * N * K:].view(S, N).sum(0)).abs().max().item())
    res = []
    for cold in (False, True):
        ts = []
        for _ in range(12):
            if cold:
                flush.fill_(1)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            lib.call("fiber_tn_fold_multi", lib.ptr(table), 1, blocks)
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) * 1e3)
        ts.sort()
        res.append(ts[len(ts) // 2])
    by = (S + 1) * N * K * 4
    print(f"N={N:5d} K={K:5d} S={S:3d}  {by / 1e6:6.1f} MB  warm {res[0]:6.1f} us {by / res[0] / 1e6:5.2f} TB/s   cold {res[1]:6.1f} us {by / res[1] / 1e6:5.2f} TB/s   max err {err:.1e}")
