"""Window attention forward / fused backward per tensor layout through the C ABI: 0 = reference [token][3][heads][32]; 3 = planar
[3 * heads planes][token][32] (o / dout: [heads planes][token][32]); the planar results are checked bit for bit against layout 0."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from fiber_amd import lib
lib.load()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
def timeit(fn, reps=10, warm=3):
    for _ in range(warm): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps
def to_planar(x, planes): return x.view(x.shape[0], planes, 32).permute(1, 0, 2).contiguous()
def from_planar(x, M): return x.view(-1, M, 32).permute(1, 0, 2).reshape(M, -1)
for name, H, C, heads in (("s0", 96, 128, 4), ("s1", 48, 256, 8), ("s2", 24, 512, 16), ("s3", 12, 1024, 32)):
    M = B * H * H
    N = 144
    qkv = torch.randn(M, 3 * C, device="cuda").to(torch.bfloat16)
    do = torch.randn(M, C, device="cuda").to(torch.bfloat16)
    tab = torch.randn(529, heads, device="cuda") * 0.5
    nz = lib.plain("fiber_window_attn_bwd_slices", M // N, heads)
    part = torch.empty(nz * heads * N * N, device="cuda")
    cs_rows = lib.plain("fiber_window_attn_colsum_rows", M // N, heads, 12)
    cs_ws = torch.empty(cs_rows * 3 * C, device="cuda")
    delta = torch.empty(M * heads, device="cuda")
    for shift in (0, 6):
        res, line = {}, []
        for hm in (0, 1, 2, 3, 4):
            src = qkv if hm < 3 else to_planar(qkv, 3 * heads)     # (layouts 1, 2: timing only, the same bytes read as another channel order)
            dsrc = do if hm != 3 else to_planar(do, heads)
            o = torch.empty(M * C, device="cuda", dtype=torch.bfloat16)
            lse = torch.empty(M * heads, device="cuda")
            dqkv = torch.empty(M * 3 * C, device="cuda", dtype=torch.bfloat16)
            dtab = torch.empty_like(tab); csum = torch.empty(3 * C, device="cuda")
            def f(): lib.call("fiber_window_attn_fwd_bf16", lib.ptr(src), lib.ptr(tab), lib.ptr(o), lib.ptr(lse), B, H, H, C, heads, 12, shift, hm)
            def b(): lib.call("fiber_window_attn_bwd_bf16", lib.ptr(src), lib.ptr(tab), lib.ptr(o), lib.ptr(dsrc), lib.ptr(lse), lib.ptr(dqkv), lib.ptr(dtab),
                              lib.ptr(delta), lib.ptr(part), lib.ptr(csum), lib.ptr(cs_ws), B, H, H, C, heads, 12, shift, hm)
            tf = timeit(f); tb = timeit(b)
            res[hm] = (o.clone(), lse.clone(), dqkv.clone(), dtab.clone(), csum.clone())
            line.append(f"L{hm}: {tf:6.1f} | {tb:6.1f}")
        ok = (torch.equal(from_planar(res[3][0], M), res[0][0].view(M, C)) and torch.equal(res[0][1], res[3][1])
              and torch.equal(from_planar(res[3][2], M), res[0][2].view(M, 3 * C)) and torch.equal(res[0][3], res[3][3]) and torch.equal(res[0][4], res[3][4])
              and torch.equal(res[4][0], res[0][0]) and torch.equal(res[4][2], res[0][2]) and torch.equal(res[4][3], res[0][3]))
        print(f"{name}/{shift}  " + "   ".join(line) + f"   bit-equal {ok}")
