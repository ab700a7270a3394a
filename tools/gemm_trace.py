"""Per-segment s_memtime trace of the wide GEMM K loop (TRACE build, act bit 0x200): cycles per K tile spent in
load issue / wait+barrier / MFMA issue / wait+barrier, for the first 8 workgroups x 8 waves."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fiber_amd import lib
lib.load()
M, N, K = (int(v) for v in sys.argv[1:4]) if len(sys.argv) > 3 else (73728, 1024, 4096)
x = torch.randn(M, K, device="cuda").to(torch.bfloat16)
w = (torch.randn(N, K, device="cuda") * K ** -0.5).to(torch.bfloat16)
y = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
dbg = torch.zeros(8 * 8 * 16, device="cuda")
for _ in range(3):
    lib.call("fiber_gemm_nt_bf16", lib.ptr(x), lib.ptr(w), None, None, lib.ptr(y), None, None, 0, None, 0, lib.ptr(dbg),
             M, N, K, K, K, N, 0, 0x200)
torch.cuda.synchronize()
d = dbg.view(8, 8, 16).cpu()
nk = d[0, 0, 9].item()
print(f"nk={nk:.0f}  (s_memtime ticks per K tile; 100 MHz-class constant clock -> compare ratios)")
for wg in range(0, 8, 3):
    for wv in range(8):
        r = d[wg, wv]
        seg = " | ".join(f"{n} {r[i] / nk:6.1f}" for i, n in enumerate(("L0+dma", "bar", "M0", "bar", "L1", "bar", "M1", "bar")))
        print(f"wg{wg} wave{wv} (group {wv // 4}): {seg} | total/ktile {r[8] / nk:7.1f}")
