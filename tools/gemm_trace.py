"""Per-segment s_memtime trace of the wide GEMM K loop (TRACE build, act bit 0x200): cycles per K tile spent in
load issue / wait+barrier / MFMA issue / wait+barrier, for the first 8 workgroups x 8 waves."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fiber_amd import lib
lib.load()

def wide():
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


def persist(M, N, K, act=0):
    """Per-wave totals of the persistent kernel (TRACE build, act bit 0x400)."""
    x = torch.randn(M, K, device="cuda").to(torch.bfloat16)
    w = (torch.randn(N, K, device="cuda") * K ** -0.5).to(torch.bfloat16)
    b = torch.randn(N, device="cuda")
    y = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    pre = torch.empty(M, N, device="cuda", dtype=torch.bfloat16) if act == 1 else None
    dbg = torch.zeros(8 * 8 * 16, device="cuda")
    for _ in range(3):
        lib.call("fiber_gemm_nt_bf16", lib.ptr(x), lib.ptr(w), lib.ptr(b), None, lib.ptr(y), lib.ptr(pre), None, 0, None, 0, lib.ptr(dbg),
                 M, N, K, K, K, N, 0, 0x400 | act)
    torch.cuda.synchronize()
    d = dbg.view(8, 8, 16).cpu()
    nk, T = d[0, 0, 9].item(), d[0, 0, 10].item()
    print(f"persist M={M} N={N} K={K} act={act}: nk={nk:.0f} tiles/wg={T:.0f}  (10-ns ticks PER TILE)")
    for wg in (0, 5):
        for wv in (0, 3, 4, 7):
            r = d[wg, wv] / T
            print(f"wg{wg} wave{wv} g{wv // 4}: kt0 {r[0]:6.1f} | other kt {r[1]:7.1f} ({r[1] / max(nk - 1, 1):5.1f} each) | epilogue {r[2]:6.1f} | "
                  f"barriers {r[3]:6.1f} {r[4]:6.1f} {r[5]:6.1f} {r[6]:6.1f} | load+dma issue {r[7]:6.1f} | tile total {r[8]:7.1f}")


def q8(M, N, K, act=0):
    """Per-wave, per-phase totals of the v5 (q8) kernel (TRACE build, act bit 0x1000): cycles per K tile."""
    x = torch.randn(M, K, device="cuda").to(torch.bfloat16)
    w = (torch.randn(N, K, device="cuda") * K ** -0.5).to(torch.bfloat16)
    b = torch.randn(N, device="cuda")
    y = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    pre = torch.empty(M, N, device="cuda", dtype=torch.bfloat16) if act == 1 else None
    dbg = torch.zeros(8 * 8 * 24, device="cuda")
    for _ in range(3):
        lib.call("fiber_gemm_nt_bf16", lib.ptr(x), lib.ptr(w), lib.ptr(b), None, lib.ptr(y), lib.ptr(pre), None, 0, None, 0, lib.ptr(dbg),
                 M, N, K, K, K, N, 0, 0x1000 | act)
    torch.cuda.synchronize()
    d = dbg.view(8, 8, 24).cpu()
    nk, T = d[0, 0, 18].item(), d[0, 0, 19].item()
    print(f"q8 M={M} N={N} K={K} act={act}: nk={nk:.0f} tiles/wg={T:.0f}  (s_memtime ticks per K TILE; per phase: load half incl. read return / wait+barrier / MFMA issue / barrier)")
    for wg in (0, 5):
        for wv in (0, 3, 4, 7):
            r = d[wg, wv] / (T * nk)
            ph = " | ".join(f"P{p + 1} {r[4 * p]:5.0f} {r[4 * p + 1]:5.0f} {r[4 * p + 2]:5.0f} {r[4 * p + 3]:5.0f}" for p in range(4))
            print(f"wg{wg} wave{wv} g{wv // 4}: {ph} | epilogue/tile {d[wg, wv, 16] / T:7.0f} | total/Ktile {d[wg, wv, 17] / (T * nk):6.0f}")


if len(sys.argv) > 1 and sys.argv[1] == "q8":
    for shp in ((294912, 2048, 512), (294912, 512, 2048), (73728, 1024, 4096)):
        q8(*shp, 0)
    q8(294912, 2048, 512, 1)
elif len(sys.argv) > 1 and sys.argv[1] == "persist":
    for shp in ((294912, 2048, 512), (294912, 512, 2048), (73728, 1024, 4096)):
        persist(*shp, 0)
    persist(294912, 2048, 512, 1)
else:
    wide()
