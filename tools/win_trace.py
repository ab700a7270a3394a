"""Per-segment cycle trace of the one-pass window-attention backward (the kernel writes its s_memtime sums into the otherwise unused
delta workspace).  Build the TRACE library next to the product one and point the loader at it:
    cd fiber_amd/csrc && hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DFIBER_WIN_TRACE -c win_attn.hip -o /tmp/win_trace.o && \
      hipcc --offload-arch=gfx950 -shared -fPIC -o ../../tools/bin/libfiber_hip_wtrace.so gemm.o gemm_tn.o input.o dcn.o norm.o attn.o \
      /tmp/win_trace.o elementwise.o embed.o optim.o loss.o
    FIBER_HIP_LIB=tools/bin/libfiber_hip_wtrace.so python tools/win_trace.py [shift]
The forward kernel of the same build keeps its sums in a device symbol (read through fiber_win_trace_read)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fiber_amd import lib
lib.load()
B, H, C, heads, ws = 512, 24, 512, 16, 12
shift = int(sys.argv[1]) if len(sys.argv) > 1 else 0
NWV = int(os.environ.get("FIBER_WIN_BWD_WAVES", "9"))
WAVES = list(range(NWV)) if len(sys.argv) > 2 and sys.argv[2] == "all" else (0, 4, 8)
WNAME = f"0-{NWV - 1}" if len(WAVES) == NWV else "0 / 4 / 8"
rows, N = B * H * H, ws * ws
dev = "cuda"
qkv = torch.randn(rows, 3 * C, device=dev).to(torch.bfloat16)
tab = torch.randn(529, heads, device=dev) * 0.5
o = torch.empty(rows, C, device=dev, dtype=torch.bfloat16)
lse = torch.empty(rows, heads, device=dev)
lib.call("fiber_window_attn_fwd_bf16", lib.ptr(qkv), lib.ptr(tab), lib.ptr(o), lib.ptr(lse), B, H, H, C, heads, ws, shift, 1)
do = torch.randn(rows, C, device=dev).to(torch.bfloat16)
dqkv = torch.empty_like(qkv)
dtab = torch.empty_like(tab)
delta = torch.zeros(rows, heads, device=dev)
nz = lib.plain("fiber_window_attn_bwd_slices", rows // N, heads)
part = torch.empty(nz * heads * N * N, device=dev)
cs_rows = lib.plain("fiber_window_attn_colsum_rows", rows // N, heads, ws)
csum = torch.empty(3 * C, device=dev)
cs_ws = torch.empty(cs_rows * 3 * C, device=dev)
for _ in range(3):
    lib.call("fiber_window_attn_bwd_bf16", lib.ptr(qkv), lib.ptr(tab), lib.ptr(o), lib.ptr(do), lib.ptr(lse), lib.ptr(dqkv),
             lib.ptr(dtab), lib.ptr(delta), lib.ptr(part), lib.ptr(csum), lib.ptr(cs_ws), B, H, H, C, heads, ws, shift, 1)
torch.cuda.synchronize()
d = delta.flatten().cpu()
names = ["top barrier", "prefetch landed (vmcnt 0)", "staging + delta", "barrier 2", "dk/dv stores + colsum", "barrier 3", "phase 2 + dq store + colsum", "kf read + next prefetch issued", "phase 1 body"]
for blk in range(2):
    nwin = float(d[1000 + blk])
    print(f"workgroup {blk}: {nwin:.0f} windows; ticks per window and segment (s_memtime), waves {WNAME}")
    for i in range(9):
        print(f"  {names[i]:52s}", " ".join(f"{float(d[(blk * NWV + w) * 12 + i]) / nwin:8.0f}" for w in WAVES))
    print("  total", " ".join(f"{sum(float(d[(blk * NWV + w) * 12 + i]) for i in range(9)) / nwin:8.0f}" for w in WAVES))

# forward kernel: the TRACE build keeps its sums in a device symbol
import ctypes as C
L = C.CDLL(lib.LIB_PATH)
if hasattr(L, "fiber_win_trace_read"):
    buf = (C.c_float * 256)()
    L.fiber_win_trace_read(buf)
    fn = ["top barrier", "prefetch landed (vmcnt 0)", "staging", "barrier 2", "next prefetch issued", "QK^T, softmax, PV", "store o / lse"]
    for blk in range(2):
        nwin = buf[200 + blk]
        if nwin:
            print(f"forward workgroup {blk}: {nwin:.0f} windows; ticks per window and segment, waves {WNAME}")
            for i in range(7):
                print(f"  {fn[i]:30s}", " ".join(f"{buf[(blk * 10 + w) * 8 + i] / nwin:8.0f}" for w in WAVES))
            print("  total", " ".join(f"{sum(buf[(blk * 10 + w) * 8 + i] for i in range(7)) / nwin:8.0f}" for w in WAVES))
