"""What does each OPTIONAL feature of a hot kernel cost?  Times the step's calls with the feature on and off through the public ops
(round 4: the TN kernel's bias column sums had been compiled into selects that every wave executed -- 10-12 % of the kernel -- and only an
on / off timing showed it).  us per call, warm (3 + 20 launches, repeated twice)."""
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


def row(name, cases):
    for rep in range(2):
        print(f"{name:34s} " + " | ".join(f"{k} {t(f):8.1f} us" for k, f in cases), flush=True)


B = 512
torch.manual_seed(0)
# warm the clocks
x = torch.randn(294912, 512, device=dev).to(bf); w = torch.randn(2048, 512, device=dev).to(bf)
for _ in range(30): ops.gemm_nt(x, w)
# ---- NT GEMM: bias seed, stage-2 qkv / fc2 shapes
for M, N, K in ((294912, 1536, 512), (294912, 512, 2048), (294912, 512, 512)):
    x = torch.randn(M, K, device=dev).to(bf); w = (torch.randn(N, K, device=dev) * K ** -0.5).to(bf); b = torch.randn(N, device=dev)
    r = torch.randn(M, N, device=dev).to(bf); rs = torch.ones(M // 576, device=dev)
    row(f"NT {M}x{N}x{K}", (("plain", lambda: ops.gemm_nt(x, w)), ("bias", lambda: ops.gemm_nt(x, w, b)), ("bias+res", lambda: ops.gemm_nt(x, w, b, r)),
                             ("bias+res+droppath", lambda: ops.gemm_nt(x, w, b, r, 0, False, rs, 576))))
    del x, w, r
# ---- window attention forward / backward: shift
for name, H, C, heads in (("s2", 24, 512, 16), ("s0", 96, 128, 4)):
    qkv = torch.randn(B, H * H, 3 * C, device=dev).to(bf).requires_grad_(True)
    tab = torch.randn(529, heads, device=dev, requires_grad=True)
    def fb(shift):
        o = ops.window_attention(qkv, tab, B, H, H, heads, 12, shift, head_major=True)
        o.backward(o)
    with torch.no_grad():
        row(f"window fwd {name}", (("shift 0", lambda: ops.window_attention(qkv, tab, B, H, H, heads, 12, 0, head_major=True)),
                                   ("shift 6", lambda: ops.window_attention(qkv, tab, B, H, H, heads, 12, 6, head_major=True))))
    row(f"window fwd+bwd {name}", (("shift 0", lambda: fb(0)), ("shift 6", lambda: fb(6))))
    del qkv
# ---- cross attention t2i (40 queries x 576 keys, d 64, 12 heads): key mask, dropout
q = torch.randn(B * 40, 768, device=dev).to(bf).requires_grad_(True)
kv = torch.randn(B * 576, 1536, device=dev).to(bf).requires_grad_(True)
km = torch.zeros(B, 576, device=dev)
def t2i(mask, p):
    o = ops.mha_kv_packed(q, kv, km if mask else None, B, 12, 0.125, p, 1234)
    o.backward(o)
row("t2i fwd+bwd", (("plain", lambda: t2i(False, 0.0)), ("mask", lambda: t2i(True, 0.0)), ("dropout 0.1", lambda: t2i(False, 0.1)), ("mask+dropout", lambda: t2i(True, 0.1))))
# ---- i2t (576 queries x 40 keys, d 32, 16 heads)
q2 = torch.randn(B * 576, 512, device=dev).to(bf).requires_grad_(True)
k2 = torch.randn(B * 40, 512, device=dev).to(bf).requires_grad_(True)
v2 = torch.randn(B * 40, 512, device=dev).to(bf).requires_grad_(True)
km2 = torch.zeros(B, 40, device=dev)
def i2t(mask):
    o = ops.mha(q2, k2, v2, km2 if mask else None, B, 16, 32 ** -0.5)
    o.backward(o)
row("i2t fwd+bwd", (("plain", lambda: i2t(False)), ("mask", lambda: i2t(True))))
# ---- LayerNorm with / without the residual-gradient input
xx = torch.randn(294912, 512, device=dev).to(bf).requires_grad_(True)
g = torch.ones(512, device=dev, requires_grad=True); bb = torch.zeros(512, device=dev, requires_grad=True)
def ln(res):
    if res:
        y, r = ops.layernorm_res(xx, g, bb, 1e-5)
        (y, r)
        torch.autograd.backward([y, r], [y, r])
    else:
        y = ops.layernorm(xx, g, bb, 1e-5)
        y.backward(y)
row("LayerNorm fwd+bwd 294912x512", (("plain", lambda: ln(False)), ("+ residual gradient", lambda: ln(True))))
