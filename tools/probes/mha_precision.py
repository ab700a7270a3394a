"""rel-L2 error of ops.mha (forward, dq, dk, dv) against an fp64 reference at small text / cross-attention shapes; run once per kernel
selection (FIBER_ATTN_T2I_ONEPASS / FIBER_ATTN_I2T_ONEPASS = 0 | 1) to compare the one-pass kernels of attn_x.hip with the generic passes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from fiber_amd import lib, ops
lib.load()
def rel(a, b): return float((a.double() - b.double()).norm() / b.double().norm())
for (B, heads, Lq, Lk, D, masked) in ((2, 2, 12, 36, 64, False), (2, 2, 12, 9, 64, False), (2, 2, 12, 12, 64, True), (8, 12, 40, 576, 64, False),
                                      (8, 12, 40, 40, 64, True), (2, 4, 48, 12, 32, True), (8, 16, 576, 40, 32, True)):
    C = heads * D
    errs = []
    for seed in range(4):
        g = torch.Generator(device="cuda").manual_seed(seed)
        q, k, v = (torch.randn(B * L, C, device="cuda", generator=g).to(torch.bfloat16).requires_grad_(True) for L in (Lq, Lk, Lk))
        do = torch.randn(B * Lq, C, device="cuda", generator=g).to(torch.bfloat16)
        km = None
        if masked:
            lens = torch.randint(max(1, Lk // 3), Lk + 1, (B,), device="cuda", generator=g); lens[0] = Lk
            km = (torch.arange(Lk, device="cuda")[None] >= lens[:, None]).float() * -10000.0
        o = ops.mha(q, k, v, km, B, heads, D ** -0.5)
        o.backward(do)
        qr, kr, vr = (t.detach().double().view(B, L, heads, D).permute(0, 2, 1, 3).requires_grad_(True) for t, L in ((q, Lq), (k, Lk), (v, Lk)))
        s = qr @ kr.transpose(-1, -2) * D ** -0.5
        if km is not None: s = s + km.double()[:, None, None, :]
        orf = (torch.softmax(s, -1) @ vr).permute(0, 2, 1, 3).reshape(B * Lq, C)
        orf.backward(do.double())
        back = lambda t, L: t.permute(0, 2, 1, 3).reshape(B * L, C)
        errs.append((rel(o, orf), rel(q.grad, back(qr.grad, Lq)), rel(k.grad, back(kr.grad, Lk)), rel(v.grad, back(vr.grad, Lk))))
    e = torch.tensor(errs).mean(0)
    print(f"B{B} h{heads} {Lq}x{Lk} d{D} masked={masked}:  o {e[0]:.2e}  dq {e[1]:.2e}  dk {e[2]:.2e}  dv {e[3]:.2e}")
