"""Debug aid: window-attention backward (dq / dk / dv / dbias / colsum) against the torch reference of tests/test_hip_ops.py."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fiber_amd import lib, ops
from tests.test_hip_ops import _window_ref, rnd
from tests.hip_util import bf, DEV
lib.load()


def rel(a, b):
    return float((a.float() - b.float()).norm() / max(float(b.float().norm()), 1e-9))


for (B, H, W, heads, ws, shift) in [(2, 12, 12, 2, 12, 0), (1, 24, 24, 4, 12, 6), (1, 24, 24, 4, 12, 0), (4, 48, 48, 4, 12, 6)]:
    C = heads * 32
    qkv = bf(rnd(B, H * W, 3 * C)).requires_grad_(True)
    table = rnd((2 * ws - 1) ** 2, heads, std=0.5).to(DEV).requires_grad_(True)
    o = ops.window_attention(qkv, table, B, H, W, heads, ws, shift)
    qr = qkv.detach().float().requires_grad_(True)
    tr = table.detach().clone().requires_grad_(True)
    oref = _window_ref(qr, tr, B, H, W, heads, ws, shift)
    do = bf(rnd(B, H * W, C, seed=5))
    o.backward(do)
    oref.backward(do.float())
    g, gr = qkv.grad, qr.grad
    print((B, H, W, heads, ws, shift), "o", rel(o, oref), "dq", rel(g[..., :C], gr[..., :C]), "dk", rel(g[..., C:2 * C], gr[..., C:2 * C]),
          "dv", rel(g[..., 2 * C:], gr[..., 2 * C:]), "dbias", rel(table.grad, tr.grad), flush=True)
