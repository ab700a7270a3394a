"""GPU parity: each HIP kernel (called through the C ABI via fiber_amd.ops / fiber_amd.lib) against a plain PyTorch
fp32 reference of the same op on identical bf16-rounded inputs.  Tolerances are rel-L2 and written per test:
bf16 storage rounds each output to 2^-9 relative, so a single op sits at ~2-3e-3."""
import math

import pytest
import torch
import torch.nn.functional as F

from oracle import fiber_ref as R
from tests.hip_util import BF, DEV, assert_close, bf, rel_l2

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ops():
    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    from fiber_amd import lib, ops
    lib.load()
    return ops


def rnd(*shape, std=1.0, seed=0):
    g = torch.Generator().manual_seed(seed + sum(shape))
    return torch.randn(*shape, generator=g) * std


@pytest.mark.parametrize("M,N,K,act,res", [
    (300, 384, 128, False, False), (1280, 768, 768, True, True), (389, 136, 96, False, True),
    (4608, 2048, 512, True, False), (64, 64, 64, False, False), (9216, 128, 512, False, True), (37, 3072, 768, True, False),
])
def test_gemm_nt(ops, M, N, K, act, res):
    x, w, b = bf(rnd(M, K)), rnd(N, K, std=K ** -0.5).to(DEV), rnd(N, seed=1).to(DEV)
    r = bf(rnd(M, N, seed=2)) if res else None
    x.requires_grad_(True)
    w.requires_grad_(True)
    b.requires_grad_(True)
    y = ops.linear(x, w, b, residual=r, act="gelu" if act else None)
    wr = w.detach().to(BF).float()
    ref = x.detach().float() @ wr.t() + b.detach()
    if act:
        ref = F.gelu(ref)
    if res:
        ref = ref + r.float()
    assert_close("y", y, ref, 4e-3)
    g = bf(rnd(M, N, seed=3))
    y.backward(g)
    xr = x.detach().float().requires_grad_(True)
    wr2 = wr.clone().requires_grad_(True)
    br = b.detach().clone().requires_grad_(True)
    yr = xr @ wr2.t() + br
    if act:
        yr = F.gelu(yr)
    yr.backward(g.float())
    assert_close("dx", x.grad, xr.grad, 8e-3)
    assert_close("dw", w.grad, wr2.grad, 8e-3)
    assert_close("db", b.grad, br.grad, 8e-3)


@pytest.mark.parametrize("M,N,K", [(33000, 2048, 512), (70001, 1024, 256), (40960, 1536, 1024), (52000, 512, 2048),
                                   (33000, 2056, 512), (30000, 384, 128), (26000, 832, 256)])   # last two: N = 1.5 / 3.25 tile columns
@pytest.mark.parametrize("mode", ["plain", "bias", "gelu_pre", "res_rowscale", "gelu_grad_colsum"])
def test_gemm_large_tiles(ops, M, N, K, mode):
    """Shapes with >= 200 256x256 tiles and N % 256 == 0 take the wide kernel (4-stage K=32 ring, epilogue in two halves);
    the ragged-N case stays on the 256x128 ring kernel.  Ragged M, every epilogue variant, and two runs must agree bit for
    bit (race screen for the counted-vmcnt ring)."""
    x, wb = bf(rnd(M, K)), bf(rnd(N, K, std=K ** -0.5))
    b = rnd(N, seed=1).to(DEV)
    ref = x.float() @ wb.float().t()
    kw, want = {}, None
    if mode == "bias":
        kw, want = dict(bias=b), ref + b
    elif mode == "gelu_pre":
        kw = dict(bias=b, act=1, want_pre=True)
        pre = (ref + b).to(BF).float()
        want = F.gelu(pre)
    elif mode == "res_rowscale":
        rps = 1000
        r = bf(rnd(M, N, seed=2))
        rs = (torch.arange(-(-M // rps), device=DEV) % 3).float() * 0.5
        kw = dict(bias=b, residual=r, rowscale=rs, rows_per_sample=rps)
        want = (ref + b) * rs.repeat_interleave(rps)[:M, None] + r.float()
    elif mode == "gelu_grad_colsum":
        h = bf(rnd(M, N, seed=4))
        kw = dict(act=2, aux=h, want_colsum=True)
        hf = h.float().requires_grad_(True)
        F.gelu(hf).sum().backward()
        want = ref * hf.grad
    else:
        want = ref
    y1, e1 = ops.gemm_nt(x, wb, **kw)
    y2, e2 = ops.gemm_nt(x, wb, **kw)
    assert torch.equal(y1, y2)
    assert_close("y", y1, want, 5e-3)
    if mode == "gelu_pre":
        assert_close("pre", e1, ref + b, 4e-3)
        assert torch.equal(e1, e2)
    if mode == "gelu_grad_colsum":
        assert_close("colsum", e1, y1.float().sum(0), 2e-3)


def _ln_mlp_ref(x, g, b, w1, b1, w2, b2, scale_rows):
    xn = F.layer_norm(x, (x.shape[-1],), g, b, 1e-5)
    br = F.linear(F.gelu(F.linear(xn, w1, b1)), w2, b2)
    return x + (br * scale_rows[:, None] if scale_rows is not None else br)


# (B, L, C, DropPath): rows that are a multiple of the kernel's 128-row tile and rows that are not (ragged last tile, rows of one
# strip in two samples), both channel widths, with / without DropPath (one sample dropped), samples whose rows are / are not whole
# 64-row tiles (the weight-gradient kernel's DropPath fold applies only to the former)
@pytest.mark.parametrize("B,L,C,dp", [(2, 576, 128, False), (3, 144, 128, True), (2, 2304, 256, True), (5, 49, 256, False),
                                      (4, 196, 128, True), (1, 9216, 128, False)])
def test_ln_mlp_fused(ops, B, L, C, dp):
    """LayerNorm -> fc1 -> GELU -> fc2 -> DropPath -> residual as ONE kernel per direction (csrc/mlp_rows.hip; reference
    swin_transformer.py:391 with timm Mlp :325) against the fp32 formulation; gradients of x, both LayerNorm parameters (which ride
    folded in the fc1 weight copy), fc1 and fc2."""
    H = 4 * C
    x = bf(rnd(B, L, C) * 1.3 + 0.2).requires_grad_(True)
    g = (1 + 0.2 * rnd(C, seed=1)).to(DEV).requires_grad_(True)
    b = (0.2 * rnd(C, seed=2)).to(DEV).requires_grad_(True)
    w1 = bf(rnd(H, C, std=C ** -0.5, seed=3)).float().requires_grad_(True)
    b1 = (0.1 * rnd(H, seed=4)).to(DEV).requires_grad_(True)
    w2 = bf(rnd(C, H, std=H ** -0.5, seed=5)).float().requires_grad_(True)
    b2 = (0.1 * rnd(C, seed=6)).to(DEV).requires_grad_(True)
    rs, rv = None, None
    if dp:
        rv = 1.25
        rs = torch.full((B,), rv, device=DEV)
        rs[B // 2] = 0.0
    y = ops._LnMlp.apply(x, g, b, 1e-5, w1, b1, w2, b2, rs, rv)           # (the kernels themselves, whatever widths ops.ln_mlp routes to them)
    ref_in = [t.detach().float().clone().requires_grad_(True) for t in (x, g, b, w1, b1, w2, b2)]
    scale_rows = rs.repeat_interleave(L) if rs is not None else None
    yr = _ln_mlp_ref(ref_in[0].view(-1, C), *ref_in[1:], scale_rows).view(B, L, C)
    assert_close("y", y, yr, 4e-3)
    dy = bf(rnd(B, L, C, seed=7))
    y.backward(dy)
    yr.backward(dy.float())
    assert_close("dx", x.grad, ref_in[0].grad, 8e-3)
    for name, t, r in zip(("dgamma", "dbeta", "dw1", "db1", "dw2", "db2"), (g, b, w1, b1, w2, b2), ref_in[1:]):
        assert_close(name, t.grad, r.grad, 1.2e-2)


def test_ln_mlp_fused_matches_separate_kernels(ops):
    """The fused op against the separate-kernel path it replaces (layernorm_res + mlp) on the same inputs: outputs and input gradients
    agree to bf16 rounding; both are the same function."""
    B, L, C = 2, 1152, 128
    H = 4 * C
    mk = lambda: [bf(rnd(B, L, C)).requires_grad_(True), (1 + 0.1 * rnd(C, seed=1)).to(DEV).requires_grad_(True),
                  (0.1 * rnd(C, seed=2)).to(DEV).requires_grad_(True), bf(rnd(H, C, std=C ** -0.5, seed=3)).float().requires_grad_(True),
                  (0.1 * rnd(H, seed=4)).to(DEV).requires_grad_(True), bf(rnd(C, H, std=H ** -0.5, seed=5)).float().requires_grad_(True),
                  (0.1 * rnd(C, seed=6)).to(DEV).requires_grad_(True)]
    a, c = mk(), mk()
    ya = ops._LnMlp.apply(a[0], a[1], a[2], 1e-5, *a[3:], None, None)
    v, r = ops.layernorm_res(c[0], c[1], c[2], 1e-5)
    yc = ops.mlp(v, *c[3:], residual=r)
    assert_close("y", ya, yc, 6e-3)
    dy = bf(rnd(B, L, C, seed=7))
    ya.backward(dy)
    yc.backward(dy)
    for i, name in enumerate(("dx", "dgamma", "dbeta", "dw1", "db1", "dw2", "db2")):
        assert_close(name, a[i].grad, c[i].grad, 1.2e-2)


@pytest.mark.parametrize("rows,C", [(1000, 128), (77, 32), (513, 96), (1280, 768), (300, 2048), (9216, 256), (100, 3072), (100, 4096)])
def test_layernorm(ops, rows, C):
    x = bf(rnd(rows, C) * 1.5 + 0.3).requires_grad_(True)
    g, b = (1 + 0.1 * rnd(C)).to(DEV).requires_grad_(True), (0.1 * rnd(C, seed=1)).to(DEV).requires_grad_(True)
    y = ops.layernorm(x, g, b, 1e-5)
    xr = x.detach().float().requires_grad_(True)
    gr, br = g.detach().clone().requires_grad_(True), b.detach().clone().requires_grad_(True)
    yr = F.layer_norm(xr, (C,), gr, br, 1e-5)
    assert_close("y", y, yr, 4e-3)
    dy = bf(rnd(rows, C, seed=2))
    y.backward(dy)
    yr.backward(dy.float())
    assert_close("dx", x.grad, xr.grad, 6e-3)
    assert_close("dgamma", g.grad, gr.grad, 3e-3)
    assert_close("dbeta", b.grad, br.grad, 3e-3)


# (2, 14, 14, 768): 4C = 3072, Swin-L's last PatchMerging (swin_large_patch4_window7_224 / window12_384 are registered factories)
@pytest.mark.parametrize("B,H,W,C", [(2, 8, 8, 32), (1, 24, 24, 512), (3, 12, 12, 96), (2, 14, 14, 768)])
def test_patch_merge_ln(ops, B, H, W, C):
    x = bf(rnd(B, H * W, C)).requires_grad_(True)
    g, b = (1 + 0.1 * rnd(4 * C)).to(DEV).requires_grad_(True), (0.1 * rnd(4 * C, seed=1)).to(DEV).requires_grad_(True)
    y = ops.patch_merge_ln(x, g, b, H, W)
    xr = x.detach().float().requires_grad_(True)
    gr, br = g.detach().clone().requires_grad_(True), b.detach().clone().requires_grad_(True)
    xv = xr.view(B, H, W, C)
    z = torch.cat([xv[:, 0::2, 0::2], xv[:, 1::2, 0::2], xv[:, 0::2, 1::2], xv[:, 1::2, 1::2]], -1).view(B, -1, 4 * C)
    yr = F.layer_norm(z, (4 * C,), gr, br, 1e-5)
    assert_close("y", y, yr, 4e-3)
    dy = bf(rnd(B, H * W // 4, 4 * C, seed=2))
    y.backward(dy)
    yr.backward(dy.float())
    assert_close("dx", x.grad, xr.grad, 6e-3)
    assert_close("dgamma", g.grad, gr.grad, 3e-3)
    assert_close("dbeta", b.grad, br.grad, 3e-3)


def _window_ref(qkv, table, B, H, W, heads, ws, shift):
    """fp32 reference in the reference's own formulation: roll -> partition -> attention -> reverse -> roll."""
    C = qkv.shape[-1] // 3
    d = C // heads
    N = ws * ws
    u = qkv.view(B, H, W, 3 * C)
    if shift:
        u = torch.roll(u, (-shift, -shift), (1, 2))
    uw = R.to_windows(u, ws)
    q, k, v = uw.view(-1, N, 3, heads, d).permute(2, 0, 3, 1, 4)
    a = (q * d ** -0.5) @ k.transpose(-1, -2)
    idx = R.rel_pos_index(ws).to(qkv.device)
    a = a + table[idx.view(-1)].view(N, N, heads).permute(2, 0, 1)[None]
    if shift:
        m = R.shift_attn_mask(H, W, ws, shift).to(qkv.device)
        nW = m.shape[0]
        a = (a.view(-1, nW, heads, N, N) + m[None, :, None]).view(-1, heads, N, N)
    o = (a.softmax(-1) @ v).transpose(1, 2).reshape(-1, N, C)
    o = R.from_windows(o, ws, H, W)
    if shift:
        o = torch.roll(o, (shift, shift), (1, 2))
    return o.reshape(B, H * W, C)


@pytest.mark.parametrize("B,H,W,heads,ws,shift", [
    (2, 6, 6, 1, 3, 0), (2, 6, 6, 2, 3, 1), (2, 8, 8, 2, 4, 2), (2, 14, 14, 3, 7, 3), (1, 24, 24, 4, 12, 6),
    (2, 12, 12, 8, 12, 0), (1, 48, 48, 4, 12, 6), (1, 36, 36, 2, 18, 9), (2, 18, 18, 4, 18, 0),
    (1, 28, 28, 2, 14, 7), (1, 32, 32, 2, 16, 0),      # 21-tile kernels with 13 / 16 real tiles (the rest is padding)
    # fine-grained backbone geometries (fusion_swin_transformer_v2.py:293-345): rectangular padded grids, and a SHIFTED
    # single-window grid (there odd blocks shift regardless of the resolution)
    (1, 24, 36, 4, 12, 6), (2, 12, 24, 2, 12, 6), (2, 12, 12, 2, 12, 6), (1, 48, 60, 2, 12, 0),
    # runs of several windows per workgroup in the persistent kernels (G = 96 windows on 64 slices; 160 on 32: ragged last run)
    (6, 48, 48, 4, 12, 6), (40, 24, 24, 16, 12, 0), (33, 24, 24, 16, 12, 6),
])
def test_window_attention(ops, B, H, W, heads, ws, shift):
    C = heads * 32
    qkv = bf(rnd(B, H * W, 3 * C)).requires_grad_(True)
    table = rnd((2 * ws - 1) ** 2, heads, std=0.5).to(DEV).requires_grad_(True)
    o = ops.window_attention(qkv, table, B, H, W, heads, ws, shift)
    qr = qkv.detach().float().requires_grad_(True)
    tr = table.detach().clone().requires_grad_(True)
    oref = _window_ref(qr, tr, B, H, W, heads, ws, shift)
    assert_close("o", o, oref, 5e-3)
    do = bf(rnd(B, H * W, C, seed=5))
    o.backward(do)
    oref.backward(do.float())
    assert_close("dqkv", qkv.grad, qr.grad, 1e-2)
    assert_close("dbias_table", table.grad, tr.grad, 1e-2)
    # with FIBER_WIN_COLSUM=1 (round 4: off by default, the qkv linear's weight-gradient kernel takes the sums) the backward passes also
    # offer the column sums of dqkv (the qkv bias gradient) to the next linear backward; an offer nobody took does not outlive its
    # autograd pass.  By default no offer is made.
    from fiber_amd import ops as ops_mod
    assert getattr(ops_mod._hint_tls, "slot", None) is None
    real_offer, was = ops_mod._offer_colsum, ops_mod._WIN_COLSUM
    for mode in (False, True):
        seen = []
        ops_mod._offer_colsum = lambda t, sums: (seen.append((tuple(t.shape), sums)), real_offer(t, sums))[1]
        ops_mod._WIN_COLSUM = mode
        try:
            qkv.grad = None
            ops.window_attention(qkv, table, B, H, W, heads, ws, shift).backward(do)
        finally:
            ops_mod._offer_colsum, ops_mod._WIN_COLSUM = real_offer, was
        assert getattr(ops_mod._hint_tls, "slot", None) is None
        assert_close("dqkv", qkv.grad, qr.grad, 1e-2)
        if mode:
            assert len(seen) == 1 and seen[0][0] == (B * H * W, 3 * C)
            assert_close("colsum(dqkv)", seen[0][1], qr.grad.reshape(-1, 3 * C).sum(0), 1e-2)
        else:
            assert not seen


def test_window_attention_qkv_bias_grad_handover(ops):
    """linear -> window attention: the qkv bias gradient comes from the weight-gradient kernel's column sums (default) or, with
    FIBER_WIN_COLSUM=1, from the attention backward's fused column sums -- never from a colsum launch of its own --, also with the
    head-major layout, and equals the column sums of the dqkv it describes; nothing is left in the hand-over slot."""
    from fiber_amd import lib as lib_mod, ops as ops_mod
    B, H, W, heads, ws, shift = 2, 24, 24, 4, 12, 6
    C = heads * 32
    x = bf(rnd(B, H * W, C)).requires_grad_(True)
    w = rnd(3 * C, C, std=C ** -0.5).to(DEV).requires_grad_(True)
    b = rnd(3 * C, seed=1, std=0.1).to(DEV).requires_grad_(True)
    table = rnd((2 * ws - 1) ** 2, heads, std=0.5).to(DEV).requires_grad_(True)
    do = bf(rnd(B, H * W, C, seed=5))
    real_call, launched = lib_mod.call, []
    lib_mod.call = lambda name, *a: (launched.append(name), real_call(name, *a))[1]
    was = ops_mod._WIN_COLSUM
    try:
      for mode in (False, True):
        ops_mod._WIN_COLSUM = mode
        for hm in (False, True):
            b.grad = None
            launched.clear()
            qkv = ops.linear_qkv_head_major(x, w, b, heads) if hm else ops.linear(x, w, b)
            qkv.retain_grad()
            ops.window_attention(qkv, table, B, H, W, heads, ws, shift, head_major=hm).backward(do)
            assert "fiber_colsum_bf16" not in launched
            ref = qkv.grad.float().reshape(-1, 3 * C).sum(0)
            if hm:
                ref = ref[ops_mod._qkv_perm(C, heads, ref.device)[1]]
            assert_close("db", b.grad, ref, 1e-2)
    finally:
        lib_mod.call = real_call
        ops_mod._WIN_COLSUM = was
    assert getattr(ops_mod._hint_tls, "slot", None) is None


@pytest.mark.parametrize("B,H,W,heads,ws,shift", [(2, 8, 8, 2, 4, 2), (1, 24, 24, 4, 12, 6), (2, 14, 14, 3, 7, 3), (6, 48, 48, 4, 12, 6),
                                                   (33, 24, 24, 16, 12, 0)])
def test_window_attention_head_major_layout(ops, B, H, W, heads, ws, shift):
    """[heads][3][32] channel layout (permuted qkv projection) gives the same result as the reference [3][heads][32]."""
    C = heads * 32
    x = bf(rnd(B, H * W, C)).requires_grad_(True)
    w = rnd(3 * C, C, std=C ** -0.5).to(DEV).requires_grad_(True)
    b = rnd(3 * C, seed=1, std=0.1).to(DEV).requires_grad_(True)
    table = rnd((2 * ws - 1) ** 2, heads, std=0.5).to(DEV).requires_grad_(True)
    do = bf(rnd(B, H * W, C, seed=5))
    outs = []
    for hm in (False, True):
        for t in (x, w, b, table):
            t.grad = None
        qkv = ops.linear_qkv_head_major(x, w, b, heads) if hm else ops.linear(x, w, b)
        o = ops.window_attention(qkv, table, B, H, W, heads, ws, shift, head_major=hm)
        o.backward(do)
        outs.append((o.detach().clone(), x.grad.clone(), w.grad.clone(), b.grad.clone(), table.grad.clone()))
    for name, a, r in zip(("o", "dx", "dw", "db", "dtable"), outs[1], outs[0]):
        assert_close(name, a, r, 6e-3)


def test_window_attention_backward_helper_wave_variant():
    """FIBER_WIN_BWD_WAVES=11 (key strip 8 of the one-pass 12x12 backward cut over three helper waves, csrc/win_attn.hip) against the shipped
    nine-wave kernel on the same inputs, all four stages, shifted and unshifted, with and without the column-sum offer: dqkv may differ by the
    summation order of strip 8's dK / dV partials (a bf16 ulp: 4e-3 of the tensor's maximum), the bias-table gradient and the column sums by fp32
    round-off.  The switch is read once per process, so the two kernels run in child processes (tools/probes/win_bwd_waves_ab.py)."""
    import os
    import sys
    from tests.mp_util import run_bounded
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = run_bounded([sys.executable, os.path.join(root, "tools", "probes", "win_bwd_waves_ab.py"), "3"], timeout=240, cwd=root)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    rows = [ln.split() for ln in r.stdout.splitlines() if ln[:2] in ("s0", "s1", "s2", "s3")]
    assert len(rows) == 16, r.stdout
    for case, _, _, dq, dt, dc in rows:
        assert float(dq) < 4e-3 and float(dt) < 1e-5 and float(dc) < 1e-5, (case, dq, dt, dc)


@pytest.mark.parametrize("M,N,K", [
    (131072, 384, 128),      # stage-0 qkv: 128x128 tiles, deep split
    (65536, 512, 2048),      # fc2: 256x256 tiles, staggered wave groups
    (36864, 2048, 512),      # fc1
    (4096, 768, 768),        # text layer
    (1000, 96, 288),         # Swin-T-like, M not a multiple of 64 (zero-row tail), N/K not multiples of the tile
    (70, 16, 24),            # tiny: one K tile with a tail, one partly filled 128x128 tile
    (64, 8, 8), (130, 264, 520),
    (18432, 128, 64),        # patch embedding (K = 48 padded to 64)
    (73728, 1536, 512),      # qkv at stage 2: the q8-schedule kernel (round 4), 12 tiles x 21 splits
    (8192, 1024, 1024), (128, 256, 256), (64, 512, 256),   # ... its short K loops: 2 K tiles per split, a single K tile
])
def test_wgrad_tn_kernel(ops, M, N, K):
    """dW = dY^T X (+ bias gradient = column sums of dY) on csrc/gemm_tn.hip against fp32 products of the same bf16 operands.
    Operands are asymmetric random data (a transposed or row/column-swapped result cannot pass); rel-L2 <= 2e-3 (fp32
    accumulation; only the summation order differs)."""
    dy = bf(rnd(M, N, seed=1) + 0.25)                 # non-zero mean: column sums are not ~0
    x = bf(rnd(M, K, seed=2) * torch.linspace(0.5, 2.0, K))
    dw, db = ops.wgrad(dy, x, want_bias=True)
    assert dw.dtype == torch.float32 and dw.shape == (N, K) and db.shape == (N,)
    ref = dy.double().t() @ x.double()
    assert_close("dW", dw, ref, 2e-3)
    assert_close("db", db, dy.double().sum(0), 2e-3)
    assert_close("dW (no bias)", ops.wgrad(dy, x), ref, 2e-3)
    # row-strided operands (column views of wider tensors, e.g. a packed projection)
    wide = bf(rnd(M, N + K, seed=4))
    dw2 = ops.wgrad(wide[:, :N], wide[:, N:])
    assert_close("dW strided", dw2, wide[:, :N].double().t() @ wide[:, N:].double(), 2e-3)


def test_wgrad_tn_identity_layout(ops):
    """A = I-structured check with an ASYMMETRIC partner: dY = one-hot rows selects rows of X, so dW[n] must equal the sum of
    the X rows whose hot column is n -- catches any row/column or operand swap exactly (values are small integers)."""
    M, N, K = 512, 64, 128
    hot = torch.arange(M) % N
    dy = torch.zeros(M, N)
    dy[torch.arange(M), hot] = 1.0
    x = (torch.arange(M)[:, None] % 7 + torch.arange(K)[None, :] % 5).float()
    dw = ops.wgrad(bf(dy), bf(x))
    ref = dy.t() @ x
    assert torch.equal(dw.cpu(), ref), (dw.cpu() - ref).abs().max()


def test_own_dgrad_nt(ops):
    """dX = dY W on the NT kernel with the transposed bf16 working copy of W."""
    for M, N, K in ((131072, 384, 128), (65536, 512, 2048), (4096, 768, 768), (80, 3072, 768)):
        dy = bf(rnd(M, N, seed=1))
        w = rnd(N, K, seed=3, std=K ** -0.5).to(DEV)
        dx = ops._dgrad(dy, w)
        assert_close("dX", dx, dy.float() @ w.to(torch.bfloat16).float(), 6e-3)


def _mha_ref(q, k, v, kmask, B, heads, scale):
    D = q.shape[1] // heads
    qh = q.view(B, -1, heads, D).transpose(1, 2)
    kh = k.view(B, -1, heads, D).transpose(1, 2)
    vh = v.view(B, -1, heads, D).transpose(1, 2)
    a = qh @ kh.transpose(-1, -2) * scale
    if kmask is not None:
        a = a + kmask[:, None, None, :]
    return (a.softmax(-1) @ vh).transpose(1, 2).reshape(q.shape[0], heads * D)


@pytest.mark.parametrize("B,heads,Lq,Lk,D,masked", [
    (2, 12, 40, 40, 64, True), (2, 16, 576, 40, 32, True), (2, 12, 40, 576, 64, False), (3, 2, 12, 12, 64, True),
    (2, 32, 144, 40, 32, True), (2, 12, 40, 144, 64, False), (1, 2, 9, 6, 32, True), (2, 4, 50, 324, 64, False),
    # many (head, sample) pairs: the single-chunk key side is staged once per workgroup, which then walks several query strips
    (48, 16, 576, 40, 32, True), (64, 12, 200, 40, 64, True),
    # the one-pass image -> text backward (attn_x.hip: head_dim 32, <= 48 keys, Lq % 16 == 0, heads % 4 == 0): key-count edges, no mask;
    # and its neighbours that must fall back to the generic passes (49 keys; 3 heads; a ragged last strip)
    (2, 8, 64, 48, 32, False), (3, 4, 32, 17, 32, True), (2, 4, 16, 2, 32, False), (2, 4, 64, 49, 32, True), (2, 3, 64, 40, 32, True),
    (2, 4, 72, 40, 32, True),
    # the one-pass backward for few queries at head_dim 64 (attn_x.hip t2i_bwd_kernel: Lq <= 48; text -> image cross attention, text self
    # attention): ragged key / query counts, masks, a key count that leaves waves without a tile; 49 queries fall back to the generic passes
    (3, 12, 40, 576, 64, True), (2, 2, 48, 33, 64, True), (2, 3, 7, 100, 64, False), (2, 12, 40, 40, 64, True), (1, 1, 49, 64, 64, False),
])
def test_mha(ops, B, heads, Lq, Lk, D, masked):
    C = heads * D
    q, k, v = (bf(rnd(B * L, C, seed=s)).requires_grad_(True) for L, s in ((Lq, 0), (Lk, 1), (Lk, 2)))
    km = None
    if masked:
        lens = torch.randint(max(1, Lk // 3), Lk + 1, (B,), generator=torch.Generator().manual_seed(4))
        lens[0] = Lk
        km = ((torch.arange(Lk)[None] >= lens[:, None]).float() * -10000.0).to(DEV)
    scale = D ** -0.5
    o = ops.mha(q, k, v, km, B, heads, scale)
    qr, kr, vr = (t.detach().float().requires_grad_(True) for t in (q, k, v))
    oref = _mha_ref(qr, kr, vr, km, B, heads, scale)
    assert_close("o", o, oref, 5e-3)
    do = bf(rnd(B * Lq, C, seed=7))
    o.backward(do)
    oref.backward(do.float())
    assert_close("dq", q.grad, qr.grad, 1e-2)
    assert_close("dk", k.grad, kr.grad, 1e-2)
    assert_close("dv", v.grad, vr.grad, 1e-2)


def _random_mha_shapes():
    """48 shapes drawn once (fixed seed) around the one-pass kernels' admission rules (attn_x.hip) and just outside them."""
    import random
    r = random.Random(20240929)
    out = []
    for i in range(48):
        if i % 2 == 0:                                   # head_dim 64: <= 48 queries -> t2i kernels; 49.. and > 1024 keys -> generic
            Lq = r.choice([1, 5, 15, 16, 17, 31, 32, 33, 40, 47, 48, 49, 60])
            Lk = r.choice([1, 3, 15, 16, 17, 40, 63, 64, 65, 144, 200, 577, 1024, 1030])
            out.append((r.choice([1, 2, 3]), r.choice([1, 2, 3, 5, 12]), Lq, Lk, 64, r.random() < 0.5))
        else:                                            # head_dim 32: <= 48 keys, Lq % 16 == 0, heads % 4 == 0 -> i2t kernels
            Lq = r.choice([16, 32, 48, 80, 144, 150, 9])
            Lk = r.choice([1, 2, 15, 16, 17, 32, 40, 47, 48, 49, 70])
            out.append((r.choice([1, 2, 5]), r.choice([4, 8, 16, 3, 6]), Lq, Lk, 32, r.random() < 0.5))
    return out


@pytest.mark.parametrize("B,heads,Lq,Lk,D,masked", _random_mha_shapes())
def test_mha_random_shapes(ops, B, heads, Lq, Lk, D, masked):
    """Forward and backward of ops.mha against torch at shapes on both sides of every admission rule of the one-pass kernels: ragged query /
    key tiles, waves without a tile, more keys than the LDS mask table holds, head counts that are not a multiple of 4, one query, one key."""
    C = heads * D
    q, k, v = (bf(rnd(B * L, C, seed=s)).requires_grad_(True) for L, s in ((Lq, 10), (Lk, 11), (Lk, 12)))
    km = None
    if masked:
        lens = torch.randint(1, Lk + 1, (B,), generator=torch.Generator().manual_seed(Lq * 131 + Lk))
        lens[0] = Lk
        km = ((torch.arange(Lk)[None] >= lens[:, None]).float() * -10000.0).to(DEV)
    scale = D ** -0.5
    o = ops.mha(q, k, v, km, B, heads, scale)
    qr, kr, vr = (t.detach().float().requires_grad_(True) for t in (q, k, v))
    oref = _mha_ref(qr, kr, vr, km, B, heads, scale)
    assert torch.isfinite(o.float()).all()
    assert_close("o", o, oref, 6e-3)
    do = bf(rnd(B * Lq, C, seed=17))
    o.backward(do)
    oref.backward(do.float())
    for name, got, ref in (("dq", q.grad, qr.grad), ("dk", k.grad, kr.grad), ("dv", v.grad, vr.grad)):
        assert torch.isfinite(got.float()).all(), name
        if float(ref.norm()) < 1e-6 * float(do.float().norm()):       # one key: softmax = 1, dq and dk are exactly zero -- what remains
            assert float(got.float().abs().max()) < 5e-2, name       # is the bf16 rounding of O inside delta = sum(dO * O)
        else:
            assert_close(name, got, ref, 1.2e-2)


def test_mha_packed_views(ops):
    """K/V as column views of one packed [B*S, 2C] projection (the i2t layout, swin_transformer.py:230-235)."""
    B, heads, Lq, Lk, D = 2, 4, 64, 40, 32
    C = heads * D
    q = bf(rnd(B * Lq, C))
    kv = bf(rnd(B * Lk, 2 * C, seed=1))
    o = ops.mha(q, kv[:, :C], kv[:, C:], None, B, heads, D ** -0.5)
    oref = _mha_ref(q.float(), kv[:, :C].float(), kv[:, C:].float(), None, B, heads, D ** -0.5)
    assert_close("o", o, oref, 5e-3)


def test_mha_dropout_adjoint(ops):
    """Attention-prob dropout: same seed -> same mask in forward and both backward passes (adjoint identity
    <O(V), dO> = <V, dV> holds only if they agree), unbiased in expectation."""
    B, heads, L, D = 2, 12, 40, 64
    C = heads * D
    q, k = bf(rnd(B * L, C)), bf(rnd(B * L, C, seed=1))
    v = bf(rnd(B * L, C, seed=2)).requires_grad_(True)
    from fiber_amd.ops import _MHA
    o1 = _MHA.apply(q, k, v, None, B, heads, D ** -0.5, 0.1, 1234)
    o2 = _MHA.apply(q, k, v, None, B, heads, D ** -0.5, 0.1, 1234)
    o3 = _MHA.apply(q, k, v, None, B, heads, D ** -0.5, 0.1, 99)
    assert torch.equal(o1, o2) and not torch.equal(o1, o3)
    do = bf(rnd(B * L, C, seed=3))
    o1.backward(do)
    lhs = (o1.float() * do.float()).sum().item()
    rhs = (v.detach().float() * v.grad.float()).sum().item()
    assert abs(lhs - rhs) <= 2e-2 * (abs(lhs) + 1.0), (lhs, rhs)
    base = ops.mha(q, k, v.detach(), None, B, heads, D ** -0.5)
    acc = torch.zeros_like(base, dtype=torch.float32)
    n = 48
    for s in range(n):
        acc += _MHA.apply(q, k, v.detach(), None, B, heads, D ** -0.5, 0.1, 1000 + s).float()
    assert rel_l2(acc / n, base) < 0.08


@pytest.mark.parametrize("Lq,Lk", [(40, 40), (40, 144), (33, 100), (64, 40)])
def test_mha_dropout_mask_is_the_same_in_all_three_passes(ops, Lq, Lk):
    """The keep mask is recomputed from (seed, row, key) in the forward and in the backward (one pass for <= 48 queries, else a dQ pass and a
    dK/dV pass).  Recover it from forwards with one-hot value rows (o[i, head, j] = P[i, j] * keep[i, j] / (1 - p); 64 keys per forward), then
    check dq / dk / dv of a real problem against torch autograd through softmax * mask."""
    B, heads, D, p = 2, 12, 64, 0.1
    C = heads * D
    q, k, v = (bf(rnd(B * L, C, seed=s)).requires_grad_(True) for L, s in ((Lq, 0), (Lk, 1), (Lk, 2)))
    from fiber_amd.ops import _MHA
    keep = torch.zeros(B, heads, Lq, Lk, device=q.device)
    for c0 in range(0, Lk, D):
        n = min(D, Lk - c0)
        eye = torch.zeros(B, Lk, heads, D, device=q.device)
        eye[:, c0 + torch.arange(n), :, torch.arange(n)] = 1.0              # value row j = e_(j - c0) in every head
        probe = _MHA.apply(q.detach(), k.detach(), bf(eye.view(B * Lk, C)), None, B, heads, D ** -0.5, p, 4321)
        keep[..., c0:c0 + n] = (probe.view(B, Lq, heads, D)[..., :n] != 0).permute(0, 2, 1, 3).float()
    assert 0.85 < keep.mean().item() < 0.95
    o = _MHA.apply(q, k, v, None, B, heads, D ** -0.5, p, 4321)
    do = bf(rnd(B * Lq, C, seed=3))
    o.backward(do)
    qr, kr, vr = (t.detach().float().view(B, L, heads, D).permute(0, 2, 1, 3).requires_grad_(True) for t, L in ((q, Lq), (k, Lk), (v, Lk)))
    pr = torch.softmax(qr @ kr.transpose(-1, -2) * D ** -0.5, -1) * keep / (1 - p)
    oref = (pr @ vr).permute(0, 2, 1, 3).reshape(B * Lq, C)
    oref.backward(do.float())
    assert_close("o", o, oref, 1e-2)
    for name, got, ref, L in (("dq", q.grad, qr.grad, Lq), ("dk", k.grad, kr.grad, Lk), ("dv", v.grad, vr.grad, Lk)):
        assert_close(name, got, ref.permute(0, 2, 1, 3).reshape(B * L, C), 2e-2)


def test_roberta_embed(ops):
    from oracle import detgen
    V, C, S, B = 1000, 768, 40, 3
    emb = R.RobertaEmbeddings(V, C, 514, dropout=0.0)
    detgen.fill_(emb)
    b = detgen.synth_batch(B, image_size=8, vocab=V, seed=3)
    ids = b["text_ids"]
    assert (ids == 1).any()
    yr = emb(ids)
    g = rnd(B, S, C, seed=9)
    yr.backward(g.to(BF).float())
    dev = lambda t: t.detach().clone().to(DEV).requires_grad_(True)
    w, p, t, ga, be = dev(emb.word_embeddings.weight), dev(emb.position_embeddings.weight), dev(emb.token_type_embeddings.weight), \
        dev(emb.LayerNorm.weight), dev(emb.LayerNorm.bias)
    y = ops.roberta_embed(ids.to(DEV), w, p, t, ga, be, pad=1, eps=1e-5)
    assert_close("y", y, yr, 4e-3)
    y.backward(bf(g))
    assert_close("dword", w.grad, emb.word_embeddings.weight.grad, 2e-3)
    assert_close("dpos", p.grad, emb.position_embeddings.weight.grad, 2e-3)
    assert_close("dtype", t.grad, emb.token_type_embeddings.weight.grad, 2e-3)
    assert_close("dgamma", ga.grad, emb.LayerNorm.weight.grad, 2e-3)
    assert_close("dbeta", be.grad, emb.LayerNorm.bias.grad, 2e-3)


@pytest.mark.parametrize("B,img,dim", [(2, 32, 32), (1, 384, 128)])
def test_patch_embed(ops, B, img, dim):
    x = rnd(B, 3, img, img).to(DEV)
    w, b = rnd(dim, 3, 4, 4, std=0.1).to(DEV).requires_grad_(True), rnd(dim, seed=1, std=0.1).to(DEV).requires_grad_(True)
    y = ops.patch_embed_proj(x, w, b)
    xr = x.to(BF).float()
    wr, br = w.detach().to(BF).float().requires_grad_(True), b.detach().clone().requires_grad_(True)
    yr = F.conv2d(xr, wr, br, stride=4).flatten(2).transpose(1, 2)
    assert_close("y", y, yr, 4e-3)
    g = bf(rnd(*yr.shape, seed=2))
    y.backward(g)
    yr.backward(g.float())
    assert_close("dw", w.grad, wr.grad, 8e-3)
    assert_close("db", b.grad, br.grad, 4e-3)


@pytest.mark.parametrize("B,img,dim", [(5, 32, 32), (3, 384, 128)])
def test_patch_embed_of_an_image_pair_is_the_patch_embed_of_the_concatenated_batch(ops, B, img, dim):
    """ops.ImagePair (the [true ; where(label, true, false)] image batch of the one-pass MLM + ITM step, gathered from its two sources by
    fiber_im2col_patch4_pair) against ops.patch_embed_proj on the batch built with torch.where / torch.cat: bitwise the same output and gradients."""
    a, alt = rnd(B, 3, img, img).to(DEV), rnd(B, 3, img, img, seed=3).to(DEV)
    sel = torch.tensor([i % 3 != 1 for i in range(B)], device=DEV)
    w, b = rnd(dim, 3, 4, 4, std=0.1).to(DEV).requires_grad_(True), rnd(dim, seed=1, std=0.1).to(DEV).requires_grad_(True)
    pair = ops.ImagePair(a, alt, sel)
    assert tuple(pair.shape) == (2 * B, 3, img, img)
    ref_batch = torch.cat([a, torch.where(sel.view(-1, 1, 1, 1), a, alt)], 0)
    assert torch.equal(pair.tensor(), ref_batch)
    g = bf(rnd(2 * B, (img // 4) ** 2, dim, seed=2))
    outs = []
    for x in (pair, ref_batch):
        w.grad = b.grad = None
        y = ops.patch_embed_proj(x, w, b)
        y.backward(g)
        outs.append((y.detach().clone(), w.grad.clone(), b.grad.clone()))
    for name, p_, r_ in zip(("y", "dw", "db"), outs[0], outs[1]):
        assert torch.equal(p_, r_), name


def test_elementwise(ops):
    a, b = bf(rnd(4, 100, 64)).requires_grad_(True), bf(rnd(4, 100, 64, seed=1)).requires_grad_(True)
    alpha = torch.tensor([0.37], device=DEV, requires_grad=True)
    out = ops.scale_add(a, b, alpha)
    assert_close("scale_add", out, a.float() + 0.37 * b.float(), 4e-3)
    g = bf(rnd(4, 100, 64, seed=2))
    out.backward(g)
    assert_close("da", a.grad, g, 1e-6)
    assert_close("db", b.grad, 0.37 * g.float(), 4e-3)
    assert_close("dalpha", alpha.grad, (g.float() * b.detach().float()).sum().view(1), 2e-3)
    s = ops.add(a.detach(), b.detach())
    assert_close("add", s, a.float() + b.float(), 4e-3)
    x = bf(rnd(8, 4096)).requires_grad_(True)
    y = ops.dropout(x, 0.1, True)
    keep = (y != 0).float().mean().item()
    assert abs(keep - 0.9) < 0.01
    kept = y != 0
    assert_close("dropout scale", y[kept], x.detach().float()[kept] / 0.9, 4e-3)
    y.backward(torch.ones_like(y))
    assert torch.equal(x.grad != 0, kept)
    r, xx = bf(rnd(6, 50, 64)), bf(rnd(6, 50, 64, seed=5)).requires_grad_(True)
    z = ops.drop_path_add(r, xx, 0.5, True)
    diff = (z.float() - r.float()).view(6, -1)
    ratio = diff.abs().sum(1) / xx.detach().float().view(6, -1).abs().sum(1)
    assert all(abs(v) < 1e-2 or abs(v - 2.0) < 2e-2 for v in ratio.tolist()), ratio


def test_adamw_multi_matches_hf_rule(ops):
    """csrc/optim.hip vs the transformers 4.6.0 AdamW(correct_bias=True) rule (fiber_utils.py:248-252; restated in plain
    torch as fiber_amd.optim.HFAdamW: eps on the un-corrected sqrt(v), weight decay after the update): two
    groups with different lr / weight decay, ragged and unaligned sizes, a scalar parameter, a parameter without a gradient,
    three steps with a changing lr; the cached bf16 working copy of a weight is rewritten by the same kernel."""
    from fiber_amd.optim import FiberAdamW, HFAdamW
    shapes = [(768, 512), (4099,), (), (3, 5, 7), (1 << 20,), (17,)]
    mine = [rnd(*s, seed=i).to(DEV).requires_grad_(True) if s else torch.tensor(0.3, device=DEV, requires_grad=True)
            for i, s in enumerate(shapes)]
    ref = [p.detach().clone().requires_grad_(True) for p in mine]
    wb = ops.bf16_weight(mine[0])                                    # creates the cached working copy
    groups = lambda ps: [{"params": ps[:3], "weight_decay": 0.01, "lr": 1e-3}, {"params": ps[3:], "weight_decay": 0.0, "lr": 5e-3}]
    om = FiberAdamW(groups(mine), lr=1e-3, betas=(0.9, 0.98), eps=1e-8)
    ot = HFAdamW(groups(ref), lr=1e-3, betas=(0.9, 0.98), eps=1e-8)
    for step in range(3):
        for i, (a, b) in enumerate(zip(mine, ref)):
            if i == 5:
                continue                                             # never receives a gradient
            g = rnd(*shapes[i], seed=100 * step + i).to(DEV) if shapes[i] else torch.tensor(0.1 * (step + 1), device=DEV)
            a.grad, b.grad = g.clone(), g.clone()
        for o in (om, ot):
            for gidx, grp in enumerate(o.param_groups):
                grp["lr"] = (1e-3, 5e-3)[gidx] * (1.0 - 0.2 * step)
            o.step()
        for a, b in zip(mine, ref):
            assert torch.allclose(a, b, rtol=2e-6, atol=1e-7), (step, a.shape, (a - b).abs().max().item())
        assert ops.bf16_weight(mine[0]) is wb                        # still the same tensor, now holding the new weights
        assert torch.equal(wb, mine[0].detach().to(BF))
    assert torch.equal(mine[5], ref[5])


def test_adamw_multi_bucket_view_gradients(ops):
    """Gradients that are views into one flat buffer at 4-byte (not 16-byte) aligned offsets -- what DDP's
    gradient_as_bucket_view hands the optimizer after a 1-element parameter (alpha_i2t / alpha_t2i) sits in the bucket."""
    from fiber_amd.optim import FiberAdamW, HFAdamW
    sizes = [1, 4096 + 8, 1, 333, 2048]
    flat = rnd(sum(sizes) + 3, seed=9).to(DEV)
    mine = [rnd(n, seed=20 + i).to(DEV).requires_grad_(True) for i, n in enumerate(sizes)]
    ref = [p.detach().clone().requires_grad_(True) for p in mine]
    off = 3                                                  # first view starts 12 bytes into the buffer
    for a, b, n in zip(mine, ref, sizes):
        a.grad = flat[off:off + n]
        b.grad = flat[off:off + n].clone()
        off += n
    om = FiberAdamW(mine, lr=1e-2, betas=(0.9, 0.98), eps=1e-8, weight_decay=0.01)
    ot = HFAdamW(ref, lr=1e-2, betas=(0.9, 0.98), eps=1e-8, weight_decay=0.01)
    for _ in range(2):
        om.step()
        ot.step()
    for a, b in zip(mine, ref):
        assert torch.allclose(a, b, rtol=2e-6, atol=1e-7), (a.shape, (a - b).abs().max().item())


def test_adamw_load_state_dict_and_moved_storage(ops):
    """A state dict loaded AFTER a step must be what the kernel updates from then on (the cached device tables hold raw
    pointers), and a parameter whose storage moved under an unchanged id must not be written through the old pointer."""
    from fiber_amd.optim import FiberAdamW, HFAdamW
    mk = lambda: [rnd(300, 40, seed=1).to(DEV).requires_grad_(True), rnd(77, seed=2).to(DEV).requires_grad_(True)]
    a, b, r = mk(), mk(), mk()
    oa = FiberAdamW(a, lr=1e-2, betas=(0.9, 0.98), eps=1e-8, weight_decay=0.01)
    ob = FiberAdamW(b, lr=1e-2, betas=(0.9, 0.98), eps=1e-8, weight_decay=0.01)
    orf = HFAdamW(r, lr=1e-2, betas=(0.9, 0.98), eps=1e-8, weight_decay=0.01)
    grads = lambda k: [rnd(300, 40, seed=10 + k).to(DEV), rnd(77, seed=20 + k).to(DEV)]

    def step(opt, ps, k):
        for p, g in zip(ps, grads(k)):
            p.grad = g.clone()
        opt.step()
    for k in range(2):
        step(oa, a, k)
        step(orf, r, k)
    step(ob, b, 7)                                          # ob has stepped once with other gradients: its tables are built
    with torch.no_grad():
        for pb, pa in zip(b, a):
            pb.copy_(pa)
    ob.load_state_dict(oa.state_dict())                     # now it must continue a's trajectory
    b[0].data = b[0].data.clone()                           # storage moves, id(b[0]) stays
    for k in range(2, 4):
        step(ob, b, k)
        step(orf, r, k)
    for pb, pr in zip(b, r):
        assert torch.allclose(pb, pr, rtol=2e-6, atol=1e-7), (pb - pr).abs().max().item()
    sd = ob.state_dict()["state"]
    assert all(int(v["step"]) == 4 for v in sd.values())


@pytest.mark.parametrize("B,L,C", [(4, 576, 512), (5, 2304, 256), (3, 144, 1024), (4, 128, 64)])
def test_droppath_folded_backward(ops, B, L, C):
    """timm DropPath on both residual branches of a Swin block (swin_transformer.py:390-391): y = r + s_b * branch(x) with
    s_b in {0, 1/keep}.  With `rowscale_value` = 1/keep the backward never materialises s_b * dy: the factor rides in the dgrad
    epilogue / the fused gelu' GEMM and the weight-gradient kernel skips the dropped samples (csrc/gemm_tn.hip); 144 rows per
    sample is not a multiple of the 64-row K tile and takes the separate pass.  Checked against autograd on the fp32 formula."""
    keep = 0.8
    s = torch.tensor([1 / keep, 0.0, 1 / keep, 1 / keep, 0.0][:B], device=DEV)
    g = bf(rnd(B, L, C, seed=3))
    # proj-like linear with residual
    x = bf(rnd(B, L, C)).requires_grad_(True)
    w = rnd(C, C, std=C ** -0.5).to(DEV).requires_grad_(True)
    b = rnd(C, seed=1).to(DEV).requires_grad_(True)
    r = bf(rnd(B, L, C, seed=2)).requires_grad_(True)
    y = ops.linear(x, w, b, residual=r, rowscale=s, rowscale_value=1 / keep)
    xr, wr, br, rr = (t.detach().float().requires_grad_(True) for t in (x, w.to(BF), b, r))
    yr = rr + s.view(B, 1, 1) * (xr @ wr.t() + br)
    assert_close("y", y, yr, 4e-3)
    y.backward(g)
    yr.backward(g.float())
    for name, got, ref in (("dx", x.grad, xr.grad), ("dw", w.grad, wr.grad), ("db", b.grad, br.grad), ("dr", r.grad, rr.grad)):
        assert_close("linear " + name, got, ref, 8e-3)
    assert float(x.grad[1].abs().max()) == 0.0                        # dropped sample: exactly no gradient
    # MLP with residual
    w1 = rnd(4 * C, C, seed=5, std=C ** -0.5).to(DEV).requires_grad_(True)
    b1 = rnd(4 * C, seed=6, std=0.1).to(DEV).requires_grad_(True)
    w2 = rnd(C, 4 * C, seed=7, std=(4 * C) ** -0.5).to(DEV).requires_grad_(True)
    b2 = rnd(C, seed=8, std=0.1).to(DEV).requires_grad_(True)
    x2 = bf(rnd(B, L, C, seed=9)).requires_grad_(True)
    r2 = bf(rnd(B, L, C, seed=10)).requires_grad_(True)
    y2 = ops.mlp(x2, w1, b1, w2, b2, residual=r2, rowscale=s, rowscale_value=1 / keep)
    ref = [t.detach().float().requires_grad_(True) for t in (x2, w1.to(BF), b1, w2.to(BF), b2, r2)]
    h = F.gelu(ref[0] @ ref[1].t() + ref[2])
    y2r = ref[5] + s.view(B, 1, 1) * (h.to(BF).float() @ ref[3].t() + ref[4])
    assert_close("mlp y", y2, y2r, 6e-3)
    y2.backward(g)
    y2r.backward(g.float())
    for name, got, rf in zip(("dx", "dw1", "db1", "dw2", "db2", "dr"), (x2.grad, w1.grad, b1.grad, w2.grad, b2.grad, r2.grad),
                             (t.grad for t in ref)):
        assert_close("mlp " + name, got, rf, 1.2e-2)
    assert float(x2.grad[1].abs().max()) == 0.0


@pytest.mark.parametrize("B,L,K,N", [(3, 1000, 128, 136), (4, 576, 512, 512)])
def test_linear_droppath_scale_backward(ops, B, L, K, N):
    """y = shortcut + s_b * (x.W^T + b): the backward computes s_b * dy and the bias gradient in ONE pass
    (fiber_rowscale_colsum_bf16); checked against autograd on the fp32 formula (timm DropPath, swin_transformer.py:390-391)."""
    x = bf(rnd(B, L, K)).requires_grad_(True)
    w = rnd(N, K, std=K ** -0.5).to(DEV).requires_grad_(True)
    b = rnd(N, seed=1).to(DEV).requires_grad_(True)
    r = bf(rnd(B, L, N, seed=2)).requires_grad_(True)
    s = torch.tensor([0.0, 1.25, 1.0, 2.0][:B], device=DEV)
    y = ops.linear(x, w, b, residual=r, rowscale=s)
    xr, wr, br, rr = (t.detach().float().requires_grad_(True) for t in (x, w.to(BF), b, r))
    yr = rr + s.view(B, 1, 1) * (xr @ wr.t() + br)
    assert_close("y", y, yr, 4e-3)
    g = bf(rnd(B, L, N, seed=3))
    y.backward(g)
    yr.backward(g.float())
    assert_close("dx", x.grad, xr.grad, 8e-3)
    assert_close("dw", w.grad, wr.grad, 8e-3)
    assert_close("db", b.grad, br.grad, 8e-3)
    assert_close("dres", r.grad, rr.grad, 1e-6)


@pytest.mark.parametrize("rows,V", [(37, 50265), (256, 1000), (5, 17)])
def test_cross_entropy_bf16(ops, rows, V):
    """csrc/loss.hip vs F.cross_entropy(logits.float(), labels, ignore_index=-100) (objectives.py:24-28): odd vocabulary size
    (rows start at unaligned addresses), mostly ignored rows as in MLM, one all-ignored edge (handled by the clamp)."""
    x = bf(rnd(rows, V, seed=1) * 3).requires_grad_(True)
    lab = torch.randint(0, V, (rows,), generator=torch.Generator().manual_seed(0))
    lab[torch.rand(rows, generator=torch.Generator().manual_seed(1)) < 0.7] = -100
    lab[0] = V - 1
    lab = lab.to(DEV)
    loss = ops.cross_entropy(x, lab)
    xr = x.detach().float().requires_grad_(True)
    want = F.cross_entropy(xr, lab, ignore_index=-100)
    assert abs(loss.item() - want.item()) < 2e-5 * max(1.0, abs(want.item())), (loss.item(), want.item())
    (loss * 1.7).backward()
    (want * 1.7).backward()
    assert_close("dlogits", x.grad, xr.grad, 6e-3)
    assert float(x.grad[lab == -100].abs().max()) == 0.0


@pytest.mark.parametrize("rows,V", [(64, 50265), (600, 1000), (5000, 777)])
def test_cross_entropy_hands_over_argmax_and_labelled_rows(ops, rows, V):
    """Two by-products of the cross-entropy kernels for the MLM head (csrc/loss.hip): (1) the arg max of every labelled row -- coarse bf16 logits
    with many exact ties, so the smallest-index rule of torch.argmax is exercised -- as fiber_utils.Accuracy consumes it; (2) the decoder's bias
    gradient as column sums over the labelled rows only, against dlogits.float().sum(0), through ops.lib_linear's backward."""
    from fiber_amd.modules.fiber_utils import Accuracy
    g = torch.Generator().manual_seed(3)
    h = bf(torch.randn(rows, 48, generator=g))
    w = (torch.randn(V, 48, generator=g) * 0.3).to(DEV).to(BF).requires_grad_(True)
    b = (torch.randn(V, generator=g) * 0.1).to(DEV).to(BF).requires_grad_(True)
    lab = torch.randint(0, V, (rows,), generator=g)
    lab[torch.rand(rows, generator=g) < 0.8] = -100
    lab[0] = 3
    lab = lab.to(DEV)
    logits = ops.lib_linear(h, w, b)
    logits.data.copy_((logits.detach().float() * 2).round() / 2)          # half-integer logits: ties in almost every row
    loss = ops.cross_entropy(logits, lab)
    pred, lab_seen, ignore = logits._fiber_argmax
    valid = lab != -100
    want = logits.detach().argmax(-1)
    assert torch.equal(pred[valid].long(), want[valid]) and bool((pred[~valid] == -1).all())
    ties = (logits.detach()[valid] == logits.detach()[valid].max(-1, keepdim=True).values).sum(-1)
    assert int((ties > 1).sum()) > 0, "the case is meant to contain ties"
    acc = Accuracy()(logits, lab)
    logits2 = logits.detach().clone()                                       # (no hand-over on this object: the metric's own argmax)
    assert float(acc) == float(Accuracy()(logits2, lab))
    loss.backward()
    x2 = logits.detach().requires_grad_(True)
    ops.cross_entropy(x2, lab).backward()
    want_db = x2.grad.float().sum(0)
    assert float(x2.grad[~valid].abs().max()) == 0.0
    assert_close("db", b.grad, want_db, 4e-3)


def test_transposed_weight_copies_refreshed_in_one_launch():
    """FiberAdamW.step() rewrites every cached W^T working copy with one multi-tensor transpose: same storage as before the
    step, contents = transpose of the updated bf16 copy, and the lazy path is not taken again (stamps current)."""
    from fiber_amd import ops as ops_mod
    from fiber_amd.optim import FiberAdamW
    torch.manual_seed(0)
    shapes = [(128, 64), (72, 200), (256, 256), (8, 40), (3072, 768)]
    ws = [torch.nn.Parameter(torch.randn(n, k, device=DEV)) for n, k in shapes]
    opt = FiberAdamW(ws, lr=1e-2, weight_decay=0.01)
    x = [torch.randn(4, k, device=DEV).to(BF).requires_grad_(True) for _, k in shapes]
    for it in range(3):
        opt.zero_grad(set_to_none=True)
        loss = sum(ops_mod.linear(xi, w).float().square().mean() for xi, w in zip(x, ws))
        loss.backward()                                           # dX uses bf16_weight_t(w): creates / reuses the copies
        ptrs = [ops_mod.bf16_weight_t(w).data_ptr() for w in ws]
        opt.step()
        for w, p0 in zip(ws, ptrs):
            hit = ops_mod._cache_get(("T", id(w)), w)
            assert hit is not None and hit[0] == ops_mod._stamp(w), "transposed copy not marked current after the step"
            assert hit[1].data_ptr() == p0
            assert torch.equal(hit[1], w.detach().to(BF).t().contiguous())
            assert torch.equal(ops_mod.bf16_weight(w), w.detach().to(BF))


def test_head_major_qkv_copies_refreshed_in_one_launch():
    """FiberAdamW.step() rewrites the permuted working copies of every head-major qkv projection (bf16 weight rows in [heads][3][32] order, their
    transpose, the permuted fp32 bias) with fiber_rowperm_cast_multi_bf16: same storage as before the step, contents = what the lazy ATen path
    would build from the updated parameters, stamps current (the lazy path is not taken again)."""
    from fiber_amd import ops as ops_mod
    from fiber_amd.optim import FiberAdamW
    torch.manual_seed(0)
    cfgs = [(128, 4), (256, 8), (64, 2), (512, 16)]
    lins = [torch.nn.Linear(C, 3 * C).to(DEV) for C, _ in cfgs]
    for m in lins:
        m.bias.data.normal_(0, 0.1)
    opt = FiberAdamW([p for m in lins for p in m.parameters()], lr=1e-2, weight_decay=0.01)
    xs = [torch.randn(2, 16, C, device=DEV).to(BF).requires_grad_(True) for C, _ in cfgs]
    for it in range(3):
        opt.zero_grad(set_to_none=True)
        loss = sum(ops_mod.linear_qkv_head_major(x, m.weight, m.bias, h).float().square().mean() for x, m, (_, h) in zip(xs, lins, cfgs))
        loss.backward()
        ptrs = [[t.data_ptr() for t in ops_mod._wcache[("HM", id(m.weight))][1][:3]] for m in lins]
        opt.step()
        for m, (C, h), p0 in zip(lins, cfgs, ptrs):
            hit = ops_mod._cache_get(("HM", id(m.weight)), m.weight)
            assert hit is not None and hit[0] == (ops_mod._stamp(m.weight), ops_mod._stamp(m.bias)), "head-major copies not marked current after the step"
            wp, bp, wpt = hit[1][:3]
            assert [wp.data_ptr(), bp.data_ptr(), wpt.data_ptr()] == p0
            perm = ops_mod._qkv_perm(C, h, m.weight.device)[0]
            assert torch.equal(wp, m.weight.detach()[perm].to(BF))
            assert torch.equal(wpt, m.weight.detach()[perm].to(BF).t().contiguous())
            assert torch.equal(bp, m.bias.detach()[perm])


@pytest.mark.parametrize("B,L,C,N", [(3, 144, 512, 768), (2, 37, 64, 24)])
def test_packed_projection_fork_adds_the_other_consumers_gradient_in_its_dgrad(ops, B, L, C, N):
    """ops.linear_packed(..., fork_sink=[]) hands out an alias of its input for the input's OTHER consumer; that consumer's gradient then reaches
    the node itself and is added by the dX GEMM's residual epilogue.  Against the unforked graph (autograd's own fan-in add): the same outputs,
    the same weight gradients, the input gradient equal up to one bf16 rounding (sum rounded once instead of twice)."""
    x0 = bf(rnd(B, L, C))
    w = [rnd(N, C, seed=s_, std=C ** -0.5).to(DEV).requires_grad_(True) for s_ in (1, 2)]
    b = [rnd(N, seed=s_ + 5, std=0.1).to(DEV).requires_grad_(True) for s_ in (1, 2)]
    g1, g2 = bf(rnd(B, L, 2 * N, seed=9)), bf(rnd(B, L, C, seed=10))
    res = []
    for fork in (False, True):
        x = x0.clone().requires_grad_(True)
        for t in w + b:
            t.grad = None
        sink = [] if fork else None
        y = ops.linear_packed(x, [(w[0], b[0]), (w[1], b[1])], fork_sink=sink)
        assert (len(sink) == 1 and sink[0].data_ptr() == x.data_ptr()) if fork else True
        other = (sink[0] if fork else x) * 0.5                     # the other consumer
        torch.autograd.backward([y, other], [g1, g2])
        res.append((y.detach().clone(), x.grad.clone(), [t.grad.clone() for t in w + b]))
    assert torch.equal(res[0][0], res[1][0])
    for a_, r_ in zip(res[1][2], res[0][2]):
        assert torch.equal(a_, r_)
    want = (g1.float().view(-1, 2 * N) @ torch.cat([w[0], w[1]]).detach().to(BF).float()).view(B, L, C) + 0.5 * g2.float()
    assert_close("dx forked", res[1][1], want, 4e-3)
    assert_close("dx plain", res[0][1], want, 6e-3)


@pytest.mark.parametrize("B,S,L,hid,kvdim", [(3, 40, 144, 768, 1024), (2, 12, 36, 64, 128)])
def test_packed_projections_and_packed_mha(ops, B, S, L, hid, kvdim):
    """ops.linear_packed + ops.mha_qkv_packed / mha_kv_packed (q | k | v of RobertaSelfAttention and key | value of t2i as ONE GEMM,
    the attention backward writing one packed gradient) against three separate nn.Linear + softmax attention in fp32: outputs,
    input gradients and every member's weight / bias gradient; the members' Parameters keep their identity (biases are re-pointed
    at slices of one buffer, values unchanged) and a parameter update through the members is seen by the next call."""
    heads = hid // 64 if hid >= 64 * 2 else 1
    D = hid // heads
    torch.manual_seed(0)
    lin = {n: torch.nn.Linear(hid if n in "qkv" else kvdim, hid) for n in ("q", "k", "v", "ck", "cv")}
    for m in lin.values():
        m.to(DEV)
        m.weight.data.mul_(2.0)
        m.bias.data.normal_(0, 0.1)
    b0 = {n: m.bias.detach().clone() for n, m in lin.items()}
    x = bf(rnd(B, S, hid)).requires_grad_(True)
    img = bf(rnd(B, L, kvdim, seed=1)).requires_grad_(True)
    mask = torch.zeros(B, S)
    mask[1:, S - 3:] = -10000.0
    scale = D ** -0.5

    def ours():
        qkv = ops.linear_packed(x, [(lin["q"].weight, lin["q"].bias), (lin["k"].weight, lin["k"].bias), (lin["v"].weight, lin["v"].bias)])
        o1 = ops.mha_qkv_packed(qkv.view(B * S, 3 * hid), mask.to(DEV), B, heads, scale)
        q2 = ops.linear(o1.view(B, S, hid), lin["q"].weight, lin["q"].bias).view(B * S, hid)
        kv = ops.linear_packed(img, [(lin["ck"].weight, lin["ck"].bias), (lin["cv"].weight, lin["cv"].bias)])
        return ops.mha_kv_packed(q2, kv.view(B * L, 2 * hid), None, B, heads, scale).view(B, S, hid)

    def ref(xr, ir, P):
        def attn(q, k, v, m):
            q = q.view(B, -1, heads, D).transpose(1, 2); k = k.view(B, -1, heads, D).transpose(1, 2); v = v.view(B, -1, heads, D).transpose(1, 2)
            a = q @ k.transpose(-1, -2) * scale
            if m is not None:
                a = a + m[:, None, None, :]
            return (a.softmax(-1) @ v).transpose(1, 2).reshape(B, -1, hid)
        f = lambda n, t: F.linear(t, P[n + ".w"], P[n + ".b"])
        o1 = attn(f("q", xr), f("k", xr), f("v", xr), mask.to(DEV)).to(BF).float()
        return attn(f("q", o1), f("ck", ir), f("cv", ir), None)

    out = ours()
    for n, m in lin.items():
        assert torch.equal(m.bias.detach(), b0[n]), "re-pointing the biases must not change their values"
    g = bf(rnd(B, S, hid, seed=5))
    out.backward(g)
    P = {}
    for n, m in lin.items():
        P[n + ".w"] = m.weight.detach().to(BF).float().requires_grad_(True)
        P[n + ".b"] = m.bias.detach().clone().requires_grad_(True)
    xr, ir = x.detach().float().requires_grad_(True), img.detach().float().requires_grad_(True)
    outr = ref(xr, ir, P)
    outr.backward(g.float())
    assert_close("out", out, outr, 1.2e-2)
    assert_close("dx", x.grad, xr.grad, 3e-2)
    assert_close("dimg", img.grad, ir.grad, 3e-2)
    for n, m in lin.items():
        assert_close(f"dW {n}", m.weight.grad, P[n + ".w"].grad, 3e-2)
        if n in ("k", "ck"):                               # key biases: mathematically zero gradients (softmax shift invariance)
            assert float(m.bias.grad.abs().max()) < 2e-2 * float(lin["v"].bias.grad.abs().max() + 1e-6) + 1e-3
        else:
            assert_close(f"db {n}", m.bias.grad, P[n + ".b"].grad, 3e-2)
    # an update through the member Parameters (what any optimizer does) must reach the packed copies
    with torch.no_grad():
        lin["k"].weight.add_(0.05)
        lin["v"].bias.add_(0.5)
    out2 = ours()
    P["k.w"] = lin["k"].weight.detach().to(BF).float()
    P["v.b"] = lin["v"].bias.detach().clone()
    assert_close("out after update", out2, ref(x.detach().float(), img.detach().float(), {k: v.detach() for k, v in P.items()}), 1.2e-2)
    assert rel_l2(out2, out) > 1e-3
