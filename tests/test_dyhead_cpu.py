"""CPU: the DyHead mirror's WIRING (fiber_amd/modules/dyhead.py: three-level mixing, the level-l offsets re-read with the level-(l+1)
geometry, GroupNorm, scale attention, DYReLU, parameter names) against the fixtures produced by the reference's own
layers/dyhead.py -- with the HIP operator replaced, inside this test only, by the oracle's restatement, so that the comparison is
fp32-exact (<= 1e-5) and runs without a GPU.  The kernels themselves are held to the oracle by tests/test_hip_dcn.py."""
import numpy as np
import pytest
import torch

from oracle import dcn_ref
from oracle import gen_dyhead_golden as G


def _oracle_op(x, offset, mask, weight, bias=None, stride=1, pad=1, out_fp32=False):
    B, H, W, _ = x.shape
    Ho, Wo = dcn_ref._geom(H, W, 3, 3, stride, pad, 1)
    xn = x.permute(0, 3, 1, 2).float()
    off = offset.view(B, Ho, Wo, 18).permute(0, 3, 1, 2) if offset is not None else torch.zeros(B, 18, Ho, Wo)
    m = mask.view(B, Ho, Wo, 9).permute(0, 3, 1, 2) if mask is not None else torch.ones(B, 9, Ho, Wo)
    return dcn_ref.modulated_deform_conv(xn, off, m, weight, bias, stride, pad).permute(0, 2, 3, 1)


@pytest.mark.parametrize("name", sorted(G.DY_CASES))
def test_dyhead_wiring_against_reference_fixture(name, monkeypatch):
    from fiber_amd import ops
    from fiber_amd.modules import dyhead as DH
    monkeypatch.setattr(ops, "deform_conv", _oracle_op)
    monkeypatch.setattr(DH, "BF16", torch.float32)          # no rounding of the level maps: the wiring is compared exactly
    c = G.DY_CASES[name]
    gold = np.load(f"tests/golden/{name}.npz")
    model = DH.DyHead(G.make_cfg(c), c["C"])
    assert [k for k, _ in model.named_parameters()] == list(gold["param_names"])
    G.set_weights(model, name)
    xs = [x.clone().requires_grad_(True) for x in G.inputs(name)]
    outs = model(xs)
    G.projection(name, outs).backward()
    for i, o in enumerate(outs):
        assert torch.allclose(o, torch.from_numpy(gold[f"out{i}"]), atol=2e-5, rtol=1e-4), f"out{i}"
        assert torch.allclose(xs[i].grad, torch.from_numpy(gold[f"dx{i}"]), atol=2e-5, rtol=1e-3), f"dx{i}"
    for (k, p), r in zip(model.named_parameters(), gold["grad_norms"]):
        assert abs(float(p.grad.norm()) - r) <= 1e-3 * r + 1e-6, k


def test_offsets_larger_than_the_output_are_read_like_the_reference_kernels():
    from fiber_amd.modules import dyhead as DH
    t = torch.arange(2 * 18 * 6 * 8, dtype=torch.float32).view(2, 18, 6, 8)
    a = DH._as_read_by_kernel(t, 3, 4)
    assert torch.equal(a, dcn_ref.as_read_by_kernel(t, 3, 4))
    assert torch.equal(a[1, 0].flatten(), t[1].flatten()[:12])           # channel 0 = the first Ho*Wo values of the image's block
    assert DH._as_read_by_kernel(t, 6, 8) is t
    with pytest.raises(ValueError):
        DH._as_read_by_kernel(t, 7, 8)
