"""CPU pins of the deformable-convolution oracle (oracle/dcn_ref.py): the ordinary-convolution identity against
torch.nn.functional.conv2d, and the explicit-loop restatements of the three reference CUDA kernels against the vectorised form."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import dcn_ref as D


@pytest.mark.parametrize("stride", [1, 2])
def test_zero_offsets_unit_mask_is_conv2d(stride):
    g = torch.Generator().manual_seed(0)
    B, C, H, W, Co = 2, 5, 9, 11, 7
    x = torch.randn(B, C, H, W, generator=g, dtype=torch.float64, requires_grad=True)
    w = torch.randn(Co, C, 3, 3, generator=g, dtype=torch.float64, requires_grad=True)
    b = torch.randn(Co, generator=g, dtype=torch.float64, requires_grad=True)
    Ho, Wo = D._geom(H, W, 3, 3, stride, 1, 1)
    y = D.modulated_deform_conv(x, torch.zeros(B, 18, Ho, Wo, dtype=torch.float64), torch.ones(B, 9, Ho, Wo, dtype=torch.float64), w, b, stride, 1)
    ref = F.conv2d(x, w, b, stride=stride, padding=1)
    assert torch.allclose(y, ref, atol=1e-12)
    dy = torch.randn(ref.shape, generator=g, dtype=torch.float64)
    ga = torch.autograd.grad(y, (x, w, b), dy)
    gb = torch.autograd.grad(ref, (x, w, b), dy)
    for a, r in zip(ga, gb):
        assert torch.allclose(a, r, atol=1e-11)


@pytest.mark.parametrize("stride", [1, 2])
def test_cuda_kernel_restatements_agree_with_vectorised_form(stride):
    g = torch.Generator().manual_seed(1)
    B, C, H, W, Co = 2, 3, 6, 7, 4
    Ho, Wo = D._geom(H, W, 3, 3, stride, 1, 1)
    x = torch.randn(B, C, H, W, generator=g, dtype=torch.float64, requires_grad=True)
    off = (torch.randn(B, 18, Ho, Wo, generator=g, dtype=torch.float64) * 1.5).requires_grad_()     # some samples leave the map
    mask = torch.rand(B, 9, Ho, Wo, generator=g, dtype=torch.float64).requires_grad_()
    w = torch.randn(Co, C, 3, 3, generator=g, dtype=torch.float64)
    cols = D.sample_columns(x, off, mask, 3, 3, stride, 1)                       # [B, C, T, P]
    loops = D.im2col_loops(x.detach().numpy(), off.detach().numpy(), mask.detach().numpy(), 3, 3, stride, 1)   # [C*T, B, Ho, Wo]
    assert np.allclose(cols.detach().numpy().transpose(1, 2, 0, 3).reshape(C * 9, B, Ho, Wo), loops, atol=1e-12)
    dcols = torch.randn(cols.shape, generator=g, dtype=torch.float64)
    gx, goff, gmask = torch.autograd.grad(cols, (x, off, mask), dcols)
    dc = dcols.numpy().transpose(1, 2, 0, 3).reshape(C * 9, B, Ho, Wo)
    assert np.allclose(D.col2im_loops(dc, off.detach().numpy(), mask.detach().numpy(), C, H, W, 3, 3, stride, 1), gx.numpy(), atol=1e-11)
    doff, dmask = D.col2im_coord_loops(dc, x.detach().numpy(), off.detach().numpy(), mask.detach().numpy(), 3, 3, stride, 1)
    assert np.allclose(doff, goff.numpy(), atol=1e-11)
    assert np.allclose(dmask, gmask.numpy(), atol=1e-11)
    # the module wrapper is the same function
    m = D.ModulatedDeformConv(C, Co, 3, stride=stride, padding=1).double()
    with torch.no_grad():
        m.weight.copy_(w)
    assert torch.equal(m(x, off, mask), D.modulated_deform_conv(x, off, mask, w, m.bias, stride, 1))
