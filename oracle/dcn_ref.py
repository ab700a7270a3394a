"""TEST INFRASTRUCTURE ONLY -- CPU restatement of the reference's modulated deformable convolution (DCNv2) for the DyHead of the
fine-grained model.  Imported by tests/, oracle/gen_dyhead_golden.py and nothing else; the product path (fiber_amd/) never
touches it.

What it restates (all under /root/reference/fine_grained/maskrcnn_benchmark):
  csrc/cuda/deform_conv_kernel_cuda.cu:474-503   dmcn_im2col_bilinear            -> bilinear()
  csrc/cuda/deform_conv_kernel_cuda.cu:578-640   modulated_deformable_im2col     -> im2col_loops()
  csrc/cuda/deform_conv_kernel_cuda.cu:505-534, 643-700   col2im (+ gradient weight)   -> col2im_loops()
  csrc/cuda/deform_conv_kernel_cuda.cu:536-575, 703-773   col2im_coord (+ coordinate weight) -> col2im_coord_loops()
  csrc/cuda/deform_conv_cuda.cu:497-572          per-image im2col + addmm        -> modulated_deform_conv()
  layers/deform_conv.py:300-353                  ModulatedDeformConv             -> class ModulatedDeformConv

PARITY UNPINNED against the reference binary: the algorithm lives in CUDA sources that need nvcc and torch's CUDA extension
headers (fine_grained/setup.py CUDAExtension), so it cannot be built or run in this image, and the reference holds no test or
golden vector for it.  What pins this file instead:
  (1) with zero offsets and a unit mask the operator must equal torch.nn.functional.conv2d (an independent implementation) --
      forward, input gradient and weight gradient (tests/test_dcn_oracle.py);
  (2) the explicit-loop restatements of the three CUDA kernels (one Python loop iteration per CUDA thread, same index
      arithmetic) agree with the vectorised, differentiable form used for everything else: im2col with the forward, col2im and
      col2im_coord with torch.autograd through it.
"""
import math

import numpy as np
import torch
import torch.nn as nn


# ---------------------------------------------------------------------------------------------------------------------------
# explicit loops: one iteration per CUDA thread of the reference kernels (small cases only)
def bilinear(plane, h, w):
    """deform_conv_kernel_cuda.cu:474-503 (called only when -1 < h < H and -1 < w < W)."""
    H, W = plane.shape
    h_low, w_low = math.floor(h), math.floor(w)
    h_high, w_high = h_low + 1, w_low + 1
    lh, lw = h - h_low, w - w_low
    hh, hw = 1 - lh, 1 - lw
    v1 = plane[h_low, w_low] if (h_low >= 0 and w_low >= 0) else 0.0
    v2 = plane[h_low, w_high] if (h_low >= 0 and w_high <= W - 1) else 0.0
    v3 = plane[h_high, w_low] if (h_high <= H - 1 and w_low >= 0) else 0.0
    v4 = plane[h_high, w_high] if (h_high <= H - 1 and w_high <= W - 1) else 0.0
    return hh * hw * v1 + hh * lw * v2 + lh * hw * v3 + lh * lw * v4


def _geom(H, W, kh, kw, stride, pad, dil):
    Ho = (H + 2 * pad - (dil * (kh - 1) + 1)) // stride + 1
    Wo = (W + 2 * pad - (dil * (kw - 1) + 1)) // stride + 1
    return Ho, Wo


def im2col_loops(x, offset, mask, kh, kw, stride, pad, dil=1):
    """deform_conv_kernel_cuda.cu:578-640.  x [B,C,H,W], offset [B,2*kh*kw,Ho,Wo], mask [B,kh*kw,Ho,Wo] (numpy, float64)
    -> cols [C*kh*kw, B, Ho, Wo]."""
    B, C, H, W = x.shape
    Ho, Wo = _geom(H, W, kh, kw, stride, pad, dil)
    cols = np.zeros((C * kh * kw, B, Ho, Wo))
    for c in range(C):
        for b in range(B):
            for ho in range(Ho):
                for wo in range(Wo):
                    h_in, w_in = ho * stride - pad, wo * stride - pad
                    for i in range(kh):
                        for j in range(kw):
                            oh = offset[b, 2 * (i * kw + j), ho, wo]
                            ow = offset[b, 2 * (i * kw + j) + 1, ho, wo]
                            m = mask[b, i * kw + j, ho, wo]
                            h_im, w_im = h_in + i * dil + oh, w_in + j * dil + ow
                            val = 0.0
                            if h_im > -1 and w_im > -1 and h_im < H and w_im < W:
                                val = bilinear(x[b, c], h_im, w_im)
                            cols[c * kh * kw + i * kw + j, b, ho, wo] = val * m
    return cols


def _gradient_weight(ah, aw, h, w, H, W):
    """dmcn_get_gradient_weight, deform_conv_kernel_cuda.cu:505-534."""
    if ah <= -1 or ah >= H or aw <= -1 or aw >= W:
        return 0.0
    hl, wl = math.floor(ah), math.floor(aw)
    hh, wh = hl + 1, wl + 1
    wt = 0.0
    if h == hl and w == wl:
        wt = (h + 1 - ah) * (w + 1 - aw)
    if h == hl and w == wh:
        wt = (h + 1 - ah) * (aw + 1 - w)
    if h == hh and w == wl:
        wt = (ah + 1 - h) * (w + 1 - aw)
    if h == hh and w == wh:
        wt = (ah + 1 - h) * (aw + 1 - w)
    return wt


def col2im_loops(dcols, offset, mask, C, H, W, kh, kw, stride, pad, dil=1):
    """deform_conv_kernel_cuda.cu:643-700: gradient to the input map.  dcols [C*kh*kw, B, Ho, Wo] -> dx [B,C,H,W]."""
    _, B, Ho, Wo = dcols.shape
    dx = np.zeros((B, C, H, W))
    for c in range(C):
        for i in range(kh):
            for j in range(kw):
                for b in range(B):
                    for ho in range(Ho):
                        for wo in range(Wo):
                            h_in, w_in = ho * stride - pad, wo * stride - pad
                            oh = offset[b, 2 * (i * kw + j), ho, wo]
                            ow = offset[b, 2 * (i * kw + j) + 1, ho, wo]
                            m = mask[b, i * kw + j, ho, wo]
                            ch, cw = h_in + i * dil + oh, w_in + j * dil + ow
                            top = dcols[(c * kh + i) * kw + j, b, ho, wo] * m
                            cur_h, cur_w = int(ch), int(cw)                    # C cast: truncation toward zero
                            for dy in range(-2, 3):
                                for dxx in range(-2, 3):
                                    y, xx = cur_h + dy, cur_w + dxx
                                    if 0 <= y < H and 0 <= xx < W and abs(ch - y) < 1 and abs(cw - xx) < 1:
                                        dx[b, c, y, xx] += _gradient_weight(ch, cw, y, xx, H, W) * top
    return dx


def _coordinate_weight(ah, aw, H, W, plane, bp_dir):
    """dmcn_get_coordinate_weight, deform_conv_kernel_cuda.cu:536-575."""
    if ah <= -1 or ah >= H or aw <= -1 or aw >= W:
        return 0.0
    hl, wl = math.floor(ah), math.floor(aw)
    hh, wh = hl + 1, wl + 1
    wt = 0.0
    if bp_dir == 0:
        if hl >= 0 and wl >= 0:
            wt += -1 * (wl + 1 - aw) * plane[hl, wl]
        if hl >= 0 and wh <= W - 1:
            wt += -1 * (aw - wl) * plane[hl, wh]
        if hh <= H - 1 and wl >= 0:
            wt += (wl + 1 - aw) * plane[hh, wl]
        if hh <= H - 1 and wh <= W - 1:
            wt += (aw - wl) * plane[hh, wh]
    else:
        if hl >= 0 and wl >= 0:
            wt += -1 * (hl + 1 - ah) * plane[hl, wl]
        if hl >= 0 and wh <= W - 1:
            wt += (hl + 1 - ah) * plane[hl, wh]
        if hh <= H - 1 and wl >= 0:
            wt += -1 * (ah - hl) * plane[hh, wl]
        if hh <= H - 1 and wh <= W - 1:
            wt += (ah - hl) * plane[hh, wh]
    return wt


def col2im_coord_loops(dcols, x, offset, mask, kh, kw, stride, pad, dil=1):
    """deform_conv_kernel_cuda.cu:703-773 (deformable_group = 1): gradients to the offsets and the mask."""
    B, C, H, W = x.shape
    _, _, Ho, Wo = dcols.shape
    T = kh * kw
    doff, dmask = np.zeros((B, 2 * T, Ho, Wo)), np.zeros((B, T, Ho, Wo))
    for b in range(B):
        for oc in range(2 * T):
            for ho in range(Ho):
                for wo in range(Wo):
                    val = mval = 0.0
                    t, bp_dir = oc // 2, oc % 2
                    i, j = t // kw, t % kw
                    for c in range(C):                                        # col_c = t, t + T, ...: channel c, tap t
                        h_in, w_in = ho * stride - pad, wo * stride - pad
                        inv_h = h_in + i * dil + offset[b, 2 * t, ho, wo]
                        inv_w = w_in + j * dil + offset[b, 2 * t + 1, ho, wo]
                        m = mask[b, t, ho, wo]
                        dc = dcols[c * T + t, b, ho, wo]
                        if inv_h <= -1 or inv_w <= -1 or inv_h >= H or inv_w >= W:
                            inv_h = inv_w = -2
                        else:
                            mval += dc * bilinear(x[b, c], inv_h, inv_w)
                        val += _coordinate_weight(inv_h, inv_w, H, W, x[b, c], bp_dir) * dc * m
                    doff[b, oc, ho, wo] = val
                    if bp_dir == 0:
                        dmask[b, t, ho, wo] = mval
    return doff, dmask


# ---------------------------------------------------------------------------------------------------------------------------
# vectorised, differentiable form (torch; any float dtype) -- the oracle proper
def sample_columns(x, offset, mask, kh, kw, stride, pad, dil=1):
    """The im2col of deform_conv_kernel_cuda.cu:578-640 for the whole batch: [B, C, kh*kw, Ho*Wo]."""
    B, C, H, W = x.shape
    Ho, Wo = _geom(H, W, kh, kw, stride, pad, dil)
    dev, dt = x.device, x.dtype
    ho = torch.arange(Ho, device=dev, dtype=dt).view(1, Ho, 1) * stride - pad
    wo = torch.arange(Wo, device=dev, dtype=dt).view(1, 1, Wo) * stride - pad
    flat = x.reshape(B, C, H * W)
    out = []
    for t in range(kh * kw):
        i, j = t // kw, t % kw
        h = ho + i * dil + offset[:, 2 * t]
        w = wo + j * dil + offset[:, 2 * t + 1]
        inside = (h > -1) & (w > -1) & (h < H) & (w < W)
        h0, w0 = torch.floor(h), torch.floor(w)
        lh, lw = h - h0, w - w0
        h0, w0 = h0.long(), w0.long()
        val = 0
        for dh, dw, wt in ((0, 0, (1 - lh) * (1 - lw)), (0, 1, (1 - lh) * lw), (1, 0, lh * (1 - lw)), (1, 1, lh * lw)):
            hc, wc = h0 + dh, w0 + dw
            ok = inside & (hc >= 0) & (hc <= H - 1) & (wc >= 0) & (wc <= W - 1)
            idx = (hc.clamp(0, H - 1) * W + wc.clamp(0, W - 1)).view(B, 1, Ho * Wo).expand(B, C, Ho * Wo)
            v = flat.gather(2, idx) * ok.view(B, 1, Ho * Wo).to(dt)
            val = val + v * wt.view(B, 1, Ho * Wo)
        out.append(val * mask[:, t].reshape(B, 1, Ho * Wo))
    return torch.stack(out, 2)


def modulated_deform_conv(input, offset, mask, weight, bias=None, stride=1, padding=0, dilation=1, groups=1, deformable_groups=1):
    """deform_conv_cuda.cu:497-572 (columns, then weight.flatten(1) @ columns per image, plus bias) for groups = deformable_groups = 1."""
    assert groups == 1 and deformable_groups == 1, "DyHead instantiates the default groups (layers/dyhead.py:14)"
    Cout, C, kh, kw = weight.shape
    B, _, H, W = input.shape
    Ho, Wo = _geom(H, W, kh, kw, stride, padding, dilation)
    cols = sample_columns(input, offset, mask, kh, kw, stride, padding, dilation)       # [B, C, T, P]
    out = torch.einsum("oct,bctp->bop", weight.reshape(Cout, C, kh * kw), cols)
    if bias is not None:
        out = out + bias.view(1, -1, 1)
    return out.view(B, Cout, Ho, Wo)


def as_read_by_kernel(t, Ho, Wo):
    """What the reference kernels read when the offset / mask maps are LARGER than the convolution's output -- which DyConv does on
    purpose (layers/dyhead.py:93-101: the offsets computed from level l are also handed to the stride-1 convolution of level l+1,
    whose output is half the size).  Nothing checks the sizes; the kernels take `offset[b]` as a bare pointer and index it with the
    OUTPUT geometry, ((2*tap) * height_col + h) * width_col + w (deform_conv_kernel_cuda.cu:598-609, 663-668, 743-748), i.e. they
    reinterpret the first channels*Ho*Wo floats of the contiguous [channels, H', W'] block as [channels, Ho, Wo].  The backward
    writes the same prefix of a zero-initialised gradient (deform_conv_cuda.cu:619, 770-779), which is what autograd through
    this slice gives."""
    if tuple(t.shape[2:]) == (Ho, Wo):
        return t
    B, ch = t.shape[:2]
    assert t.shape[2] * t.shape[3] >= Ho * Wo, "the reference would read past the buffer"
    return t.reshape(B, -1)[:, : ch * Ho * Wo].reshape(B, ch, Ho, Wo)


class ModulatedDeformConv(nn.Module):
    """layers/deform_conv.py:300-353: same constructor, parameters (`weight`, `bias`), initialisation and forward signature."""

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1, groups=1, deformable_groups=1, bias=True):
        super().__init__()
        ks = (kernel_size, kernel_size) if isinstance(kernel_size, int) else tuple(kernel_size)
        self.in_channels, self.out_channels, self.kernel_size = in_channels, out_channels, ks
        self.stride, self.padding, self.dilation, self.groups, self.deformable_groups = stride, padding, dilation, groups, deformable_groups
        self.with_bias = bias
        self.weight = nn.Parameter(torch.Tensor(out_channels, in_channels // groups, *ks))
        if bias:
            self.bias = nn.Parameter(torch.Tensor(out_channels))
        else:
            self.register_parameter("bias", None)
        n = in_channels * ks[0] * ks[1]
        self.weight.data.uniform_(-1.0 / math.sqrt(n), 1.0 / math.sqrt(n))
        if self.bias is not None:
            self.bias.data.zero_()

    def forward(self, input, offset, mask):
        dt = self.weight.dtype                                                  # reference: custom_fwd(cast_inputs=float32)
        Ho, Wo = _geom(input.shape[2], input.shape[3], *self.kernel_size, self.stride, self.padding, self.dilation)
        offset, mask = as_read_by_kernel(offset, Ho, Wo), as_read_by_kernel(mask, Ho, Wo)
        return modulated_deform_conv(input.to(dt), offset.to(dt), mask.to(dt), self.weight, self.bias, self.stride, self.padding,
                                     self.dilation, self.groups, self.deformable_groups)
