"""DyHead tower of the fine-grained model on the MI355X kernels (the deformable-convolution half).

Mirrors fine_grained/maskrcnn_benchmark/layers/dyhead.py (Conv3x3Norm :9-27, DyConv :30-117, DyHead :120-149), layers/dyrelu.py
(h_sigmoid :29-36, DYReLU :38-128) and layers/deform_conv.py:300-353 (ModulatedDeformConv): same constructors, forward signatures
(lists of NCHW level maps in and out) and parameter names, hence the same checkpoint keys (`dyhead_tower.<i>.DyConv.<j>.conv.weight`,
`...bn.weight`, `...offset.weight`, `...AttnConv.1.weight`, `...relu.fc.0.weight`).

Every 3x3 convolution of the tower -- the modulated deformable ones, the 27-channel offset / mask predictor and the plain ones of
the non-deformable configuration -- runs as ONE channels-last gather launch (csrc/dcn.hip) + ONE hand-written MFMA GEMM over the
whole batch (ops.deform_conv; the reference loops over images around a per-image column buffer and a library GEMM).  Level maps
are converted to channels-last bf16 once per DyConv.  The small per-level glue (GroupNorm statistics, the 1x1 scale attention on
pooled features, DYReLU's two-layer MLP on pooled features, bilinear up-sampling of the coarser level) is plain torch on the
device: a few per-cent of the tower's bytes and no kernels of its own in the reference either.

The reference hands the offsets / masks predicted at level l also to the stride-1 convolution of level l+1 (half the size) and
its kernels index them with the OUTPUT geometry; `_as_read_by_kernel` reproduces that read (see oracle/dcn_ref.py for the
line-by-line account) so results stay identical.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import ops

BF16 = torch.bfloat16


def _nhwc(x):
    """NCHW (any float dtype) -> contiguous channels-last bf16 [B, H, W, C]"""
    return x.permute(0, 2, 3, 1).to(BF16).contiguous()


def _nchw(y):
    return y.permute(0, 3, 1, 2).float()


def _as_read_by_kernel(t, Ho, Wo):
    """[B, ch, H', W'] -> what deform_conv_kernel_cuda.cu:598-609 reads for an Ho x Wo output: the first ch*Ho*Wo values of each
    image's contiguous block, viewed as [ch, Ho, Wo]."""
    if tuple(t.shape[2:]) == (Ho, Wo):
        return t
    B, ch = t.shape[:2]
    if t.shape[2] * t.shape[3] < Ho * Wo:
        raise ValueError("offset / mask maps smaller than the convolution output (the reference would read out of bounds)")
    return t.contiguous().reshape(B, -1)[:, : ch * Ho * Wo].reshape(B, ch, Ho, Wo)


def _rows(t):
    """[B, ch, Ho, Wo] -> fp32 [B*Ho*Wo, ch] (the kernel's per-position records)"""
    return t.permute(0, 2, 3, 1).reshape(-1, t.shape[1]).float().contiguous()


class ModulatedDeformConv(nn.Module):
    """layers/deform_conv.py:300-353.  forward(input NCHW, offset [B, 2*kh*kw, H', W'], mask [B, kh*kw, H', W']) -> NCHW fp32.
    `input` may also be given channels-last bf16 through forward_nhwc (what DyConv does)."""

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1, groups=1, deformable_groups=1, bias=True):
        super().__init__()
        ks = (kernel_size, kernel_size) if isinstance(kernel_size, int) else tuple(kernel_size)
        if dilation != 1 or groups != 1 or deformable_groups != 1:
            raise NotImplementedError("DyHead instantiates dilation = groups = deformable_groups = 1 (layers/dyhead.py:14)")
        self.in_channels, self.out_channels, self.kernel_size = in_channels, out_channels, ks
        self.stride, self.padding, self.dilation, self.groups, self.deformable_groups = stride, padding, dilation, groups, deformable_groups
        self.with_bias = bias
        self.weight = nn.Parameter(torch.Tensor(out_channels, in_channels // groups, *ks))
        if bias:
            self.bias = nn.Parameter(torch.Tensor(out_channels))
        else:
            self.register_parameter("bias", None)
        self.reset_parameters()

    def reset_parameters(self):
        n = self.in_channels * self.kernel_size[0] * self.kernel_size[1]
        self.weight.data.uniform_(-n ** -0.5, n ** -0.5)
        if self.bias is not None:
            self.bias.data.zero_()

    def forward_nhwc(self, x, offset, mask):
        B, H, W, _ = x.shape
        Ho, Wo = ops._conv_out(H, self.kernel_size[0], self.stride, self.padding), ops._conv_out(W, self.kernel_size[1], self.stride, self.padding)
        off = _rows(_as_read_by_kernel(offset, Ho, Wo))
        msk = _rows(_as_read_by_kernel(mask, Ho, Wo))
        return ops.deform_conv(x, off, msk, self.weight, self.bias, self.stride, self.padding)

    def forward(self, input, offset, mask):
        return _nchw(self.forward_nhwc(_nhwc(input), offset, mask))


class _Conv2dNHWC(nn.Conv2d):
    """nn.Conv2d (same parameters / keys) whose 3x3 forward runs on the gather + MFMA GEMM path."""

    out_fp32 = False                                     # set on the offset / mask predictor

    def forward_nhwc(self, x, **_):
        return ops.deform_conv(x, None, None, self.weight, self.bias, self.stride[0], self.padding[0], out_fp32=self.out_fp32)

    def forward(self, input, **_):
        return _nchw(self.forward_nhwc(_nhwc(input)))


class Conv3x3Norm(nn.Module):
    def __init__(self, in_channels, out_channels, stride, deformable=False, use_gn=False):
        super().__init__()
        if deformable:
            self.conv = ModulatedDeformConv(in_channels, out_channels, kernel_size=3, stride=stride, padding=1)
        else:
            self.conv = _Conv2dNHWC(in_channels, out_channels, kernel_size=3, stride=stride, padding=1)
        self.bn = nn.GroupNorm(num_groups=16, num_channels=out_channels) if use_gn else None

    def forward_nhwc(self, x, **kwargs):
        y = _nchw(self.conv.forward_nhwc(x, **kwargs) if isinstance(self.conv, _Conv2dNHWC) else self.conv.forward_nhwc(x, kwargs["offset"], kwargs["mask"]))
        return self.bn(y) if self.bn is not None else y

    def forward(self, input, **kwargs):
        return self.forward_nhwc(_nhwc(input), **kwargs)


class h_sigmoid(nn.Module):
    def __init__(self, inplace=True, h_max=1):
        super().__init__()
        self.h_max = h_max

    def forward(self, x):
        return F.relu6(x + 3) * self.h_max / 6


class DYReLU(nn.Module):
    """layers/dyrelu.py:38-128, the configuration DyConv builds (K2, bias, no spatial branch)."""

    def __init__(self, inp, oup, reduction=4, lambda_a=1.0, K2=True, use_bias=True, use_spatial=False, init_a=(1.0, 0.0), init_b=(0.0, 0.0)):
        super().__init__()
        if not (K2 and use_bias) or use_spatial or reduction != 4:
            raise NotImplementedError("DyConv builds DYReLU(in, out) with the defaults")
        self.oup, self.lambda_a, self.exp = oup, lambda_a * 2, 4
        self.init_a, self.init_b = list(init_a), list(init_b)
        self.avg_pool = nn.AdaptiveAvgPool2d(1)
        self.fc = nn.Sequential(nn.Linear(inp, inp // reduction), nn.ReLU(inplace=True), nn.Linear(inp // reduction, oup * 4), h_sigmoid())
        self.spa = None

    def forward(self, x):
        b, c, _, _ = x.shape
        y = self.fc(self.avg_pool(x).view(b, c)).view(b, self.oup * 4, 1, 1)
        a1, b1, a2, b2 = torch.split(y, self.oup, dim=1)
        a1 = (a1 - 0.5) * self.lambda_a + self.init_a[0]
        a2 = (a2 - 0.5) * self.lambda_a + self.init_a[1]
        b1 = b1 - 0.5 + self.init_b[0]
        b2 = b2 - 0.5 + self.init_b[1]
        return torch.max(x * a1 + b1, x * a2 + b2)


class DyConv(nn.Module):
    def __init__(self, in_channels=256, out_channels=256, conv_func=Conv3x3Norm, use_dyfuse=True, use_dyrelu=False, use_deform=False):
        super().__init__()
        self.DyConv = nn.ModuleList([conv_func(in_channels, out_channels, 1), conv_func(in_channels, out_channels, 1), conv_func(in_channels, out_channels, 2)])
        if use_dyfuse:
            self.AttnConv = nn.Sequential(nn.AdaptiveAvgPool2d(1), nn.Conv2d(in_channels, 1, kernel_size=1), nn.ReLU(inplace=True))
            self.h_sigmoid = h_sigmoid()
        else:
            self.AttnConv = None
        self.relu = DYReLU(in_channels, out_channels) if use_dyrelu else nn.ReLU()
        self.offset = _Conv2dNHWC(in_channels, 27, kernel_size=3, stride=1, padding=1) if use_deform else None
        if self.offset is not None:
            self.offset.out_fp32 = True                  # sampling positions and modulation logits stay fp32
        self.init_weights()

    def init_weights(self):
        for m in self.DyConv.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.normal_(m.weight.data, 0, 0.01)
                if m.bias is not None:
                    m.bias.data.zero_()
        if self.AttnConv is not None:
            for m in self.AttnConv.modules():
                if isinstance(m, nn.Conv2d):
                    nn.init.normal_(m.weight.data, 0, 0.01)
                    if m.bias is not None:
                        m.bias.data.zero_()

    def forward(self, x):
        xs = [_nhwc(f) for f in x]                       # one layout change per level; each level feeds up to three convolutions
        next_x = []
        for level, feature in enumerate(x):
            conv_args = {}
            if self.offset is not None:
                om = _nchw(self.offset.forward_nhwc(xs[level]))
                conv_args = dict(offset=om[:, :18], mask=om[:, 18:].sigmoid())
            temp_fea = [self.DyConv[1].forward_nhwc(xs[level], **conv_args)]
            if level > 0:
                temp_fea.append(self.DyConv[2].forward_nhwc(xs[level - 1], **conv_args))
            if level < len(x) - 1:
                up = self.DyConv[0].forward_nhwc(xs[level + 1], **conv_args)
                temp_fea.append(F.interpolate(up, size=[feature.size(2), feature.size(3)], mode="bilinear", align_corners=True))
            if self.AttnConv is not None:
                res_fea = torch.stack(temp_fea)
                spa_pyr_attn = self.h_sigmoid(torch.stack([self.AttnConv(fea) for fea in temp_fea]))
                mean_fea = torch.mean(res_fea * spa_pyr_attn, dim=0, keepdim=False)
            else:
                mean_fea = torch.mean(torch.stack(temp_fea), dim=0, keepdim=False)
            next_x.append(self.relu(mean_fea))
        return next_x


class DyHead(nn.Module):
    def __init__(self, cfg, in_channels):
        super().__init__()
        self.cfg = cfg
        d = cfg.MODEL.DYHEAD
        conv_func = lambda i, o, s: Conv3x3Norm(i, o, s, deformable=d.USE_DFCONV, use_gn=d.USE_GN)   # noqa: E731
        tower = [DyConv(in_channels if i == 0 else d.CHANNELS, d.CHANNELS, conv_func=conv_func, use_dyrelu=d.USE_DYRELU,
                        use_dyfuse=d.USE_DYFUSE, use_deform=d.USE_DFCONV) for i in range(d.NUM_CONVS)]
        self.add_module("dyhead_tower", nn.Sequential(*tower))

    def forward(self, x):
        return self.dyhead_tower(x)
