"""Swin backbone with image->text cross-attention, MI355X-native.

Same module tree, constructor meaning, state-dict keys and forward signatures as the reference's
coarse_grained/fiber/modules/swin_transformer.py (WindowAttention :129, SwinTransformerBlock :264, PatchMerging :396,
BasicLayer :444, SwinTransformer :528, factories :702-789), but the forward never builds windows: every per-token op
(LayerNorm, qkv / proj / MLP GEMMs, i2t cross-attention) runs in image-token order and the cyclic shift, window
partition/reverse, relative-position-bias gather and shift mask live inside the HIP window-attention kernel
(fiber_amd/csrc/attn.hip).  Parameters are nn.Module containers only; all arithmetic goes through fiber_amd.ops.
"""
import torch
import torch.nn as nn

from .. import ops

DIM_TEXT = 768          # reference module global (swin_transformer.py:13); fiber_module's DIM_TXT typo never changes it
NUM_FUSE_BLOCK = 6


def _rel_pos_index(ws):
    r = torch.arange(ws)
    rr, cc = torch.meshgrid(r, r, indexing="ij")
    rr, cc = rr.reshape(-1), cc.reshape(-1)
    return (rr[:, None] - rr[None, :] + ws - 1) * (2 * ws - 1) + (cc[:, None] - cc[None, :] + ws - 1)


def _shift_mask(H, W, ws, shift):
    def region(n):
        lab = torch.zeros(n, dtype=torch.long)
        lab[n - ws:n - shift] = 1
        lab[n - shift:] = 2
        return lab
    lab = region(H)[:, None] * 3 + region(W)[None, :]
    lab = lab.view(H // ws, ws, W // ws, ws).permute(0, 2, 1, 3).reshape(-1, ws * ws)
    diff = lab[:, None, :] - lab[:, :, None]
    return torch.where(diff != 0, torch.tensor(-100.0), torch.tensor(0.0))


class Mlp(nn.Module):
    def __init__(self, in_features, hidden_features):
        super().__init__()
        self.fc1 = nn.Linear(in_features, hidden_features)
        self.fc2 = nn.Linear(hidden_features, in_features)


class PatchEmbed(nn.Module):
    """timm 0.4.12 PatchEmbed: Conv2d(k=s=patch) -> flatten -> LayerNorm, as im2col + MFMA GEMM + LN kernels."""

    def __init__(self, img_size=224, patch_size=4, in_chans=3, embed_dim=96):
        super().__init__()
        assert patch_size == 4 and in_chans == 3, "the HIP im2col kernel covers the 4x4 / RGB patch embed FIBER uses"
        self.img_size = (img_size, img_size)
        self.grid_size = (img_size // patch_size, img_size // patch_size)
        self.num_patches = self.grid_size[0] * self.grid_size[1]
        self.proj = nn.Conv2d(in_chans, embed_dim, kernel_size=patch_size, stride=patch_size)
        self.norm = nn.LayerNorm(embed_dim)

    def forward(self, img):
        x = ops.patch_embed_proj(img, self.proj.weight, self.proj.bias)
        # the residual stream starts here: with config["residual_dtype"] = "fp32" the LayerNorm also emits its fp32 result
        return ops.layernorm(x, self.norm.weight, self.norm.bias, self.norm.eps, want_f32=ops.residual_fp32())


class WindowAttention(nn.Module):
    def __init__(self, dim, window_size, num_heads, dim_text=None):
        super().__init__()
        ws = window_size[0] if isinstance(window_size, (tuple, list)) else window_size
        self.dim, self.window_size, self.num_heads = dim, (ws, ws), num_heads
        assert dim // num_heads == 32, "Swin head_dim is 32 for every FIBER variant (kernel contract)"
        self.scale = (dim // num_heads) ** -0.5
        self.relative_position_bias_table = nn.Parameter(torch.zeros((2 * ws - 1) ** 2, num_heads))
        self.register_buffer("relative_position_index", _rel_pos_index(ws))
        self.qkv = nn.Linear(dim, dim * 3)
        self.proj = nn.Linear(dim, dim)
        nn.init.trunc_normal_(self.relative_position_bias_table, std=0.02)
        if dim_text is not None:
            self.qkv_text_i2t = nn.Linear(dim_text, dim * 2)
            self.qkv_i2t = nn.Linear(dim, dim)
            self.proj_i2t = nn.Linear(dim, dim)
            self.alpha_i2t = nn.Parameter(torch.Tensor([0]))
            self.norm_i2t_i = nn.LayerNorm(dim)

    def forward(self, u, res, shift, shortcut=None, y=None, y_mask=None, rowscale=None, rowscale_value=None):
        """u: LayerNorm'ed tokens [B, H*W, C] in image order.  Returns [rowscale *] (proj(attn) (+ i2t branch)) (+ shortcut).
        rowscale_value = 1/keep, the one non-zero value of the DropPath factors (lets the backward fold them into its GEMMs)."""
        B, L, C = u.shape
        H, W = res
        ws = self.window_size[0]
        if ops.head_major_supported(ws):
            qkv = ops.linear_qkv_head_major(u, self.qkv.weight, self.qkv.bias, self.num_heads)
            o = ops.window_attention(qkv, self.relative_position_bias_table, B, H, W, self.num_heads, ws, shift, head_major=True)
        else:
            qkv = ops.linear(u, self.qkv.weight, self.qkv.bias)
            o = ops.window_attention(qkv, self.relative_position_bias_table, B, H, W, self.num_heads, ws, shift)
        if y is None:
            return ops.linear(o, self.proj.weight, self.proj.bias, residual=shortcut, rowscale=rowscale, rowscale_value=rowscale_value)
        a = ops.linear(o, self.proj.weight, self.proj.bias)
        S = y.shape[1]
        assert y.shape[0] == B, "text batch must match image batch"
        kv = ops.linear(y, self.qkv_text_i2t.weight, self.qkv_text_i2t.bias).view(B * S, 2 * C)
        # `a` feeds the i2t query LayerNorm AND the block's sum below: one autograd node for both uses, so that the LayerNorm backward adds
        # the other use's gradient in its own pass (no separate fan-in add over [B, L, C] in the backward)
        an, a = ops.layernorm_res(a, self.norm_i2t_i.weight, self.norm_i2t_i.bias, self.norm_i2t_i.eps)
        qi = ops.linear(an, self.qkv_i2t.weight, self.qkv_i2t.bias).view(B * L, C)
        km = y_mask.reshape(B, S) if y_mask is not None else None
        yi = ops.mha(qi, kv[:, :C], kv[:, C:], km, B, self.num_heads, self.scale)
        yi = ops.linear(yi.view(B, L, C), self.proj_i2t.weight, self.proj_i2t.bias)
        if shortcut is None:
            return ops.scale_add(a, yi, self.alpha_i2t)
        # shortcut + DropPath(a + alpha_i2t * yi) in one pass (swin_transformer.py:259 and :390); keeps the stream's fp32 payload
        return ops.stream_add(shortcut, a, b=yi, alpha=self.alpha_i2t, rowscale=rowscale)


class SwinTransformerBlock(nn.Module):
    def __init__(self, dim, input_resolution, num_heads, window_size=7, shift_size=0, mlp_ratio=4.0, drop_path=0.0,
                 dim_text=None):
        super().__init__()
        self.dim, self.input_resolution, self.num_heads = dim, input_resolution, num_heads
        self.window_size, self.shift_size = window_size, shift_size
        if min(input_resolution) <= window_size:          # swin_transformer.py:304-307
            self.shift_size, self.window_size = 0, min(input_resolution)
        assert 0 <= self.shift_size < self.window_size
        self.norm1 = nn.LayerNorm(dim)
        self.attn = WindowAttention(dim, self.window_size, num_heads, dim_text)
        self.drop_path_rate = float(drop_path)
        self.norm2 = nn.LayerNorm(dim)
        self.mlp = Mlp(dim, int(dim * mlp_ratio))
        H, W = input_resolution
        self.register_buffer("attn_mask", _shift_mask(H, W, self.window_size, self.shift_size) if self.shift_size > 0 else None)

    def forward(self, x, y=None, y_mask=None):
        H, W = self.input_resolution
        B, L, C = x.shape
        assert L == H * W, "input feature has wrong size"
        dp = self.drop_path_rate if self.training else 0.0
        # two independent DropPath draws per block (swin_transformer.py:390-391), folded into the GEMM epilogues
        s1 = ops.drop_path_scale(B, dp, x.device) if dp > 0.0 else None
        s2 = ops.drop_path_scale(B, dp, x.device) if dp > 0.0 else None
        u, x = ops.layernorm_res(x, self.norm1.weight, self.norm1.bias, self.norm1.eps)
        rv = 1.0 / (1.0 - dp) if dp > 0.0 else None
        x = self.attn(u, (H, W), self.shift_size, shortcut=x, y=y, y_mask=y_mask, rowscale=s1, rowscale_value=rv)
        # norm2 -> Mlp -> DropPath -> residual (swin_transformer.py:391): one kernel per direction at C = 128 / 256 (csrc/mlp_rows.hip)
        return ops.ln_mlp(x, self.norm2.weight, self.norm2.bias, self.norm2.eps, self.mlp.fc1.weight, self.mlp.fc1.bias,
                          self.mlp.fc2.weight, self.mlp.fc2.bias, rowscale=s2, rowscale_value=rv)


class PatchMerging(nn.Module):
    def __init__(self, input_resolution, dim):
        super().__init__()
        self.input_resolution, self.dim = input_resolution, dim
        self.reduction = nn.Linear(4 * dim, 2 * dim, bias=False)
        self.norm = nn.LayerNorm(4 * dim)

    def forward(self, x):
        H, W = self.input_resolution
        B, L, C = x.shape
        assert L == H * W and H % 2 == 0 and W % 2 == 0
        z = ops.patch_merge_ln(x, self.norm.weight, self.norm.bias, H, W, self.norm.eps)
        return ops.start_stream(ops.linear(z, self.reduction.weight))      # a new stage's residual stream begins


class BasicLayer(nn.Module):
    def __init__(self, dim, input_resolution, depth, num_heads, window_size, mlp_ratio=4.0, drop_path=0.0,
                 downsample=None, dim_text=None, layer_index=0):
        super().__init__()
        self.dim, self.input_resolution, self.depth = dim, input_resolution, depth
        self.blocks = nn.ModuleList([
            SwinTransformerBlock(
                dim=dim, input_resolution=input_resolution, num_heads=num_heads, window_size=window_size,
                shift_size=0 if (i % 2 == 0) else window_size // 2, mlp_ratio=mlp_ratio,
                drop_path=drop_path[i] if isinstance(drop_path, list) else drop_path,
                dim_text=None if layer_index == 2 and i < 20 - NUM_FUSE_BLOCK else dim_text)   # swin_transformer.py:502
            for i in range(depth)])
        self.downsample = downsample(input_resolution, dim=dim) if downsample is not None else None

    def forward(self, x, y=None, y_mask=None):
        for blk in self.blocks:
            x = blk(x, y, y_mask)
        return self.downsample(x) if self.downsample is not None else x


class SwinTransformer(nn.Module):
    def __init__(self, img_size=224, patch_size=4, in_chans=3, embed_dim=96, depths=(2, 2, 6, 2), num_heads=(3, 6, 12, 24),
                 window_size=7, mlp_ratio=4.0, drop_path_rate=0.1, **kwargs):
        super().__init__()
        window_size = int(img_size / 32)                  # swin_transformer.py:575
        self.num_layers = len(depths)
        self.embed_dim = embed_dim
        self.num_features = int(embed_dim * 2 ** (self.num_layers - 1))
        self.patch_embed = PatchEmbed(img_size, patch_size, in_chans, embed_dim)
        self.patch_grid = self.patch_embed.grid_size
        self.absolute_pos_embed = None
        self.pos_drop = nn.Identity()                     # drop_rate = 0.0 in every FIBER config
        dpr = [v.item() for v in torch.linspace(0, drop_path_rate, sum(depths))]
        self.layers = nn.Sequential(*[
            BasicLayer(dim=int(embed_dim * 2 ** i),
                       input_resolution=(self.patch_grid[0] // 2 ** i, self.patch_grid[1] // 2 ** i),
                       depth=depths[i], num_heads=num_heads[i], window_size=window_size, mlp_ratio=mlp_ratio,
                       drop_path=dpr[sum(depths[:i]):sum(depths[:i + 1])],
                       downsample=PatchMerging if i < self.num_layers - 1 else None,
                       dim_text=DIM_TEXT if i >= 2 else None, layer_index=i)
            for i in range(self.num_layers)])
        self.norm = nn.LayerNorm(self.num_features)
        self.avgpool = nn.AdaptiveAvgPool1d(1)
        self.apply(_init_vit_weights)

    def forward_features(self, x, y=None, y_mask=None):
        x = self.patch_embed(x)
        for layer in self.layers:
            x = layer(x, y, y_mask)
        return ops.layernorm(x, self.norm.weight, self.norm.bias, self.norm.eps)

    def forward(self, x, y=None, y_mask=None):
        return self.forward_features(x, y, y_mask)


def _init_vit_weights(m):
    if isinstance(m, nn.Linear):
        nn.init.trunc_normal_(m.weight, std=0.02)
        if m.bias is not None:
            nn.init.zeros_(m.bias)
    elif isinstance(m, nn.LayerNorm):
        nn.init.zeros_(m.bias)
        nn.init.ones_(m.weight)


_ARCH = {
    "swin_base_patch4_window12_384": dict(embed_dim=128, depths=(2, 2, 18, 2), num_heads=(4, 8, 16, 32)),
    "swin_base_patch4_window7_224": dict(embed_dim=128, depths=(2, 2, 18, 2), num_heads=(4, 8, 16, 32)),
    "swin_large_patch4_window12_384": dict(embed_dim=192, depths=(2, 2, 18, 2), num_heads=(6, 12, 24, 48)),
    "swin_large_patch4_window7_224": dict(embed_dim=192, depths=(2, 2, 18, 2), num_heads=(6, 12, 24, 48)),
    "swin_small_patch4_window7_224": dict(embed_dim=96, depths=(2, 2, 18, 2), num_heads=(3, 6, 12, 24)),
    "swin_tiny_patch4_window7_224": dict(embed_dim=96, depths=(2, 2, 6, 2), num_heads=(3, 6, 12, 24)),
}


def _make_factory(name):
    arch = _ARCH[name.replace("_in22k", "")]

    def factory(pretrained=False, **kwargs):
        if pretrained:
            raise RuntimeError("pretrained Swin weights need a network download; load a checkpoint via config['load_path']")
        cfg = kwargs.pop("config")
        return SwinTransformer(img_size=cfg["image_size"], patch_size=4, **arch, **kwargs)
    factory.__name__ = name
    return factory


for _n in list(_ARCH) + [n + "_in22k" for n in _ARCH if "small" not in n and "tiny" not in n]:
    globals()[_n] = _make_factory(_n)
