"""Fine-grained fused backbone, MI355X-native (BASELINE.json configs[4] -- the backbone half).

Mirrors `FusionSwinTransformer` of fine_grained/maskrcnn_benchmark/modeling/backbone/fusion_swin_transformer_v2.py:803-945 (with
its SwinTransformer :569-711, BasicLayer :402-524, SwinTransformerBlock :233-345, WindowAttention :76-230, PatchMerging :348-385,
PatchEmbed :527-566) and the text layers of language_backbone/roberta_fused_model_v2.py: same constructor meaning, forward
signature `(tokenizer_input, images) -> (visual_features, language_dict_features, None)`, parameter names
(`backbone.body.*`, `language_backbone.body.model.*`) and therefore checkpoint keys.  It is built from the SAME kernels as the
coarse path (fiber_amd.ops): what the detection variant adds is geometry, not arithmetic --

  * dynamic H x W with a fixed 12 x 12 window: the LayerNorm'ed tokens are zero-padded to a window multiple, the window kernel
    runs on the padded grid (roll, partition, shift mask of the PADDED grid and relative-position bias are index arithmetic
    inside it), and the result is cropped BEFORE the per-token ops that follow (proj, i2t cross-attention), which is the same
    function because those are per-token;
  * odd blocks always shift, also when the padded grid is a single window;
  * the i2t query is the projected self-attention output with no LayerNorm in front (:205-215);
  * each stage emits a LayerNorm'ed NCHW map (norm0 = identity for the RETINANET arch) for the FPN, which -- like the DyHead and
    the detection losses -- is outside this path (`fpn` may be passed in; default: the stage maps are returned as they are).
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import ops
from . import roberta as RB
from .swin_transformer import Mlp, _rel_pos_index


class WindowAttention(nn.Module):
    def __init__(self, dim, window_size, num_heads, dim_text=None):
        super().__init__()
        self.dim, self.ws, self.num_heads = dim, window_size, num_heads
        assert dim // num_heads == 32, "Swin head_dim is 32 (window-kernel contract)"
        self.scale = 32 ** -0.5
        self.relative_position_bias_table = nn.Parameter(torch.zeros((2 * window_size - 1) ** 2, num_heads))
        self.register_buffer("relative_position_index", _rel_pos_index(window_size))
        self.qkv = nn.Linear(dim, dim * 3)
        self.proj = nn.Linear(dim, dim)
        nn.init.trunc_normal_(self.relative_position_bias_table, std=0.02)
        if dim_text is not None:
            self.qkv_text_i2t = nn.Linear(dim_text, dim * 2)
            self.qkv_i2t = nn.Linear(dim, dim)
            self.proj_i2t = nn.Linear(dim, dim)
            self.alpha_i2t = nn.Parameter(torch.Tensor([0]))

    def forward(self, u, H, W, shift, shortcut, y=None, y_mask=None, rowscale=None, rowscale_value=None):
        """u: LayerNorm'ed tokens [B, H*W, C] in image order -> shortcut + [rowscale *] branch."""
        B, L, C = u.shape
        ws = self.ws
        pr, pb = (ws - W % ws) % ws, (ws - H % ws) % ws
        Hp, Wp = H + pb, W + pr
        if pr or pb:                                              # zeros AFTER norm1 (:303-308): padded tokens become qkv.bias
            u = F.pad(u.view(B, H, W, C), (0, 0, 0, pr, 0, pb)).reshape(B, Hp * Wp, C)
        hm = ops.head_major_supported(ws)
        if hm:
            qkv = ops.linear_qkv_head_major(u, self.qkv.weight, self.qkv.bias, self.num_heads)
        else:
            qkv = ops.linear(u, self.qkv.weight, self.qkv.bias)
        o = ops.window_attention(qkv, self.relative_position_bias_table, B, Hp, Wp, self.num_heads, ws, shift, head_major=hm)
        if pr or pb:
            o = o.view(B, Hp, Wp, C)[:, :H, :W].reshape(B, L, C)   # crop (:331-332), moved ahead of the per-token ops
        if y is None:
            return ops.linear(o, self.proj.weight, self.proj.bias, residual=shortcut, rowscale=rowscale, rowscale_value=rowscale_value)
        a = ops.linear(o, self.proj.weight, self.proj.bias)
        S = y.shape[1]
        kv = ops.linear(y, self.qkv_text_i2t.weight, self.qkv_text_i2t.bias).view(B * S, 2 * C)
        qi = ops.linear(a, self.qkv_i2t.weight, self.qkv_i2t.bias).view(B * L, C)
        km = y_mask.reshape(B, S) if y_mask is not None else None
        yi = ops.mha(qi, kv[:, :C], kv[:, C:], km, B, self.num_heads, self.scale)
        yi = ops.linear(yi.view(B, L, C), self.proj_i2t.weight, self.proj_i2t.bias)
        return ops.stream_add(shortcut, a, b=yi, alpha=self.alpha_i2t, rowscale=rowscale)


class SwinTransformerBlock(nn.Module):
    def __init__(self, dim, num_heads, window_size=12, shift_size=0, mlp_ratio=4.0, drop_path=0.0, dim_text=None):
        super().__init__()
        self.dim, self.window_size, self.shift_size, self.drop_path_rate = dim, window_size, shift_size, float(drop_path)
        self.norm1 = nn.LayerNorm(dim)
        self.attn = WindowAttention(dim, window_size, num_heads, dim_text)
        self.norm2 = nn.LayerNorm(dim)
        self.mlp = Mlp(dim, int(dim * mlp_ratio))
        self.H = self.W = None

    def forward(self, x, mask_matrix=None, x_text=None, mask_text=None):
        """`mask_matrix` is accepted for signature parity (:293) and ignored: the kernel derives the shift mask from (Hp, Wp)."""
        H, W = self.H, self.W
        B, L, C = x.shape
        assert L == H * W, "input feature has wrong size"
        dp = self.drop_path_rate if self.training else 0.0
        s1 = ops.drop_path_scale(B, dp, x.device) if dp > 0.0 else None
        s2 = ops.drop_path_scale(B, dp, x.device) if dp > 0.0 else None
        rv = 1.0 / (1.0 - dp) if dp > 0.0 else None
        u, x = ops.layernorm_res(x, self.norm1.weight, self.norm1.bias, self.norm1.eps)
        x = self.attn(u, H, W, self.shift_size, x, x_text, mask_text, s1, rv)
        v, x = ops.layernorm_res(x, self.norm2.weight, self.norm2.bias, self.norm2.eps)
        return ops.mlp(v, self.mlp.fc1.weight, self.mlp.fc1.bias, self.mlp.fc2.weight, self.mlp.fc2.bias, residual=x,
                       rowscale=s2, rowscale_value=rv)


class PatchMerging(nn.Module):
    def __init__(self, dim):
        super().__init__()
        self.dim = dim
        self.reduction = nn.Linear(4 * dim, 2 * dim, bias=False)
        self.norm = nn.LayerNorm(4 * dim)

    def forward(self, x, H, W):
        assert x.shape[1] == H * W and H % 2 == 0 and W % 2 == 0, f"x size ({H}*{W}) are not even."      # as the reference (:370)
        z = ops.patch_merge_ln(x, self.norm.weight, self.norm.bias, H, W, self.norm.eps)
        return ops.linear(z, self.reduction.weight)


class BasicLayer(nn.Module):
    def __init__(self, dim, depth, num_heads, window_size=12, mlp_ratio=4.0, drop_path=0.0, downsample=None, dim_text=None):
        super().__init__()
        self.window_size, self.shift_size, self.depth = window_size, window_size // 2, depth
        self.blocks = nn.ModuleList([
            SwinTransformerBlock(dim, num_heads, window_size, 0 if i % 2 == 0 else window_size // 2, mlp_ratio,
                                 drop_path[i] if isinstance(drop_path, list) else drop_path,
                                 dim_text=(768 if i >= 14 else dim_text))                   # :458
            for i in range(depth)])
        self.downsample = downsample(dim) if downsample is not None else None

    def forward(self, x, H, W, x_text=None, mask_text=None):
        for blk in self.blocks:
            blk.H, blk.W = H, W
            x = blk(x, None, x_text, mask_text)
        if self.downsample is not None:
            return x, H, W, self.downsample(x, H, W), (H + 1) // 2, (W + 1) // 2
        return x, H, W, x, H, W


class PatchEmbed(nn.Module):
    def __init__(self, patch_size=4, in_chans=3, embed_dim=96):
        super().__init__()
        assert patch_size == 4 and in_chans == 3
        self.embed_dim = embed_dim
        self.proj = nn.Conv2d(in_chans, embed_dim, kernel_size=4, stride=4)
        self.norm = nn.LayerNorm(embed_dim)

    def forward(self, x):
        """image [B, 3, H, W] -> (LayerNorm'ed tokens [B, Wh*Ww, C], Wh, Ww); pads to a multiple of the patch (:553-557)."""
        _, _, H, W = x.shape
        if H % 4 or W % 4:
            x = F.pad(x, (0, (4 - W % 4) % 4, 0, (4 - H % 4) % 4))
        Wh, Ww = x.shape[2] // 4, x.shape[3] // 4
        t = ops.patch_embed_proj(x, self.proj.weight, self.proj.bias)
        return ops.layernorm(t, self.norm.weight, self.norm.bias, self.norm.eps), Wh, Ww


class SwinTransformer(nn.Module):
    def __init__(self, embed_dim=128, depths=(2, 2, 18, 2), num_heads=(4, 8, 16, 32), window_size=12, mlp_ratio=4.0,
                 drop_path_rate=0.2, out_features=("stage2", "stage3", "stage4", "stage5"), backbone_arch="SWINT-FPN-RETINANET"):
        super().__init__()
        self.num_layers, self.embed_dim, self.ape, self.out_features = len(depths), embed_dim, False, tuple(out_features)
        self.patch_embed = PatchEmbed(4, 3, embed_dim)
        self.pos_drop = nn.Identity()
        dpr = [v.item() for v in torch.linspace(0, drop_path_rate, sum(depths))]
        self.layers = nn.ModuleList([
            BasicLayer(int(embed_dim * 2 ** i), depths[i], num_heads[i], window_size, mlp_ratio,
                       dpr[sum(depths[:i]):sum(depths[:i + 1])], PatchMerging if i < self.num_layers - 1 else None,
                       dim_text=(768 if i == 3 else None))                                  # :676
            for i in range(self.num_layers)])
        self.num_features = [int(embed_dim * 2 ** i) for i in range(self.num_layers)]
        for i in range(self.num_layers):
            if f"stage{i + 2}" in self.out_features:
                ident = i == 0 and backbone_arch.endswith("RETINANET")                      # :689-690
                self.add_module(f"norm{i}", nn.Identity() if ident else nn.LayerNorm(self.num_features[i]))


class _FGRobertaLayer(RB.RobertaLayer):
    """roberta_fused_model_v2.py RobertaLayer: `alpha_t2i` only on cross layers (:402-404), no LayerNorm parameters inside
    `crossattention_t2i.output` (:314-316), the final LayerNorm always applied."""

    def __init__(self, config, add_cross, layer_index):
        super().__init__(config, layer_index=layer_index)
        if add_cross:
            self.crossattention_t2i.output.LayerNorm = nn.Identity()
        else:
            del self.alpha_t2i


class _FGRobertaModel(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.config = config
        RB.NUM_FUSE_BLOCK, RB.DIM_IMG = 6, 1024                    # hard-coded 512 / 1024 key-value widths (:198-202)
        self.embeddings = RB.RobertaEmbeddings(config)
        self.encoder = nn.Module()
        self.encoder.layer = nn.ModuleList([_FGRobertaLayer(config, i >= 6, i) for i in range(config.num_hidden_layers)])
        self.apply(RB.RobertaModel._init_weights)

    get_extended_attention_mask = staticmethod(RB.RobertaModel.get_extended_attention_mask)


class _Holder(nn.Module):
    pass


class FusionSwinTransformer(nn.Module):
    def __init__(self, embed_dim=128, depths=(2, 2, 18, 2), num_heads=(4, 8, 16, 32), window_size=12, drop_path_rate=0.2,
                 fpn=None, text_config=None):
        super().__init__()
        self.backbone = _Holder()
        self.backbone.body = SwinTransformer(embed_dim, depths, num_heads, window_size, drop_path_rate=drop_path_rate)
        self.backbone.fpn = fpn
        self.language_backbone = _Holder()
        self.language_backbone.body = _Holder()
        self.language_backbone.body.model = _FGRobertaModel(text_config or RB.roberta_base_config())

    def get_aggregated_output(self, features, input_ids, mask):
        """RobertaFusedEncoder.get_aggregated_output, USE_DOT_PRODUCT_TOKEN_LOSS branch (roberta_fused_model_v2.py:86-100)."""
        m = mask.to(features.dtype)
        embedded = features * m.unsqueeze(-1)
        aggregate = embedded.float().sum(1) / mask.sum(-1, keepdim=True).float()
        return {"aggregate": aggregate, "embedded": embedded, "masks": mask, "hidden": features}

    def forward(self, tokenizer_input, images):
        sw, tm = self.backbone.body, self.language_backbone.body.model
        x = images.tensors if hasattr(images, "tensors") else images
        x, Wh, Ww = sw.patch_embed(x)
        text = tm.embeddings(input_ids=tokenizer_input["input_ids"])
        ext = tm.get_extended_attention_mask(tokenizer_input["attention_mask"])
        for layer in tm.encoder.layer[:6]:                          # num_pre_text = 6 (:849)
            text = layer(text, ext)[0]
        outs = []

        def emit(i, t, H, W):
            if f"stage{i + 2}" in sw.out_features:
                n = getattr(sw, f"norm{i}")
                t = t if isinstance(n, nn.Identity) else ops.layernorm(t, n.weight, n.bias, n.eps)
                outs.append(t.view(-1, H, W, sw.num_features[i]).permute(0, 3, 1, 2).contiguous())
        for i in (0, 1):                                            # num_pre_vision = 2 (:854)
            x_out, H, W, x, Wh, Ww = sw.layers[i](x, Wh, Ww)
            emit(i, x_out, H, W)
        stage = sw.layers[2]
        for j, blk in enumerate(stage.blocks):
            blk.H, blk.W = Wh, Ww
            if j < 14:                                              # num_pre_block = 14 (:865)
                x = blk(x)
            else:                                                   # both sides read the other's PRE-block state (:876-885)
                fused = blk(x, None, text, ext)
                text = tm.encoder.layer[j - 14 + 6](text, ext, encoder_hidden_states=x)[0]
                x = fused
        emit(2, x, Wh, Ww)
        if stage.downsample is not None:
            x = stage.downsample(x, Wh, Ww)
            Wh, Ww = (Wh + 1) // 2, (Ww + 1) // 2
        stage = sw.layers[3]
        for j in (0, 1):
            blk = stage.blocks[j]
            blk.H, blk.W = Wh, Ww
            fused = blk(x, None, text, ext)
            text = tm.encoder.layer[10 + j](text, ext, encoder_hidden_states=x)[0]
            x = fused
        emit(3, x, Wh, Ww)
        lang = self.get_aggregated_output(text, tokenizer_input["input_ids"], tokenizer_input["attention_mask"])
        visual = self.backbone.fpn(outs) if self.backbone.fpn is not None else outs
        return visual, lang, None
