"""TEST INFRASTRUCTURE ONLY (never imported by fiber_amd/): fp32 PyTorch restatement of the FINE-GRAINED fused backbone
 -- `FusionSwinTransformer.forward` of
fine_grained/maskrcnn_benchmark/modeling/backbone/fusion_swin_transformer_v2.py:803-945 with its Swin pieces (WindowAttention :76-230,
SwinTransformerBlock :233-345, PatchMerging :348-385, BasicLayer :402-524 incl. get_attention_mask :470-496, PatchEmbed :527-566,
SwinTransformer :569-711) and the text layers of language_backbone/roberta_fused_model_v2.py (= the coarse RobertaLayer with the
final LayerNorm always applied, `alpha_t2i` only on layers >= 6 and no LayerNorm inside `crossattention_t2i.output`).

What differs from the coarse path restated in fiber_ref.py: dynamic H x W (the window stays 12; the LayerNorm'ed tokens are
zero-padded to a window multiple, the padded grid is rolled / partitioned / masked, the result is cropped); the i2t query is the
projected self-attention output WITHOUT a LayerNorm in front; every stage emits a normalised NCHW feature map (norm0 = identity
for the RETINANET arch); odd blocks always shift, also on a single-window grid.  Parameter names equal the reference's
(`backbone.body.*`, `language_backbone.body.model.*`).  Pinned by tests/golden/fg_*.npz, which oracle/gen_fusion_golden.py
produces by executing the reference's own files under oracle/shim.py.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import fiber_ref as R


class WindowAttention(nn.Module):
    def __init__(self, dim, ws, heads, dim_text=None):
        super().__init__()
        self.dim, self.ws, self.heads = dim, ws, heads
        self.relative_position_bias_table = nn.Parameter(torch.zeros((2 * ws - 1) ** 2, heads))
        self.register_buffer("relative_position_index", R.rel_pos_index(ws))
        self.qkv = nn.Linear(dim, 3 * dim)
        self.proj = nn.Linear(dim, dim)
        if dim_text is not None:
            self.qkv_text_i2t = nn.Linear(dim_text, 2 * dim)
            self.qkv_i2t = nn.Linear(dim, dim)
            self.proj_i2t = nn.Linear(dim, dim)
            self.alpha_i2t = nn.Parameter(torch.zeros(1))

    def forward(self, xw, mask=None, y=None, y_mask=None):
        Bw, N, C = xw.shape
        h, d = self.heads, C // self.heads
        q, k, v = self.qkv(xw).view(Bw, N, 3, h, d).permute(2, 0, 3, 1, 4)
        a = (q * d ** -0.5) @ k.transpose(-1, -2)
        bias = self.relative_position_bias_table[self.relative_position_index.reshape(-1)]
        a = a + bias.view(N, N, h).permute(2, 0, 1)[None]
        if mask is not None:
            nW = mask.shape[0]
            a = (a.view(Bw // nW, nW, h, N, N) + mask[None, :, None]).view(Bw, h, N, N)
        out = self.proj((a.softmax(-1) @ v).transpose(1, 2).reshape(Bw, N, C))
        if y is not None:
            B, S, _ = y.shape
            nW = Bw // B
            kt, vt = self.qkv_text_i2t(y).view(B, S, 2, h, d).permute(2, 0, 3, 1, 4)
            kt, vt = kt.repeat_interleave(nW, 0), vt.repeat_interleave(nW, 0)
            qi = self.qkv_i2t(out).view(Bw, N, h, d).transpose(1, 2) * d ** -0.5      # :205-215: no LayerNorm before qkv_i2t
            ai = qi @ kt.transpose(-1, -2)
            if y_mask is not None:
                ai = ai + y_mask.view(B, 1, 1, S).repeat_interleave(nW, 0)
            out = out + self.alpha_i2t * self.proj_i2t((ai.softmax(-1) @ vt).transpose(1, 2).reshape(Bw, N, C))
        return out


class SwinTransformerBlock(nn.Module):
    def __init__(self, dim, heads, ws, shift, dim_text=None):
        super().__init__()
        self.ws, self.shift = ws, shift
        self.norm1 = nn.LayerNorm(dim)
        self.attn = WindowAttention(dim, ws, heads, dim_text)
        self.norm2 = nn.LayerNorm(dim)
        self.mlp = R.Mlp(dim, 4 * dim)

    def forward(self, x, H, W, mask_matrix, x_text=None, mask_text=None):
        B, L, C = x.shape
        ws = self.ws
        u = self.norm1(x).view(B, H, W, C)
        pr, pb = (ws - W % ws) % ws, (ws - H % ws) % ws
        u = F.pad(u, (0, 0, 0, pr, 0, pb))                       # zeros AFTER the LayerNorm (:303-308)
        Hp, Wp = H + pb, W + pr
        if self.shift:
            u = torch.roll(u, (-self.shift, -self.shift), (1, 2))
        aw = self.attn(R.to_windows(u, ws), mask_matrix if self.shift else None, x_text, mask_text)
        u = R.from_windows(aw, ws, Hp, Wp)
        if self.shift:
            u = torch.roll(u, (self.shift, self.shift), (1, 2))
        x = x + u[:, :H, :W].reshape(B, L, C)
        return x + self.mlp(self.norm2(x))


class PatchMerging(nn.Module):
    def __init__(self, dim):
        super().__init__()
        self.reduction = nn.Linear(4 * dim, 2 * dim, bias=False)
        self.norm = nn.LayerNorm(4 * dim)

    def forward(self, x, H, W):
        B, L, C = x.shape
        assert H % 2 == 0 and W % 2 == 0                         # the reference asserts it too (:370)
        x = x.view(B, H, W, C)
        z = torch.cat([x[:, 0::2, 0::2], x[:, 1::2, 0::2], x[:, 0::2, 1::2], x[:, 1::2, 1::2]], -1)
        return self.reduction(self.norm(z.view(B, -1, 4 * C)))


class BasicLayer(nn.Module):
    def __init__(self, dim, depth, heads, ws, downsample, dim_text):
        super().__init__()
        self.ws, self.shift = ws, ws // 2
        self.blocks = nn.ModuleList([SwinTransformerBlock(dim, heads, ws, 0 if i % 2 == 0 else ws // 2,
                                                          dim_text=(768 if i >= 14 else dim_text)) for i in range(depth)])
        self.downsample = PatchMerging(dim) if downsample else None

    def get_attention_mask(self, H, W):
        ws = self.ws
        Hp, Wp = -(-H // ws) * ws, -(-W // ws) * ws
        return R.shift_attn_mask(Hp, Wp, ws, self.shift)


class PatchEmbed(nn.Module):
    def __init__(self, dim):
        super().__init__()
        self.proj = nn.Conv2d(3, dim, kernel_size=4, stride=4)
        self.norm = nn.LayerNorm(dim)

    def forward(self, img):
        _, _, H, W = img.shape
        img = F.pad(img, (0, (4 - W % 4) % 4, 0, (4 - H % 4) % 4))
        x = self.proj(img)
        Wh, Ww = x.shape[2], x.shape[3]
        return self.norm(x.flatten(2).transpose(1, 2)), Wh, Ww


class SwinBody(nn.Module):
    def __init__(self, embed_dim=128, depths=(2, 2, 18, 2), heads=(4, 8, 16, 32), ws=12):
        super().__init__()
        self.patch_embed = PatchEmbed(embed_dim)
        self.layers = nn.ModuleList([BasicLayer(embed_dim * 2 ** i, depths[i], heads[i], ws, i < 3, 768 if i == 3 else None)
                                     for i in range(4)])
        self.num_features = [embed_dim * 2 ** i for i in range(4)]
        self.norm0 = nn.Identity()                               # backbone_arch ...RETINANET (:689-690)
        for i in (1, 2, 3):
            self.add_module(f"norm{i}", nn.LayerNorm(self.num_features[i]))


class TextModel(nn.Module):
    def __init__(self):
        super().__init__()
        self.embeddings = R.RobertaEmbeddings(50265, 768, 514, dropout=0.0)
        enc = nn.Module()
        enc.layer = nn.ModuleList([R.RobertaLayer(768, 12, 3072, 1e-5, 0.0, i, 6, 1024) for i in range(12)])
        for i, lyr in enumerate(enc.layer):
            if i < 6:
                del lyr.alpha_t2i                                # roberta_fused_model_v2.py:402-404: only cross layers own it
            else:
                lyr.crossattention_t2i.output.LayerNorm = nn.Identity()          # :314-316
        self.encoder = enc


class FusionRef(nn.Module):
    """forward(input_ids, attention_mask, images) -> ([stage2..stage5 NCHW maps], language dict)."""

    def __init__(self):
        super().__init__()
        self.backbone = nn.Module()
        self.backbone.body = SwinBody()
        self.language_backbone = nn.Module()
        self.language_backbone.body = nn.Module()
        self.language_backbone.body.model = TextModel()

    def forward(self, input_ids, attention_mask, images):
        sw, tm = self.backbone.body, self.language_backbone.body.model
        x, Wh, Ww = sw.patch_embed(images)
        text = tm.embeddings(input_ids)
        ext = (1.0 - attention_mask[:, None, None, :].float()) * -10000.0
        for lyr in tm.encoder.layer[:6]:
            text = lyr(text, ext)[0]
        outs = []

        def emit(i, t, H, W):
            outs.append(getattr(sw, f"norm{i}")(t).view(-1, H, W, sw.num_features[i]).permute(0, 3, 1, 2).contiguous())
        for i in (0, 1):
            layer = sw.layers[i]
            mask = layer.get_attention_mask(Wh, Ww)
            for blk in layer.blocks:
                x = blk(x, Wh, Ww, mask)
            emit(i, x, Wh, Ww)
            x = layer.downsample(x, Wh, Ww)
            Wh, Ww = (Wh + 1) // 2, (Ww + 1) // 2
        layer = sw.layers[2]
        mask = layer.get_attention_mask(Wh, Ww)
        for j, blk in enumerate(layer.blocks):
            if j < 14:
                x = blk(x, Wh, Ww, mask)
            else:                                                # both sides read the other's PRE-block state (:866-885)
                fused = blk(x, Wh, Ww, mask, text, ext)
                text = tm.encoder.layer[j - 14 + 6](text, ext, encoder_hidden_states=x)[0]
                x = fused
        emit(2, x, Wh, Ww)
        x = layer.downsample(x, Wh, Ww)
        Wh, Ww = (Wh + 1) // 2, (Ww + 1) // 2
        layer = sw.layers[3]
        mask = layer.get_attention_mask(Wh, Ww)
        for j in (0, 1):
            fused = layer.blocks[j](x, Wh, Ww, mask, text, ext)
            text = tm.encoder.layer[10 + j](text, ext, encoder_hidden_states=x)[0]
            x = fused
        emit(3, x, Wh, Ww)
        m = attention_mask.float()
        embedded = text * m.unsqueeze(-1)                        # get_aggregated_output, USE_DOT_PRODUCT_TOKEN_LOSS branch
        lang = {"aggregate": embedded.sum(1) / m.sum(-1, keepdim=True), "embedded": embedded, "masks": attention_mask, "hidden": text}
        return outs, lang
