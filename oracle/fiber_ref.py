"""CPU oracle for the FIBER coarse-grained fused-backbone path.  TEST INFRASTRUCTURE ONLY.

A plain-PyTorch (fp32, no custom kernels) restatement of the arithmetic of the
reference hot path, written from the math in SURVEY.md Appendix A.  Only
``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may
import this file; the product package ``fiber_amd`` never does.

Parity status: PINNED by ``tests/golden/*.npz`` -- vectors produced in the build
container by running the reference's own ``swin_transformer.py`` / ``roberta.py``
(loaded by path under ``oracle/shim.py``) on weights from ``oracle/detgen.py``
(``oracle/gen_golden.py`` is the generating script).  The reference ships no tests
or golden vectors for this path, so those fixtures are the pin.

Reference lines each piece follows (paths relative to
/root/reference/coarse_grained/fiber/modules/):
  window_partition / window_reverse / roll ...... swin_transformer.py:99-126, 367, 384
  WindowAttention (self + i2t cross) ............ swin_transformer.py:195-261
  SwinTransformerBlock (+ shift mask) ........... swin_transformer.py:327-393
  PatchMerging .................................. swin_transformer.py:411-432
  BasicLayer fusion gating ...................... swin_transformer.py:502
  SwinTransformer ............................... swin_transformer.py:547-650
  PatchEmbed / Mlp / DropPath ................... timm==0.4.12 semantics (third party,
                                                  restated; call sites swin_transformer.py:588,325,322)
  RobertaEmbeddings ............................. roberta.py:169-199, 877-888
  RobertaSelfAttention .......................... roberta.py:256-326
  RobertaSelfOutput (no residual / LN) .......... roberta.py:337-340
  RobertaLayer .................................. roberta.py:441-502
  extended attention mask ....................... transformers==4.6.0 (third party): (1-m)*-10000
  infer() fused sequencing ...................... fiber_module.py:310-367
  Pooler / ITMHead / MLMHead .................... heads.py:8-43
  compute_mlm / compute_itm ..................... objectives.py:17-61
"""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F


# --------------------------------------------------------------------------- Swin side


def rel_pos_index(ws):
    """(ws*ws, ws*ws) int64 index into the (2ws-1)^2 bias table (swin_transformer.py:166-175)."""
    r = torch.arange(ws)
    rr, cc = torch.meshgrid(r, r, indexing="ij")
    rr, cc = rr.reshape(-1), cc.reshape(-1)
    dr = rr[:, None] - rr[None, :] + ws - 1
    dc = cc[:, None] - cc[None, :] + ws - 1
    return dr * (2 * ws - 1) + dc


def shift_attn_mask(H, W, ws, shift):
    """(nW, N, N) additive mask in {0,-100} for shifted windows (swin_transformer.py:327-350)."""
    def region(n):
        lab = torch.zeros(n, dtype=torch.long)
        lab[n - ws:n - shift] = 1
        lab[n - shift:] = 2
        return lab
    lab = region(H)[:, None] * 3 + region(W)[None, :]            # (H, W) region labels of the rolled grid
    lab = lab.view(H // ws, ws, W // ws, ws).permute(0, 2, 1, 3).reshape(-1, ws * ws)
    diff = lab[:, None, :] - lab[:, :, None]
    return torch.where(diff != 0, torch.tensor(-100.0), torch.tensor(0.0))


def to_windows(x, ws):
    B, H, W, C = x.shape
    x = x.view(B, H // ws, ws, W // ws, ws, C).permute(0, 1, 3, 2, 4, 5)
    return x.reshape(-1, ws * ws, C)


def from_windows(xw, ws, H, W):
    C = xw.shape[-1]
    B = xw.shape[0] // ((H // ws) * (W // ws))
    x = xw.view(B, H // ws, W // ws, ws, ws, C).permute(0, 1, 3, 2, 4, 5)
    return x.reshape(B, H, W, C)


class DropPath(nn.Module):
    """timm 0.4.12 DropPath: per-sample keep mask floor(keep + U[0,1)), scaled by 1/keep."""

    def __init__(self, p):
        super().__init__()
        self.p = float(p)

    def forward(self, x):
        if self.p == 0.0 or not self.training:
            return x
        keep = 1.0 - self.p
        m = torch.floor(keep + torch.rand((x.shape[0],) + (1,) * (x.dim() - 1), dtype=x.dtype, device=x.device))
        return x / keep * m


class Mlp(nn.Module):
    def __init__(self, dim, hidden):
        super().__init__()
        self.fc1 = nn.Linear(dim, hidden)
        self.fc2 = nn.Linear(hidden, dim)

    def forward(self, x):
        return self.fc2(F.gelu(self.fc1(x)))


class PatchEmbed(nn.Module):
    def __init__(self, img_size, patch, in_chans, dim):
        super().__init__()
        self.grid_size = (img_size // patch, img_size // patch)
        self.num_patches = self.grid_size[0] * self.grid_size[1]
        self.proj = nn.Conv2d(in_chans, dim, kernel_size=patch, stride=patch)
        self.norm = nn.LayerNorm(dim)

    def forward(self, img):
        return self.norm(self.proj(img).flatten(2).transpose(1, 2))


class WindowAttention(nn.Module):
    def __init__(self, dim, ws, heads, dim_text=None):
        super().__init__()
        self.dim, self.ws, self.heads = dim, ws, heads
        self.relative_position_bias_table = nn.Parameter(torch.zeros((2 * ws - 1) ** 2, heads))
        self.register_buffer("relative_position_index", rel_pos_index(ws))
        self.qkv = nn.Linear(dim, 3 * dim)
        self.proj = nn.Linear(dim, dim)
        nn.init.trunc_normal_(self.relative_position_bias_table, std=0.02)
        if dim_text is not None:
            self.qkv_text_i2t = nn.Linear(dim_text, 2 * dim)
            self.qkv_i2t = nn.Linear(dim, dim)
            self.proj_i2t = nn.Linear(dim, dim)
            self.alpha_i2t = nn.Parameter(torch.zeros(1))
            self.norm_i2t_i = nn.LayerNorm(dim)

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
        o = (a.softmax(-1) @ v).transpose(1, 2).reshape(Bw, N, C)
        out = self.proj(o)
        if y is not None:
            B, S, _ = y.shape
            nW = Bw // B
            assert nW * B == Bw
            kt, vt = self.qkv_text_i2t(y).view(B, S, 2, h, d).permute(2, 0, 3, 1, 4)
            kt = kt.repeat_interleave(nW, 0)
            vt = vt.repeat_interleave(nW, 0)
            qi = self.qkv_i2t(self.norm_i2t_i(out)).view(Bw, N, h, d).transpose(1, 2) * d ** -0.5
            ai = qi @ kt.transpose(-1, -2)
            if y_mask is not None:
                ai = ai + y_mask.view(B, 1, 1, S).repeat_interleave(nW, 0)
            yi = (ai.softmax(-1) @ vt).transpose(1, 2).reshape(Bw, N, C)
            out = out + self.alpha_i2t * self.proj_i2t(yi)
        return out


class SwinTransformerBlock(nn.Module):
    def __init__(self, dim, res, heads, ws, shift, mlp_ratio=4.0, drop_path=0.0, dim_text=None):
        super().__init__()
        if min(res) <= ws:
            shift, ws = 0, min(res)
        self.res, self.ws, self.shift = res, ws, shift
        self.norm1 = nn.LayerNorm(dim)
        self.attn = WindowAttention(dim, ws, heads, dim_text)
        self.drop_path = DropPath(drop_path)
        self.norm2 = nn.LayerNorm(dim)
        self.mlp = Mlp(dim, int(dim * mlp_ratio))
        self.register_buffer("attn_mask", shift_attn_mask(res[0], res[1], ws, shift) if shift > 0 else None)

    def forward(self, x, y=None, y_mask=None):
        H, W = self.res
        B, L, C = x.shape
        assert L == H * W
        u = self.norm1(x).view(B, H, W, C)
        if self.shift:
            u = torch.roll(u, (-self.shift, -self.shift), (1, 2))
        aw = self.attn(to_windows(u, self.ws), self.attn_mask, y, y_mask)
        u = from_windows(aw, self.ws, H, W)
        if self.shift:
            u = torch.roll(u, (self.shift, self.shift), (1, 2))
        x = x + self.drop_path(u.reshape(B, L, C))
        return x + self.drop_path(self.mlp(self.norm2(x)))


class PatchMerging(nn.Module):
    def __init__(self, res, dim):
        super().__init__()
        self.res = res
        self.reduction = nn.Linear(4 * dim, 2 * dim, bias=False)
        self.norm = nn.LayerNorm(4 * dim)

    def forward(self, x):
        H, W = self.res
        B, L, C = x.shape
        x = x.view(B, H, W, C)
        z = torch.cat([x[:, 0::2, 0::2], x[:, 1::2, 0::2], x[:, 0::2, 1::2], x[:, 1::2, 1::2]], -1)
        return self.reduction(self.norm(z.view(B, -1, 4 * C)))


class BasicLayer(nn.Module):
    def __init__(self, dim, res, depth, heads, ws, drop_path, downsample, dim_text, layer_index, num_fuse_block):
        super().__init__()
        self.blocks = nn.ModuleList([
            SwinTransformerBlock(
                dim, res, heads, ws, 0 if i % 2 == 0 else ws // 2, drop_path=drop_path[i],
                dim_text=None if (layer_index == 2 and i < 20 - num_fuse_block) else dim_text)
            for i in range(depth)])
        self.downsample = PatchMerging(res, dim) if downsample else None

    def forward(self, x, y=None, y_mask=None):
        for blk in self.blocks:
            x = blk(x, y, y_mask)
        return self.downsample(x) if self.downsample is not None else x


SWIN_VARIANTS = {
    # name: (embed_dim, depths, heads)
    "swin_base_patch4_window12_384_in22k": (128, (2, 2, 18, 2), (4, 8, 16, 32)),
    "swin_base_patch4_window12_384": (128, (2, 2, 18, 2), (4, 8, 16, 32)),
    "swin_base_patch4_window7_224_in22k": (128, (2, 2, 18, 2), (4, 8, 16, 32)),
    "swin_base_patch4_window7_224": (128, (2, 2, 18, 2), (4, 8, 16, 32)),
    "swin_tiny_patch4_window7_224": (96, (2, 2, 6, 2), (3, 6, 12, 24)),
    "swin_small_patch4_window7_224": (96, (2, 2, 18, 2), (3, 6, 12, 24)),
    "swin_large_patch4_window12_384_in22k": (192, (2, 2, 18, 2), (6, 12, 24, 48)),
}


class SwinTransformer(nn.Module):
    def __init__(self, img_size, embed_dim, depths, num_heads, dim_text=768, num_fuse_block=6,
                 drop_path_rate=0.1, patch_size=4, in_chans=3):
        super().__init__()
        ws = int(img_size / 32)                      # swin_transformer.py:575 overrides the factory's window
        self.patch_embed = PatchEmbed(img_size, patch_size, in_chans, embed_dim)
        self.absolute_pos_embed = None
        self.pos_drop = nn.Identity()                # drop_rate = 0
        g = self.patch_embed.grid_size
        dpr = torch.linspace(0, drop_path_rate, sum(depths)).tolist()
        self.layers = nn.Sequential(*[
            BasicLayer(embed_dim * 2 ** i, (g[0] // 2 ** i, g[1] // 2 ** i), depths[i], num_heads[i], ws,
                       dpr[sum(depths[:i]):sum(depths[:i + 1])], i < len(depths) - 1,
                       dim_text if i >= 2 else None, i, num_fuse_block)
            for i in range(len(depths))])
        self.num_features = embed_dim * 2 ** (len(depths) - 1)
        self.norm = nn.LayerNorm(self.num_features)
        self.apply(self._init)

    @staticmethod
    def _init(m):
        if isinstance(m, nn.Linear):
            nn.init.trunc_normal_(m.weight, std=0.02)
            if m.bias is not None:
                nn.init.zeros_(m.bias)
        elif isinstance(m, nn.LayerNorm):
            nn.init.ones_(m.weight)
            nn.init.zeros_(m.bias)


# --------------------------------------------------------------------------- RoBERTa side


class RobertaEmbeddings(nn.Module):
    def __init__(self, vocab, hidden, max_pos, type_vocab=1, pad=1, eps=1e-5, dropout=0.1):
        super().__init__()
        self.padding_idx = pad
        self.word_embeddings = nn.Embedding(vocab, hidden, padding_idx=pad)
        self.position_embeddings = nn.Embedding(max_pos, hidden, padding_idx=pad)
        self.token_type_embeddings = nn.Embedding(type_vocab, hidden)
        self.LayerNorm = nn.LayerNorm(hidden, eps=eps)
        self.dropout = nn.Dropout(dropout)
        self.register_buffer("position_ids", torch.arange(max_pos).expand((1, -1)))

    def forward(self, input_ids):
        m = (input_ids != self.padding_idx).long()
        pos = torch.cumsum(m, 1) * m + self.padding_idx
        e = self.word_embeddings(input_ids) + self.token_type_embeddings.weight[0] + self.position_embeddings(pos)
        return self.dropout(self.LayerNorm(e))


class RobertaSelfAttention(nn.Module):
    def __init__(self, hidden, heads, kv_dim, dropout):
        super().__init__()
        self.heads = heads
        self.query = nn.Linear(hidden, hidden)
        self.key = nn.Linear(kv_dim, hidden)
        self.value = nn.Linear(kv_dim, hidden)
        self.dropout = nn.Dropout(dropout)

    def forward(self, hq, hkv, mask):
        B, S, Hd = hq.shape
        d = Hd // self.heads
        q = self.query(hq).view(B, S, self.heads, d).transpose(1, 2)
        k = self.key(hkv).view(B, -1, self.heads, d).transpose(1, 2)
        v = self.value(hkv).view(B, -1, self.heads, d).transpose(1, 2)
        a = q @ k.transpose(-1, -2) / math.sqrt(d)
        if mask is not None:
            a = a + mask
        p = self.dropout(a.softmax(-1))
        return (p @ v).transpose(1, 2).reshape(B, S, Hd)


class RobertaSelfOutput(nn.Module):
    def __init__(self, hidden, eps, dropout):
        super().__init__()
        self.dense = nn.Linear(hidden, hidden)
        self.LayerNorm = nn.LayerNorm(hidden, eps=eps)
        self.dropout = nn.Dropout(dropout)

    def forward(self, x):                                    # no residual, no LN here (roberta.py:337-340)
        return self.dropout(self.dense(x))


class RobertaAttention(nn.Module):
    def __init__(self, hidden, heads, kv_dim, eps, dropout):
        super().__init__()
        self.self = RobertaSelfAttention(hidden, heads, kv_dim, dropout)
        self.output = RobertaSelfOutput(hidden, eps, dropout)

    def forward(self, hq, hkv, mask):
        return self.output(self.self(hq, hkv, mask))


class RobertaIntermediate(nn.Module):
    def __init__(self, hidden, inter):
        super().__init__()
        self.dense = nn.Linear(hidden, inter)

    def forward(self, x):
        return F.gelu(self.dense(x))


class RobertaOutput(nn.Module):
    def __init__(self, hidden, inter, eps, dropout):
        super().__init__()
        self.dense = nn.Linear(inter, hidden)
        self.LayerNorm = nn.LayerNorm(hidden, eps=eps)
        self.dropout = nn.Dropout(dropout)

    def forward(self, x, resid, last_norm=True):
        x = self.dropout(self.dense(x)) + resid
        return self.LayerNorm(x) if last_norm else x


class RobertaLayer(nn.Module):
    def __init__(self, hidden, heads, inter, eps, dropout, layer_index, num_fuse_block, dim_img, num_layers=12):
        super().__init__()
        self.attention = RobertaAttention(hidden, heads, hidden, eps, dropout)
        if layer_index >= num_layers - num_fuse_block:
            kv = dim_img // 2 if layer_index < 10 else dim_img            # roberta.py:236-241
            self.crossattention_t2i = RobertaAttention(hidden, heads, kv, eps, dropout)
        self.intermediate = RobertaIntermediate(hidden, inter)
        self.output = RobertaOutput(hidden, inter, eps, dropout)
        self.alpha_t2i = nn.Parameter(torch.zeros(1))

    def forward(self, h, mask=None, encoder_hidden_states=None, last_norm=True):
        a = self.attention(h, h, mask)
        if encoder_hidden_states is not None:
            a = self.alpha_t2i * self.crossattention_t2i(a, encoder_hidden_states, None) + a
        a = self.attention.output.LayerNorm(a + h)
        return (self.output(self.intermediate(a), a, last_norm),)


class RobertaEncoder(nn.Module):
    def __init__(self, **kw):
        super().__init__()
        n = kw.get("num_layers", 12)
        self.layer = nn.ModuleList([RobertaLayer(layer_index=i, **kw) for i in range(n)])


class RobertaPooler(nn.Module):
    def __init__(self, hidden):
        super().__init__()
        self.dense = nn.Linear(hidden, hidden)


class RobertaModel(nn.Module):
    def __init__(self, vocab=50265, hidden=768, heads=12, inter=3072, max_pos=514, eps=1e-5, dropout=0.1,
                 num_fuse_block=6, dim_img=1024, num_layers=12):
        super().__init__()
        self.embeddings = RobertaEmbeddings(vocab, hidden, max_pos, eps=eps, dropout=dropout)
        self.encoder = RobertaEncoder(hidden=hidden, heads=heads, inter=inter, eps=eps, dropout=dropout,
                                      num_fuse_block=num_fuse_block, dim_img=dim_img, num_layers=num_layers)
        self.pooler = RobertaPooler(hidden)                                  # present in the state dict, unused
        self.apply(self._init)

    @staticmethod
    def _init(m):
        if isinstance(m, (nn.Linear, nn.Embedding)):
            nn.init.normal_(m.weight, std=0.02)
            if isinstance(m, nn.Linear) and m.bias is not None:
                nn.init.zeros_(m.bias)
        elif isinstance(m, nn.LayerNorm):
            nn.init.ones_(m.weight)
            nn.init.zeros_(m.bias)

    @staticmethod
    def get_extended_attention_mask(mask, input_shape=None, device=None):
        return (1.0 - mask[:, None, None, :].float()) * -10000.0           # transformers 4.6.0 form


# --------------------------------------------------------------------------- heads + the fused driver


class Pooler(nn.Module):
    def __init__(self, hidden):
        super().__init__()
        self.dense = nn.Linear(hidden, hidden)

    def forward(self, x):
        return torch.tanh(self.dense(x[:, 0]))


class ITMHead(nn.Module):
    def __init__(self, hidden):
        super().__init__()
        self.fc = nn.Linear(hidden, 2)

    def forward(self, x):
        return self.fc(x)


class _PredictionHeadTransform(nn.Module):
    def __init__(self, hidden, eps):
        super().__init__()
        self.dense = nn.Linear(hidden, hidden)
        self.LayerNorm = nn.LayerNorm(hidden, eps=eps)

    def forward(self, x):
        return self.LayerNorm(F.gelu(self.dense(x)))


class MLMHead(nn.Module):
    def __init__(self, hidden, vocab, eps=1e-12):
        super().__init__()
        self.transform = _PredictionHeadTransform(hidden, eps)
        self.decoder = nn.Linear(hidden, vocab, bias=False)
        self.bias = nn.Parameter(torch.zeros(vocab))

    def forward(self, x):
        return self.decoder(self.transform(x)) + self.bias


def _init_head(m):
    if isinstance(m, (nn.Linear, nn.Embedding)):
        m.weight.data.normal_(mean=0.0, std=0.02)
    elif isinstance(m, nn.LayerNorm):
        m.bias.data.zero_()
        m.weight.data.fill_(1.0)
    if isinstance(m, nn.Linear) and m.bias is not None:
        m.bias.data.zero_()


DEFAULT_CONFIG = dict(
    vit="swin_base_patch4_window12_384_in22k", image_size=384, resolution_before=384, pretrained_vit=False,
    tokenizer="roberta-base", vocab_size=50265, max_text_len=40, hidden_size=768, num_heads=12, num_layers=12,
    mlp_ratio=4, drop_rate=0.1, num_fuse_block=6, input_image_embed_size=1024, input_text_embed_size=768,
    loss_names={"itm": 1, "mlm": 1, "itc": 0, "vqa": 0, "nlvr2": 0}, itc_pooler=True, load_path="",
    test_only=False, vqav2_label_size=3129, draw_false_image=1,
    # oracle-only knobs (the reference takes these from the roberta-base checkpoint config / factory defaults)
    text_dropout=0.1, drop_path_rate=0.1, max_position_embeddings=514,
)


class FiberRef(nn.Module):
    """Parameter tree + fused forward with the reference's state-dict key names."""

    def __init__(self, config):
        super().__init__()
        c = dict(DEFAULT_CONFIG)
        c.update(config)
        self.config = c
        hs = c["hidden_size"]
        dim, depths, heads = c.get("swin_arch") or SWIN_VARIANTS[c["vit"]]
        self.num_fuse_block, self.num_text_layer = c["num_fuse_block"], c["num_layers"]
        self.cross_modal_text_transform = nn.Linear(c["input_text_embed_size"], hs)
        self.cross_modal_image_transform = nn.Linear(c["input_image_embed_size"], hs)
        self.cross_modal_text_transform_itc = nn.Linear(c["input_text_embed_size"], hs)
        self.cross_modal_image_transform_itc = nn.Linear(c["input_image_embed_size"], hs)
        # NB reference bug kept: DIM_TXT typo => Swin's text dim is always the module default 768
        # (fiber_module.py:48 vs swin_transformer.py:13,628); overridable for tiny test configs.
        self.vit_model = SwinTransformer(c["image_size"], dim, depths, heads, dim_text=c.get("swin_dim_text", 768),
                                         num_fuse_block=self.num_fuse_block, drop_path_rate=c["drop_path_rate"])
        self.avgpool = nn.AdaptiveAvgPool1d(1)
        self.text_transformer = RobertaModel(
            vocab=c["vocab_size"], hidden=c["input_text_embed_size"], heads=c["num_heads"],
            inter=c["input_text_embed_size"] * c["mlp_ratio"], max_pos=c["max_position_embeddings"],
            dropout=c["text_dropout"], num_fuse_block=self.num_fuse_block, dim_img=c["input_image_embed_size"],
            num_layers=c["num_layers"])
        self.cross_modal_image_pooler = Pooler(hs)
        self.cross_modal_text_pooler = Pooler(hs)
        if c["itc_pooler"]:
            self.cross_modal_image_pooler_itc = Pooler(hs)
            self.cross_modal_text_pooler_itc = Pooler(hs)
        if c["loss_names"].get("mlm", 0) > 0:
            self.mlm_score = MLMHead(hs, c["vocab_size"])
        if c["loss_names"].get("itm", 0) > 0:
            self.itm_score = ITMHead(hs * 2)
            self.rank_output = nn.Linear(hs, 1)
        if c["loss_names"].get("itc", 0) > 0:             # fiber_module.py:56-67
            qs = c.get("itc_queue_size", 4096)
            self.queue_size = qs
            self.temp = nn.Parameter(torch.ones([]) * 0.07)
            self.register_buffer("image_queue", torch.randn(hs, qs))
            self.register_buffer("text_queue", torch.randn(hs, qs))
            self.register_buffer("image_input_queue", torch.randn(qs, 3, c["image_size"], c["image_size"]))
            self.register_buffer("text_input_queue", torch.zeros(qs, c["max_text_len"], dtype=torch.long))
            self.register_buffer("text_input_mask_queue", torch.zeros(qs, c["max_text_len"], dtype=torch.long))
            self.register_buffer("queue_ptr", torch.zeros(1, dtype=torch.long))
            self.register_buffer("queue_total", torch.zeros(1, dtype=torch.long))
        if c["loss_names"].get("vqa", 0) > 0:             # fiber_module.py:149-157
            self.vqa_classifier = nn.Sequential(nn.Linear(hs * 2, hs * 2), nn.LayerNorm(hs * 2), nn.GELU(),
                                                nn.Linear(hs * 2, c["vqav2_label_size"]))
        for name, m in self.named_children():
            if name not in ("vit_model", "text_transformer"):
                m.apply(_init_head)

    def infer(self, batch, mask_text=False, img=None):
        """fiber_module.py:310-367 (fused branch only)."""
        if img is None:
            img = batch["image"][0]
        sfx = "_mlm" if mask_text else ""
        ids, labels, masks = batch["text_ids" + sfx], batch["text_labels" + sfx], batch["text_masks"]
        vit, txt = self.vit_model, self.text_transformer
        x = vit.patch_embed(img)
        x = vit.layers[0](x)
        x = vit.layers[1](x)
        t = txt.embeddings(ids)
        ext = txt.get_extended_attention_mask(masks)
        npt = self.num_text_layer - self.num_fuse_block
        for lyr in txt.encoder.layer[:npt]:
            t = lyr(t, ext)[0]
        npb = 8 + npt
        for i, blk in enumerate(vit.layers[2].blocks):
            if i < npb:
                x = blk(x)
            else:
                fx = blk(x, t, ext)
                t = txt.encoder.layer[i - 8](t, ext, encoder_hidden_states=x)[0]
                x = fx
        x = vit.layers[2].downsample(x)
        for i, blk in enumerate(vit.layers[3].blocks):
            fx = blk(x, t, ext)
            t = txt.encoder.layer[i + 10](t, ext, encoder_hidden_states=x, last_norm=(i == 0))[0]
            x = fx
        t = self.cross_modal_text_transform(t)
        x = self.cross_modal_image_transform(x)
        cls_t = self.cross_modal_text_pooler(t)
        cls_i = self.cross_modal_image_pooler(x.mean(1, keepdim=True))
        return {"text_feats": t, "image_feats": x, "cls_feats": torch.cat([cls_t, cls_i], -1),
                "text_labels": labels, "text_ids": ids, "text_masks": masks, "image": img}

    def infer_text_only(self, batch):
        """fiber_module.py:247-277: all 12 layers without cross-attention, ITC transform + pooler, L2-normalised."""
        txt = self.text_transformer
        t = txt.embeddings(batch["text_ids"])
        ext = txt.get_extended_attention_mask(batch["text_masks"])
        for lyr in txt.encoder.layer:
            t = lyr(t, ext)[0]
        t = self.cross_modal_text_transform_itc(t)
        cls = self.cross_modal_text_pooler_itc(t) if self.config["itc_pooler"] else t[:, 0]
        return {"text_feats": t, "cls_feats": cls / cls.norm(dim=-1, keepdim=True)}

    def infer_image_only(self, batch):
        """fiber_module.py:279-308: every Swin block without the text input, final norm, ITC transform + pooler."""
        vit = self.vit_model
        x = vit.patch_embed(batch["image"][0])
        for layer in vit.layers:
            x = layer(x)
        x = self.cross_modal_image_transform_itc(vit.norm(x))
        avg = x.mean(1, keepdim=True)
        cls = self.cross_modal_image_pooler_itc(avg) if self.config["itc_pooler"] else avg[:, 0]
        return {"image_feats": x, "cls_feats": cls / cls.norm(dim=-1, keepdim=True)}

    def compute_itc(self, batch, neg_idx):
        """objectives.py:119-180 with the hard-negative indices `neg_idx = (image_idx, text_idx)` supplied by the caller
        (the reference draws them with torch.multinomial; fixtures record the draw).  Returns (dict, negatives)."""
        with torch.no_grad():
            self.temp.clamp_(0.001, 1.0)
        fi, ft = self.infer_image_only(batch)["cls_feats"], self.infer_text_only(batch)["cls_feats"]
        fi_all = torch.cat([fi.t().detach(), self.image_queue.detach()], dim=1)
        ft_all = torch.cat([ft.t().detach(), self.text_queue.detach()], dim=1)
        sim_i2t, sim_t2i = fi @ ft_all / self.temp, ft @ fi_all / self.temp
        tgt = torch.zeros_like(sim_i2t)
        tgt.fill_diagonal_(1)
        loss = (-(F.log_softmax(sim_i2t, 1) * tgt).sum(1).mean() - (F.log_softmax(sim_t2i, 1) * tgt).sum(1).mean()) / 2.0
        total = int(self.queue_total)
        tot_image = torch.cat([batch["image"][0], self.image_input_queue[:total]], 0)
        tot_text = torch.cat([batch["text_ids"], self.text_input_queue[:total]], 0)
        tot_mask = torch.cat([batch["text_masks"], self.text_input_mask_queue[:total]], 0)
        ii, ti = neg_idx
        out = {"itc_loss": loss, "sim_i2t": sim_i2t, "sim_t2i": sim_t2i, "image_cls": fi, "text_cls": ft}
        return out, (tot_image[ii], tot_text[ti], tot_mask[ti])

    def dequeue_and_enqueue(self, image_feat, text_feat, image_input, text_input, text_mask):
        """fiber_module.py:181-222, single process (all_gather = identity)."""
        n = image_feat.shape[0]
        with torch.no_grad():
            slot = (int(self.queue_ptr) + torch.arange(n)) % self.queue_size
            self.image_queue[:, slot] = image_feat.detach().T
            self.text_queue[:, slot] = text_feat.detach().T
            self.image_input_queue[slot] = image_input
            self.text_input_queue[slot] = text_input
            self.text_input_mask_queue[slot] = text_mask
            self.queue_ptr[0] = (int(self.queue_ptr) + n) % self.queue_size
            self.queue_total[0] = int(self.queue_total) + n

    def compute_itm_hardneg(self, batch, image_neg, text_neg, text_mask_neg):
        """objectives.py:78-116."""
        B = batch["text_ids"].shape[0]
        img = batch["image"][0]
        b3 = {"image": [torch.cat([img, img, image_neg], 0)],
              "text_ids": torch.cat([batch["text_ids"], text_neg, batch["text_ids"]], 0),
              "text_masks": torch.cat([batch["text_masks"], text_mask_neg, batch["text_masks"]], 0),
              "text_labels": torch.cat([batch["text_labels"]] * 3, 0)}
        labels = torch.cat([torch.ones(B), torch.zeros(2 * B)])
        out = self.infer(b3)
        logits = self.itm_score(out["cls_feats"])
        return {"itm_loss": F.cross_entropy(logits, labels.long()), "itm_logits": logits, "itm_labels": labels}

    def compute_mlm(self, batch):
        out = self.infer(batch, mask_text=True)
        logits = self.mlm_score(out["text_feats"])
        loss = F.cross_entropy(logits.view(-1, self.config["vocab_size"]), out["text_labels"].view(-1), ignore_index=-100)
        return {"mlm_loss": loss, "mlm_logits": logits, "mlm_labels": out["text_labels"]}

    def compute_itm(self, batch, itm_labels):
        """objectives.py:44-61 with the permuted labels supplied by the caller (fixture-driven)."""
        sel = itm_labels.view(-1, 1, 1, 1) == 1
        imgs = torch.where(sel, batch["image"][0], batch["false_image_0"][0])
        out = self.infer(batch, img=imgs)
        logits = self.itm_score(out["cls_feats"])
        return {"itm_loss": F.cross_entropy(logits, itm_labels.long()), "itm_logits": logits, "itm_labels": itm_labels}

    def compute_vqa(self, batch):
        """objectives.py:182-213: dense soft targets from the ragged (labels, scores) lists; BCE * answer-vocabulary size."""
        out = self.infer(batch)
        logits = self.vqa_classifier(out["cls_feats"])
        targets = torch.zeros_like(logits)
        for i, (labs, scs) in enumerate(zip(batch["vqa_labels"], batch["vqa_scores"])):
            for lab, sc in zip(labs, scs):
                targets[i, lab] = sc
        loss = F.binary_cross_entropy_with_logits(logits, targets) * targets.shape[1]
        return {"vqa_loss": loss, "vqa_logits": logits, "vqa_targets": targets, "cls_feats": out["cls_feats"],
                "text_feats": out["text_feats"], "image_feats": out["image_feats"]}

    def training_loss(self, batch, itm_labels):
        return self.compute_mlm(batch)["mlm_loss"] + self.compute_itm(batch, itm_labels)["itm_loss"]
