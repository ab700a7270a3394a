"""Golden vectors for the fine-grained fused backbone, produced by executing the REFERENCE's own
fusion_swin_transformer_v2.py and roberta_fused_model_v2.py (loaded by path under oracle/shim.py, CONTAINER-ONLY) through
`FusionSwinTransformer.forward`, with stand-ins only for what lies outside the backbone: `backbone.fpn` (identity: the fixtures
hold the pre-FPN stage maps), the yacs cfg read by `get_aggregated_output` (its USE_DOT_PRODUCT_TOKEN_LOSS branch is restated in
the stand-in), HF's 4.6.0 `get_extended_attention_mask`.  Weights come from oracle/detgen.py by parameter name; fixtures hold
fingerprints of the four stage maps, the text states, and -- for a scalar projection of all outputs -- the gradient norm of
every parameter plus a few sampled gradients.  Run: python -m oracle.gen_fusion_golden"""
import os
import sys
import types

import numpy as np
import torch
import torch.nn as nn

from . import cases, detgen, shim

FG = os.path.join(shim.REF, "fine_grained", "maskrcnn_benchmark", "modeling")
FG_CASES = {
    # name: image H, W (multiples of 32, as the detection pipeline pads them), batch, text length
    "fg_224x288": dict(H=224, W=288, B=2, S=16),      # stage grids 56x72 / 28x36 / 14x18 (pad 24x24) / 7x9 (pad 12x12, one window, shifted)
    "fg_320x416": dict(H=320, W=416, B=1, S=24),      # 80x104 (pad 84x108) / 40x52 (48x60) / 20x26 (24x36) / 10x13 (12x24)
}


def fg_inputs(name):
    c = FG_CASES[name]
    img = cases.randn(name + ".img", (c["B"], 3, c["H"], c["W"]))
    g = detgen._rng("fg_ids:" + name, 0)
    ids = g.integers(3, 50000, size=(c["B"], c["S"])).astype(np.int64)
    ids[:, 0] = 0
    if c["B"] > 1:
        ids[1, c["S"] * 2 // 3:] = 1                  # a padded row
    ids = torch.from_numpy(ids)
    return img, ids, (ids != 1).long()


def projections(name, outs, hidden):
    """fixed random cotangents: the scalar whose gradient the fixtures fingerprint"""
    tot = 0.0
    for i, o in enumerate(outs):
        tot = tot + (o * cases.randn(f"{name}.g{i}", tuple(o.shape))).sum() / o.numel() ** 0.5
    return tot + (hidden * cases.randn(f"{name}.gt", tuple(hidden.shape))).sum() / hidden.numel() ** 0.5


def load_reference():
    shim._install_stubs()
    pkg = "_fiber_reference_fg"
    p = types.ModuleType(pkg)
    p.__path__ = [FG]
    sys.modules[pkg] = p
    fs = shim._load("fusion_swin_transformer_v2", os.path.join(FG, "backbone", "fusion_swin_transformer_v2.py"), pkg)
    rf = shim._load("roberta_fused_model_v2", os.path.join(FG, "language_backbone", "roberta_fused_model_v2.py"), pkg)
    return fs, rf


def build_reference(fs, rf):
    sw = fs.SwinTransformer(patch_size=4, in_chans=3, embed_dim=128, depths=[2, 2, 18, 2], num_heads=[4, 8, 16, 32], window_size=12,
                            drop_path_rate=0.0, ape=False, patch_norm=True, frozen_stages=-1, backbone_arch="SWINT-FPN-RETINANET",
                            use_checkpoint=False, out_features=["stage2", "stage3", "stage4", "stage5"])
    cfg = shim.roberta_config(hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0)

    class Body(nn.Module):
        def __init__(self):
            super().__init__()
            self.model = nn.Module()
            self.model.embeddings = rf.RobertaEmbeddings(cfg)
            self.model.encoder = rf.RobertaEncoder(cfg)
            self.model.get_extended_attention_mask = lambda m, shape, device=None: (1.0 - m[:, None, None, :].float()) * -10000.0

        def get_aggregated_output(self, features, input_ids, mask):
            embedded = features * mask.unsqueeze(-1).float()
            return {"aggregate": embedded.sum(1) / mask.sum(-1).unsqueeze(-1).float(), "embedded": embedded, "masks": mask,
                    "hidden": features}
    vb, lb = nn.Module(), nn.Module()
    vb.body, vb.fpn = sw, (lambda outs: outs)
    lb.body = Body()
    return fs.FusionSwinTransformer(vb, lb).eval()


def main():
    torch.set_num_threads(8)
    fs, rf = load_reference()
    m = detgen.fill_(build_reference(fs, rf))
    names = sorted(n for n, _ in m.named_parameters())
    for name in FG_CASES:
        img, ids, am = fg_inputs(name)
        m.zero_grad(set_to_none=True)
        outs, lang, _ = m({"input_ids": ids, "attention_mask": am}, types.SimpleNamespace(tensors=img))
        d = {"param_names": np.array(names)}
        for i, o in enumerate(outs):
            cases.flatten_summary(f"stage{i + 2}", o, d)
        cases.flatten_summary("hidden", lang["hidden"], d)
        cases.flatten_summary("aggregate", lang["aggregate"], d)
        projections(name, outs, lang["hidden"]).backward()
        unused = []
        for n, p in m.named_parameters():
            if p.grad is None:
                unused.append(n)
            else:
                d[f"gradnorm/{n}"] = np.float64(p.grad.double().norm().item())
        d["unused_params"] = np.array(unused)
        P = dict(m.named_parameters())
        for n in ("backbone.body.patch_embed.proj.weight", "backbone.body.layers.2.blocks.15.attn.alpha_i2t",
                  "backbone.body.layers.3.blocks.1.attn.relative_position_bias_table",
                  "backbone.body.layers.2.blocks.3.attn.qkv.weight", "backbone.body.layers.3.blocks.0.attn.qkv_text_i2t.weight",
                  "language_backbone.body.model.encoder.layer.11.alpha_t2i",
                  "language_backbone.body.model.encoder.layer.7.crossattention_t2i.self.key.weight",
                  "language_backbone.body.model.embeddings.position_embeddings.weight"):
            cases.flatten_summary("grad/" + n, P[n].grad, d)
        np.savez_compressed(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", name + ".npz"), **d)
        print("wrote", name, [tuple(o.shape) for o in outs], "unused:", unused)


if __name__ == "__main__":
    main()
