"""Host-side pieces of the VQAv2 fine-tune configuration (BASELINE.json configs[3]) that need no GPU: the relative-position
table resize applied when a 384^2 checkpoint is loaded at 576^2, the named task config, the VQA score metric."""
import os

import numpy as np
import torch

from oracle import cases


def test_swin_adapt_position_encoding_matches_reference(golden):
    from fiber_amd.modules.swin_helpers import swin_adapt_position_encoding
    gold = golden("adapt_pos")
    ac = cases.ADAPT_CASE
    side = 2 * (ac["before"] // 32) - 1
    sd = {
        "vit_model.layers.0.blocks.0.attn.relative_position_bias_table": cases.randn("adapt.t0", (side * side, ac["heads"])),
        "vit_model.layers.2.blocks.3.attn.relative_position_bias_table": cases.randn("adapt.t1", (side * side, 2 * ac["heads"])),
        "vit_model.layers.0.blocks.0.attn.relative_position_index": torch.zeros(4, 4, dtype=torch.long),
        "vit_model.layers.0.blocks.1.attn_mask": torch.zeros(2, 4, 4),
        "vit_model.layers.0.blocks.0.norm1.weight": torch.ones(8),
    }
    same = swin_adapt_position_encoding(dict(sd), before=384, after=384)
    assert set(same) == set(sd)                                              # no-op when the resolution is unchanged
    out = swin_adapt_position_encoding(dict(sd), before=ac["before"], after=ac["after"])
    assert sorted(out.keys()) == gold["keys"].tolist()                       # stale buffers dropped, the rest kept
    for k in out:
        if k.endswith("relative_position_bias_table"):
            ref = gold["table/" + k]
            assert out[k].shape == ref.shape == ((2 * (ac["after"] // 32) - 1) ** 2, sd[k].shape[1])
            np.testing.assert_allclose(out[k].numpy(), ref, rtol=1e-6, atol=1e-6)


def test_named_vqa_config():
    from fiber_amd.config import named_config
    c = named_config("task_finetune_vqa", per_gpu_batchsize=4)
    assert c["loss_names"]["vqa"] == 1 and c["loss_names"]["mlm"] == 0 and c["loss_names"]["itm"] == 0
    assert (c["image_size"], c["max_text_len"], c["batch_size"]) == (576, 50, 512)
    assert (c["learning_rate"], c["lr_mult_head"], c["lr_mult_cross_modal"]) == (2e-5, 50, 5)
    assert c["resolution_before"] == 384 and c["vqav2_label_size"] == 3129 and c["per_gpu_batchsize"] == 4


def test_vqa_score_metric():
    from fiber_amd.modules.fiber_utils import VQAScore
    logits = torch.tensor([[0.1, 2.0, -1.0], [3.0, 0.0, 0.5], [0.0, 0.0, 1.0]])
    target = torch.tensor([[0.0, 0.6, 1.0], [0.0, 1.0, 0.0], [0.3, 0.0, 0.9]])
    m = VQAScore()
    got = m(logits, target)
    assert abs(float(got) - (0.6 + 0.0 + 0.9) / 3) < 1e-6                 # my_metrics.py:55-69
    assert abs(float(m.compute()) - 0.5) < 1e-6


def test_checkpoint_loads_across_resolutions(tmp_path):
    """fiber_module.py:138-147: a pre-training checkpoint (window 3 at 96^2) initialises a fine-tune model at 192^2
    (window 6): every bias table is resized, stale buffers do not clobber the new ones, the VQA head stays fresh."""
    from fiber_amd.config import make_config
    from fiber_amd.modules import FIBERTransformerSS
    from fiber_amd.modules.swin_helpers import swin_adapt_position_encoding
    torch.manual_seed(0)
    pre = FIBERTransformerSS(make_config(**cases.TINY))
    path = os.path.join(tmp_path, "pre.ckpt")
    torch.save({"state_dict": pre.state_dict()}, path)
    cfg = dict(cases.TINY, image_size=192, resolution_before=96, load_path=path, loss_names={"vqa": 1}, vqav2_label_size=11)
    torch.manual_seed(1)
    ft = FIBERTransformerSS(make_config(**cfg))
    key = "vit_model.layers.2.blocks.15.attn.relative_position_bias_table"
    want = swin_adapt_position_encoding({key: pre.state_dict()[key].clone()}, before=96, after=192)[key]
    assert ft.state_dict()[key].shape == (121, 4)
    assert torch.equal(ft.state_dict()[key], want)
    assert ft.state_dict()["vit_model.layers.0.blocks.1.attn_mask"].shape[-1] == 36           # rebuilt for the 6x6 window
    k2 = "text_transformer.encoder.layer.7.crossattention_t2i.self.key.weight"
    assert torch.equal(ft.state_dict()[k2], pre.state_dict()[k2])
    assert ft.vqa_classifier[3].weight.shape == (11, 128)
