"""CPU: pin oracle/fiber_ref.py (the restatement) to the golden vectors produced by the REFERENCE
modules (oracle/gen_golden.py).  fp32, tolerance 1e-5 relative."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import cases, detgen, fiber_ref as R

RT, AT = 2e-5, 2e-6


def _check_grads(module, gold, prefix="grad/"):
    n = 0
    for name, p in module.named_parameters():
        key = f"{prefix}{name}/sub"
        if key in gold:
            assert p.grad is not None, name
            cases.check_summary(f"{prefix}{name}", p.grad, gold, RT, AT)
            n += 1
    return n


@pytest.mark.parametrize("name", list(cases.BLOCK_CASES))
def test_swin_block(name, golden):
    c, gold = cases.BLOCK_CASES[name], golden(name)
    blk = R.SwinTransformerBlock(c["dim"], c["res"], c["heads"], c["ws"], c["shift"], dim_text=c["dim_text"]).eval()
    detgen.fill_(blk)
    x, y, ext, g = cases.block_inputs(name)
    x.requires_grad_(True)
    if y is not None:
        y.requires_grad_(True)
    out = blk(x, y, ext)
    (out * g).sum().backward()
    cases.check_summary("out", out, gold, RT, AT)
    cases.check_summary("grad_in/x", x.grad, gold, RT, AT)
    if y is not None:
        cases.check_summary("grad_in/y", y.grad, gold, RT, AT)
    assert _check_grads(blk, gold) >= 13


@pytest.mark.parametrize("name", list(cases.MERGE_CASES))
def test_patch_merging(name, golden):
    c, gold = cases.MERGE_CASES[name], golden(name)
    m = detgen.fill_(R.PatchMerging(c["res"], c["dim"]).eval())
    L = c["res"][0] * c["res"][1]
    x = cases.randn(name + ".x", (c["B"], L, c["dim"])).requires_grad_(True)
    g = cases.randn(name + ".g", (c["B"], L // 4, 2 * c["dim"]))
    out = m(x)
    (out * g).sum().backward()
    cases.check_summary("out", out, gold, RT, AT)
    cases.check_summary("grad_in/x", x.grad, gold, RT, AT)
    assert _check_grads(m, gold) == 3


@pytest.mark.parametrize("name", list(cases.EMBED_CASES))
def test_patch_embed(name, golden):
    c, gold = cases.EMBED_CASES[name], golden(name)
    m = detgen.fill_(R.PatchEmbed(c["img"], 4, 3, c["dim"]).eval())
    img = cases.randn(name + ".img", (c["B"], 3, c["img"], c["img"]))
    out = m(img)
    g = cases.randn(name + ".g", tuple(out.shape))
    (out * g).sum().backward()
    cases.check_summary("out", out, gold, RT, AT)
    assert _check_grads(m, gold) == 4


def test_roberta_embeddings(golden):
    gold = golden("roberta_emb")
    emb = detgen.fill_(R.RobertaEmbeddings(50265, 768, 514, dropout=0.1).eval())
    b = detgen.synth_batch(3, image_size=8, seed=3)
    assert (b["text_ids"] == 1).any(), "fixture must contain padded rows"
    out = emb(b["text_ids"])
    g = cases.randn("emb.g", tuple(out.shape))
    (out * g).sum().backward()
    cases.check_summary("out", out, gold, RT, AT)
    rows = torch.unique(b["text_ids"])
    cases.check_summary("grad/word_rows", emb.word_embeddings.weight.grad[rows], gold, RT, AT)
    cases.check_summary("grad/position_embeddings", emb.position_embeddings.weight.grad, gold, RT, AT)
    cases.check_summary("grad/token_type_embeddings", emb.token_type_embeddings.weight.grad, gold, RT, 2e-5)
    cases.check_summary("grad/LayerNorm.weight", emb.LayerNorm.weight.grad, gold, RT, AT)


@pytest.mark.parametrize("name", list(cases.ROBERTA_LAYER_CASES))
def test_roberta_layer(name, golden):
    c, gold = cases.ROBERTA_LAYER_CASES[name], golden(name)
    lyr = R.RobertaLayer(768, 12, 3072, 1e-5, 0.1, c["layer_index"], 6, 1024).eval()
    detgen.fill_(lyr)
    h, ext, img, g = cases.roberta_layer_inputs(name)
    h.requires_grad_(True)
    if img is not None:
        img.requires_grad_(True)
    out = lyr(h, ext, encoder_hidden_states=img, last_norm=c["last_norm"])[0]
    (out * g).sum().backward()
    cases.check_summary("out", out, gold, RT, AT)
    cases.check_summary("grad_in/h", h.grad, gold, RT, AT)
    if img is not None:
        cases.check_summary("grad_in/img", img.grad, gold, RT, AT)
    assert _check_grads(lyr, gold) >= 16


@pytest.mark.parametrize("name", ["path_tiny", "path_swin_t", "path_swin_b"])
def test_fused_path(name, golden):
    pc, gold = cases.PATH_CASES[name], golden(name)
    m = detgen.fill_(R.FiberRef(pc["config"]).eval())
    c = m.config
    b = detgen.synth_batch(pc["B"], c["image_size"], c["max_text_len"], c["vocab_size"], seed=1,
                           min_len=min(8, c["max_text_len"] // 2))
    rt, at = 1e-4, 1e-5      # 24 blocks deep: accumulation-order noise grows past the per-op 1e-5
    with torch.set_grad_enabled(pc["grads"]):
        o = m.infer(b, mask_text=True)
        for k in ("text_feats", "image_feats", "cls_feats"):
            cases.check_summary("mlm/" + k, o[k], gold, rt, at)
        mlm = m.compute_mlm(b)["mlm_loss"]
        itm_out = m.compute_itm(b, b["itm_labels"])
        itm = itm_out["itm_loss"]
        assert abs(mlm.item() - float(gold["mlm_loss"])) < 1e-4
        assert abs(itm.item() - float(gold["itm_loss"])) < 1e-4
        cases.check_summary("itm_logits", itm_out["itm_logits"], gold, rt, at)
        if pc["grads"]:
            (mlm + itm).backward()
            unused = set(gold["unused_params"].tolist())
            for n, p in m.named_parameters():
                if n in unused:
                    assert p.grad is None or float(p.grad.abs().max()) == 0.0, f"{n} should be unused"
                else:
                    gn = float(gold[f"gradnorm/{n}"])
                    assert abs(p.grad.double().norm().item() - gn) <= 2e-3 * gn + 1e-7, n
            params = dict(m.named_parameters())
            for key in gold:
                if key.startswith("grad/") and key.endswith("/sub"):
                    n = key[len("grad/"):-len("/sub")]
                    cases.check_summary("grad/" + n, params[n].grad, gold, 2e-3, 1e-6)


@pytest.mark.parametrize("name", list(cases.VQA_CASES))
def test_vqa_finetune_path(name, golden):
    """VQAv2 head at 576^2 (N = 324 windows, 50 text tokens): oracle vs the reference's own compute_vqa."""
    pc, gold = cases.VQA_CASES[name], golden(name)
    m = detgen.fill_(R.FiberRef(pc["config"]).eval())
    c = m.config
    b = detgen.synth_batch(pc["B"], c["image_size"], c["max_text_len"], c["vocab_size"], seed=2,
                           min_len=min(8, c["max_text_len"] // 2))
    b.update(detgen.synth_vqa(pc["B"], c["vqav2_label_size"], seed=2))
    out = m.compute_vqa(b)
    for k in ("text_feats", "image_feats", "cls_feats", "vqa_logits", "vqa_targets"):
        cases.check_summary(k, out[k], gold, 1e-4, 1e-5)
    gl = float(gold["vqa_loss"])
    assert abs(out["vqa_loss"].item() - gl) < 1e-5 * gl + 1e-4
    out["vqa_loss"].backward()
    unused = set(gold["unused_params"].tolist())
    for n, p in m.named_parameters():
        if n in unused:
            assert p.grad is None or float(p.grad.abs().max()) == 0.0, f"{n} should be unused"
        else:
            gn = float(gold[f"gradnorm/{n}"])
            assert abs(p.grad.double().norm().item() - gn) <= 2e-3 * gn + 1e-7, n
    params = dict(m.named_parameters())
    for key in gold:
        if key.startswith("grad/") and key.endswith("/sub"):
            n = key[len("grad/"):-len("/sub")]
            cases.check_summary("grad/" + n, params[n].grad, gold, 2e-3, 1e-6)


def _itc_setup(m):
    with torch.no_grad():
        for bn in ("image_queue", "text_queue", "image_input_queue"):
            getattr(m, bn).copy_(cases.randn("itcq." + bn, tuple(getattr(m, bn).shape)))


@pytest.mark.parametrize("name", list(cases.ITC_CASES))
def test_itc_pretrain_steps(name, golden):
    """MLM + ITC (feature queues) + hard-negative ITM over two training steps: oracle vs the reference's own
    compute_itc / compute_itm_hardneg / _dequeue_and_enqueue (negative draws replayed from the fixture)."""
    pc, gold = cases.ITC_CASES[name], golden(name)
    m = detgen.fill_(R.FiberRef(pc["config"]).train())
    _itc_setup(m)
    c = m.config
    for step, seed in enumerate((3, 4)):
        b = detgen.synth_batch(pc["B"], c["image_size"], c["max_text_len"], c["vocab_size"], seed=seed,
                               min_len=min(8, c["max_text_len"] // 2))
        m.zero_grad(set_to_none=True)
        mlm = m.compute_mlm(b)["mlm_loss"]
        neg = (torch.from_numpy(gold[f"s{step}/image_neg_idx"]), torch.from_numpy(gold[f"s{step}/text_neg_idx"]))
        out, negs = m.compute_itc(b, neg)
        m.dequeue_and_enqueue(out["image_cls"], out["text_cls"], b["image"][0], b["text_ids"], b["text_masks"])
        itm = m.compute_itm_hardneg(b, *negs)
        for k, v in (("mlm_loss", mlm), ("itc_loss", out["itc_loss"]), ("itm_loss", itm["itm_loss"])):
            g = float(gold[f"s{step}/{k}"])
            assert abs(v.item() - g) < 1e-4 * max(1.0, abs(g)), (step, k, v.item(), g)
        cases.check_summary(f"s{step}/itm_logits", itm["itm_logits"], gold, 1e-4, 1e-5)
        assert int(m.queue_ptr) == int(gold[f"s{step}/queue_ptr"]) and int(m.queue_total) == int(gold[f"s{step}/queue_total"])
        for bn in ("image_queue", "text_queue", "image_input_queue", "text_input_queue", "text_input_mask_queue"):
            cases.check_summary(f"s{step}/{bn}", getattr(m, bn).float(), gold, 1e-4, 1e-5)
    (mlm + out["itc_loss"] + itm["itm_loss"]).backward()
    unused = set(gold["unused_params"].tolist())
    for n, p in m.named_parameters():
        if n in unused:
            assert p.grad is None or float(p.grad.abs().max()) == 0.0, f"{n} should be unused"
        else:
            gn = float(gold[f"gradnorm/{n}"])
            assert abs(p.grad.double().norm().item() - gn) <= 2e-3 * gn + 1e-7, n
    params = dict(m.named_parameters())
    for key in gold:
        if key.startswith("grad/") and key.endswith("/sub"):
            n = key[len("grad/"):-len("/sub")]
            cases.check_summary("grad/" + n, params[n].grad, gold, 2e-3, 1e-6)


def test_state_dict_keys_match_reference_layout():
    """Key names of the oracle tree follow the reference's checkpoint layout."""
    m = R.FiberRef(cases.SWIN_T)
    sd = m.state_dict()
    for k in ["vit_model.patch_embed.proj.weight", "vit_model.layers.0.blocks.1.attn_mask",
              "vit_model.layers.2.blocks.0.attn.relative_position_index",
              "vit_model.layers.3.blocks.0.attn.qkv_text_i2t.weight", "vit_model.layers.3.blocks.1.attn.alpha_i2t",
              "vit_model.layers.2.downsample.reduction.weight", "vit_model.norm.weight",
              "text_transformer.embeddings.position_ids", "text_transformer.encoder.layer.6.crossattention_t2i.self.key.weight",
              "text_transformer.encoder.layer.11.crossattention_t2i.output.LayerNorm.weight",
              "text_transformer.encoder.layer.0.alpha_t2i", "text_transformer.pooler.dense.weight",
              "mlm_score.transform.LayerNorm.weight", "mlm_score.decoder.weight", "mlm_score.bias", "itm_score.fc.weight",
              "rank_output.weight", "cross_modal_image_pooler_itc.dense.bias"]:
        assert k in sd, k
    assert sd["text_transformer.encoder.layer.6.crossattention_t2i.self.key.weight"].shape == (768, 384)
    assert sd["text_transformer.encoder.layer.10.crossattention_t2i.self.key.weight"].shape == (768, 768)
