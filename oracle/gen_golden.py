"""Generate tests/golden/*.npz by running the REFERENCE modules.  CONTAINER-ONLY script.

    python -m oracle.gen_golden            # needs /root/reference mounted

The reference's swin_transformer.py / roberta.py / heads.py are executed unmodified (by path,
under oracle/shim.py); weights and inputs come from oracle/detgen.py + oracle/cases.py, so the
fixtures hold OUTPUTS only (strided samples + norms).  The fused sequencing driven here is a
transcription of fiber_module.py:310-367 (the LightningModule itself needs pytorch_lightning,
absent in this image).
"""
import os
import sys

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import cases, detgen, shim
from .fiber_ref import DEFAULT_CONFIG, SWIN_VARIANTS

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")


def save(name, d):
    os.makedirs(OUT, exist_ok=True)
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **d)
    print(f"  wrote {name}.npz ({len(d)} arrays)")


def grads_into(out, prefix, module, extra=()):
    for n, p in module.named_parameters():
        if p.grad is not None:
            cases.flatten_summary(f"{prefix}grad/{n}", p.grad, out)
    for n, t in extra:
        cases.flatten_summary(f"{prefix}grad_in/{n}", t.grad, out)


def gen_blocks(sw):
    for name, c in cases.BLOCK_CASES.items():
        sw.DIM_TEXT = c["dim_text"] or 768
        blk = sw.SwinTransformerBlock(c["dim"], c["res"], c["heads"], window_size=c["ws"], shift_size=c["shift"],
                                      dim_text=c["dim_text"]).eval()
        detgen.fill_(blk)
        x, y, ext, g = cases.block_inputs(name)
        x.requires_grad_(True)
        if y is not None:
            y.requires_grad_(True)
        out = blk(x, y, ext)
        (out * g).sum().backward()
        d = {}
        cases.flatten_summary("out", out, d)
        grads_into(d, "", blk, [("x", x)] + ([("y", y)] if y is not None else []))
        save(name, d)


def gen_merge(sw):
    for name, c in cases.MERGE_CASES.items():
        m = sw.PatchMerging(c["res"], c["dim"]).eval()
        detgen.fill_(m)
        L = c["res"][0] * c["res"][1]
        x = cases.randn(name + ".x", (c["B"], L, c["dim"])).requires_grad_(True)
        g = cases.randn(name + ".g", (c["B"], L // 4, 2 * c["dim"]))
        out = m(x)
        (out * g).sum().backward()
        d = {}
        cases.flatten_summary("out", out, d)
        grads_into(d, "", m, [("x", x)])
        save(name, d)


def gen_patch_embed():
    from timm.models.layers import PatchEmbed
    for name, c in cases.EMBED_CASES.items():
        m = PatchEmbed(img_size=c["img"], patch_size=4, in_chans=3, embed_dim=c["dim"], norm_layer=nn.LayerNorm).eval()
        detgen.fill_(m)
        img = cases.randn(name + ".img", (c["B"], 3, c["img"], c["img"]))
        out = m(img)
        g = cases.randn(name + ".g", tuple(out.shape))
        (out * g).sum().backward()
        d = {}
        cases.flatten_summary("out", out, d)
        grads_into(d, "", m)
        save(name, d)


def gen_roberta(rb):
    rb.NUM_FUSE_BLOCK, rb.DIM_IMG = 6, 1024
    # embeddings: real vocab, padded rows
    cfg = shim.roberta_config()
    emb = rb.RobertaEmbeddings(cfg).eval()
    detgen.fill_(emb)
    b = detgen.synth_batch(3, image_size=8, seed=3)
    out = emb(input_ids=b["text_ids"])
    g = cases.randn("emb.g", tuple(out.shape))
    (out * g).sum().backward()
    d = {}
    cases.flatten_summary("out", out, d)
    rows = torch.unique(b["text_ids"])
    cases.flatten_summary("grad/word_rows", emb.word_embeddings.weight.grad[rows], d)
    cases.flatten_summary("grad/position_embeddings", emb.position_embeddings.weight.grad, d)
    cases.flatten_summary("grad/token_type_embeddings", emb.token_type_embeddings.weight.grad, d)
    cases.flatten_summary("grad/LayerNorm.weight", emb.LayerNorm.weight.grad, d)
    cases.flatten_summary("grad/LayerNorm.bias", emb.LayerNorm.bias.grad, d)
    save("roberta_emb", d)

    for name, c in cases.ROBERTA_LAYER_CASES.items():
        lyr = rb.RobertaLayer(cfg, layer_index=c["layer_index"]).eval()
        detgen.fill_(lyr)
        h, ext, img, g = cases.roberta_layer_inputs(name)
        h.requires_grad_(True)
        if img is not None:
            img.requires_grad_(True)
        out = lyr(h, ext, encoder_hidden_states=img, last_norm=c["last_norm"])[0]
        (out * g).sum().backward()
        d = {}
        cases.flatten_summary("out", out, d)
        grads_into(d, "", lyr, [("h", h)] + ([("img", img)] if img is not None else []))
        save(name, d)


class RefText(nn.Module):
    def __init__(self, rb, cfg):
        super().__init__()
        self.embeddings = rb.RobertaEmbeddings(cfg)
        self.encoder = rb.RobertaEncoder(cfg)
        self.pooler = nn.Module()
        self.pooler.dense = nn.Linear(cfg.hidden_size, cfg.hidden_size)


class RefFused(nn.Module):
    """Reference modules wired with FIBERTransformerSS's attribute names (fiber_module.py:27-117)."""

    def __init__(self, sw, rb, heads, config):
        super().__init__()
        c = dict(DEFAULT_CONFIG)
        c.update(config)
        self.c = c
        hs = c["hidden_size"]
        sw.NUM_FUSE_BLOCK = rb.NUM_FUSE_BLOCK = c["num_fuse_block"]
        rb.DIM_IMG = c["input_image_embed_size"]
        sw.DIM_TEXT = c.get("swin_dim_text", 768)
        self.cross_modal_text_transform = nn.Linear(c["input_text_embed_size"], hs)
        self.cross_modal_image_transform = nn.Linear(c["input_image_embed_size"], hs)
        self.cross_modal_text_transform_itc = nn.Linear(c["input_text_embed_size"], hs)
        self.cross_modal_image_transform_itc = nn.Linear(c["input_image_embed_size"], hs)
        dim, depths, nh = c.get("swin_arch") or SWIN_VARIANTS[c["vit"]]
        self.vit_model = sw.SwinTransformer(img_size=c["image_size"], patch_size=4, embed_dim=dim, depths=depths,
                                            num_heads=nh, drop_path_rate=c["drop_path_rate"])
        tcfg = shim.roberta_config(
            vocab_size=c["vocab_size"], hidden_size=c["input_text_embed_size"], num_hidden_layers=c["num_layers"],
            num_attention_heads=c["num_heads"], intermediate_size=c["input_text_embed_size"] * c["mlp_ratio"],
            max_position_embeddings=c["max_position_embeddings"], hidden_dropout_prob=c["text_dropout"],
            attention_probs_dropout_prob=c["text_dropout"])
        self.text_transformer = RefText(rb, tcfg)
        self.cross_modal_image_pooler = heads.Pooler(hs)
        self.cross_modal_text_pooler = heads.Pooler(hs)
        self.cross_modal_image_pooler_itc = heads.Pooler(hs)
        self.cross_modal_text_pooler_itc = heads.Pooler(hs)
        hcfg = shim.roberta_config(vocab_size=c["vocab_size"], hidden_size=hs, layer_norm_eps=1e-12)
        ln = c["loss_names"]
        self.avgpool = nn.AdaptiveAvgPool1d(1)
        self.itc_pooler = c["itc_pooler"]
        if ln.get("itc", 0) > 0:                           # wiring of fiber_module.py:56-67
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
        if ln.get("mlm", 0) > 0:
            self.mlm_score = heads.MLMHead(hcfg)
        if ln.get("itm", 0) > 0:
            self.itm_score = heads.ITMHead(hs * 2)
            self.rank_output = nn.Linear(hs, 1)
        if ln.get("vqa", 0) > 0:                           # wiring of fiber_module.py:149-157
            self.vqa_classifier = nn.Sequential(nn.Linear(hs * 2, hs * 2), nn.LayerNorm(hs * 2), nn.GELU(),
                                                nn.Linear(hs * 2, c["vqav2_label_size"]))

    def infer(self, batch, mask_text=False, img=None, mask_image=False, image_only=False, text_only=False):
        c = self.c
        img = batch["image"][0] if img is None else img
        sfx = "_mlm" if mask_text else ""
        ids, masks = batch["text_ids" + sfx], batch["text_masks"]
        if text_only:                                      # sequencing of fiber_module.py:247-277
            t = self.text_transformer.embeddings(input_ids=ids)
            ext = (1.0 - masks[:, None, None, :].float()) * -10000.0
            for layer in self.text_transformer.encoder.layer:
                t = layer(t, ext)[0]
            t = self.cross_modal_text_transform_itc(t)
            cls = self.cross_modal_text_pooler_itc(t) if self.itc_pooler else t[:, 0]
            return {"text_feats": t, "image_feats": None, "cls_feats": cls / cls.norm(dim=-1, keepdim=True)}
        if image_only:                                     # sequencing of fiber_module.py:279-308
            x = self.vit_model.patch_embed(img)
            for layer in self.vit_model.layers:
                x = layer(x)
            x = self.cross_modal_image_transform_itc(self.vit_model.norm(x))
            avg = self.avgpool(x.transpose(1, 2)).view(x.size(0), 1, -1)
            cls = self.cross_modal_image_pooler_itc(avg) if self.itc_pooler else avg[:, 0]
            return {"text_feats": None, "image_feats": x, "cls_feats": cls / cls.norm(dim=-1, keepdim=True)}
        x = self.vit_model.patch_embed(img)
        for layer in self.vit_model.layers[:2]:
            x = layer(x)
        t = self.text_transformer.embeddings(input_ids=ids)
        ext = (1.0 - masks[:, None, None, :].float()) * -10000.0
        npt = c["num_layers"] - c["num_fuse_block"]
        for layer in self.text_transformer.encoder.layer[:npt]:
            t = layer(t, ext)[0]
        for i, blk in enumerate(self.vit_model.layers[2].blocks):
            if i < 8 + npt:
                x = blk(x)
            else:
                fx = blk(x, t, ext)
                t = self.text_transformer.encoder.layer[i - 8](t, ext, encoder_hidden_states=x)[0]
                x = fx
        x = self.vit_model.layers[2].downsample(x)
        for i, blk in enumerate(self.vit_model.layers[3].blocks):
            fx = blk(x, t, ext)
            t = self.text_transformer.encoder.layer[i + 10](t, ext, encoder_hidden_states=x, last_norm=(i == 0))[0]
            x = fx
        t = self.cross_modal_text_transform(t)
        x = self.cross_modal_image_transform(x)
        ct = self.cross_modal_text_pooler(t)
        ci = self.cross_modal_image_pooler(x.mean(1, keepdim=True))
        return {"text_feats": t, "image_feats": x, "cls_feats": torch.cat([ct, ci], -1),
                "text_labels": batch.get("text_labels" + sfx), "text_ids": ids, "text_masks": masks}


def gen_paths(sw, rb, heads):
    for name, pc in cases.PATH_CASES.items():
        torch.manual_seed(0)
        m = RefFused(sw, rb, heads, pc["config"]).eval()
        detgen.fill_(m)
        c = m.c
        b = detgen.synth_batch(pc["B"], c["image_size"], c["max_text_len"], c["vocab_size"], seed=1,
                               min_len=min(8, c["max_text_len"] // 2))
        d = {}
        with torch.set_grad_enabled(pc["grads"]):
            o = m.infer(b, mask_text=True)
            for k in ("text_feats", "image_feats", "cls_feats"):
                cases.flatten_summary("mlm/" + k, o[k], d)
            logits = m.mlm_score(o["text_feats"])
            mlm = F.cross_entropy(logits.view(-1, c["vocab_size"]), b["text_labels_mlm"].view(-1), ignore_index=-100)
            lab = b["itm_labels"]
            imgs = torch.where(lab.view(-1, 1, 1, 1) == 1, b["image"][0], b["false_image_0"][0])
            o2 = m.infer(b, img=imgs)
            for k in ("text_feats", "image_feats", "cls_feats"):
                cases.flatten_summary("itm/" + k, o2[k], d)
            itm_logits = m.itm_score(o2["cls_feats"])
            itm = F.cross_entropy(itm_logits, lab.long())
            d["mlm_loss"], d["itm_loss"] = np.float64(mlm.item()), np.float64(itm.item())
            cases.flatten_summary("itm_logits", itm_logits, d)
            if pc["grads"]:
                (mlm + itm).backward()
                unused = []
                for n, p in m.named_parameters():
                    if p.grad is None:
                        unused.append(n)
                    else:
                        d[f"gradnorm/{n}"] = np.float64(p.grad.double().norm().item())
                d["unused_params"] = np.array(unused)
                for n in ("vit_model.patch_embed.proj.weight", "text_transformer.encoder.layer.11.alpha_t2i",
                          "vit_model.layers.3.blocks.1.attn.alpha_i2t",
                          "vit_model.layers.3.blocks.0.attn.relative_position_bias_table",
                          "vit_model.layers.0.blocks.1.attn.relative_position_bias_table",
                          "text_transformer.embeddings.position_embeddings.weight"):
                    cases.flatten_summary("grad/" + n, dict(m.named_parameters())[n].grad, d)
        print(f"  {name}: mlm {mlm.item():.6f} itm {itm.item():.6f}")
        save(name, d)


class _Metric:
    def __call__(self, *a):
        return a[0].detach() if len(a) == 1 else torch.zeros(())


def gen_vqa(sw, rb, heads):
    """The reference's own objectives.compute_vqa driven over the reference modules (pl_module duck-typed)."""
    obj = shim._load("objectives", os.path.join(shim.MODS, "objectives.py"), "_fiber_reference_modules")
    for name, pc in cases.VQA_CASES.items():
        torch.manual_seed(0)
        m = RefFused(sw, rb, heads, pc["config"]).eval()
        detgen.fill_(m)
        c = m.c
        m.hparams = type("H", (), {"config": c})()
        m.device = torch.device("cpu")
        m.log = lambda *a, **k: None
        for ph in ("train", "val"):
            setattr(m, f"{ph}_vqa_loss", _Metric())
            setattr(m, f"{ph}_vqa_score", _Metric())
        b = detgen.synth_batch(pc["B"], c["image_size"], c["max_text_len"], c["vocab_size"], seed=2,
                               min_len=min(8, c["max_text_len"] // 2))
        b.update(detgen.synth_vqa(pc["B"], c["vqav2_label_size"], seed=2))
        d = {}
        feats = {}
        inner = m.infer
        m.infer = lambda batch, **kw: feats.setdefault("o", inner(batch, **kw))
        ret = obj.compute_vqa(m, b)
        for k in ("text_feats", "image_feats", "cls_feats"):
            cases.flatten_summary(k, feats["o"][k], d)
        cases.flatten_summary("vqa_logits", ret["vqa_logits"], d)
        cases.flatten_summary("vqa_targets", ret["vqa_targets"], d)
        d["vqa_loss"] = np.float64(ret["vqa_loss"].item())
        ret["vqa_loss"].backward()
        unused = []
        for n, p in m.named_parameters():
            if p.grad is None:
                unused.append(n)
            else:
                d[f"gradnorm/{n}"] = np.float64(p.grad.double().norm().item())
        d["unused_params"] = np.array(unused)
        for n in ("vit_model.patch_embed.proj.weight", "vit_model.layers.2.blocks.15.attn.relative_position_bias_table",
                  "vit_model.layers.0.blocks.1.attn.relative_position_bias_table",
                  "vit_model.layers.3.blocks.1.attn.alpha_i2t", "vqa_classifier.3.bias"):
            cases.flatten_summary("grad/" + n, dict(m.named_parameters())[n].grad, d)
        print(f"  {name}: vqa_loss {ret['vqa_loss'].item():.6f}")
        save(name, d)


def _reference_functions(path, names):
    """Compile the named top-level functions / methods straight out of a reference source file (no copy is kept)."""
    import ast
    tree = ast.parse(open(path).read())
    found = {n.name: n for n in ast.walk(tree) if isinstance(n, ast.FunctionDef) and n.name in names}
    ns = {"torch": torch}
    exec(compile(ast.Module(body=[found[n] for n in names], type_ignores=[]), path, "exec"), ns)
    return ns


def gen_itc(sw, rb, heads):
    """task_pretrain_mlm_itm_itc: the reference's compute_mlm / compute_itc / compute_itm_hardneg and its queue update
    (`_dequeue_and_enqueue`, `concat_all_gather`, compiled from fiber_module.py) over two consecutive training steps."""
    import torch.distributed as dist
    obj = shim._load("objectives", os.path.join(shim.MODS, "objectives.py"), "_fiber_reference_modules")
    fm = _reference_functions(os.path.join(shim.MODS, "fiber_module.py"), ["concat_all_gather", "_dequeue_and_enqueue"])
    if not dist.is_initialized():
        dist.init_process_group("gloo", init_method="tcp://127.0.0.1:29533", rank=0, world_size=1)
    for name, pc in cases.ITC_CASES.items():
        torch.manual_seed(0)
        m = RefFused(sw, rb, heads, pc["config"]).train()
        detgen.fill_(m)
        with torch.no_grad():
            m.temp.fill_(0.07)
            for bn in ("image_queue", "text_queue", "image_input_queue"):
                getattr(m, bn).copy_(cases.randn("itcq." + bn, tuple(getattr(m, bn).shape)))
        c = m.c
        m.hparams = type("H", (), {"config": c})()
        m.device = torch.device("cpu")
        m.log = lambda *a, **k: None
        m._dequeue_and_enqueue = lambda *a: fm["_dequeue_and_enqueue"](m, *a)
        for ph in ("train", "val"):
            for t in ("mlm", "itc", "itm"):
                setattr(m, f"{ph}_{t}_loss", _Metric())
                setattr(m, f"{ph}_{t}_accuracy", _Metric())
        d = {}
        real_multinomial = torch.multinomial
        for step, seed in enumerate((3, 4)):
            b = detgen.synth_batch(pc["B"], c["image_size"], c["max_text_len"], c["vocab_size"], seed=seed,
                                   min_len=min(8, c["max_text_len"] // 2))
            draws = []

            def recording(w, n, *a, **k):
                r = real_multinomial(w, n, *a, **k)
                draws.append(int(r.reshape(-1)[0]))
                return r
            torch.manual_seed(100 + step)
            m.zero_grad(set_to_none=True)
            mlm = obj.compute_mlm(m, b)
            torch.multinomial = recording
            try:
                ret_itc, image_neg, text_neg, text_mask_neg = obj.compute_itc(m, b)
            finally:
                torch.multinomial = real_multinomial
            itm = obj.compute_itm_hardneg(m, dict(b), image_neg, text_neg, text_mask_neg)
            B = pc["B"]
            d[f"s{step}/image_neg_idx"], d[f"s{step}/text_neg_idx"] = np.array(draws[:B]), np.array(draws[B:])
            for k, v in (("mlm_loss", mlm["mlm_loss"]), ("itc_loss", ret_itc["itc_loss"]), ("itm_loss", itm["itm_loss"])):
                d[f"s{step}/{k}"] = np.float64(v.item())
            cases.flatten_summary(f"s{step}/itm_logits", itm["itm_logits"], d)
            d[f"s{step}/queue_ptr"], d[f"s{step}/queue_total"] = np.int64(int(m.queue_ptr)), np.int64(int(m.queue_total))
            for bn in ("image_queue", "text_queue", "image_input_queue", "text_input_queue", "text_input_mask_queue"):
                cases.flatten_summary(f"s{step}/{bn}", getattr(m, bn).float(), d)
            print(f"  {name} step {step}: mlm {mlm['mlm_loss'].item():.5f} itc {ret_itc['itc_loss'].item():.5f} "
                  f"itm {itm['itm_loss'].item():.5f}  neg {draws}")
        (mlm["mlm_loss"] + ret_itc["itc_loss"] + itm["itm_loss"]).backward()
        unused = []
        for n, p in m.named_parameters():
            if p.grad is None:
                unused.append(n)
            else:
                d[f"gradnorm/{n}"] = np.float64(p.grad.double().norm().item())
        d["unused_params"] = np.array(unused)
        for n in ("temp", "vit_model.norm.weight", "cross_modal_text_pooler_itc.dense.weight",
                  "vit_model.layers.3.blocks.1.attn.alpha_i2t", "vit_model.patch_embed.proj.weight"):
            cases.flatten_summary("grad/" + n, dict(m.named_parameters())[n].grad, d)
        save(name, d)


def gen_schedule():
    """G8: the reference's set_schedule (compiled out of fiber_utils.py) on a module with the FIBER parameter names: which
    parameter lands in which optimizer group, group lr / weight decay, and the LambdaLR factors over a short run -- for the
    step-bounded pre-training config and the epoch-bounded (max_steps=None) fine-tuning config."""
    import types
    from transformers.optimization import get_cosine_schedule_with_warmup, get_polynomial_decay_schedule_with_warmup
    from .fiber_ref import FiberRef
    fn = _reference_functions(os.path.join(shim.MODS, "fiber_utils.py"), ["set_schedule"])
    ns = fn["set_schedule"].__globals__
    ns.update(AdamW=torch.optim.AdamW, get_cosine_schedule_with_warmup=get_cosine_schedule_with_warmup,
              get_polynomial_decay_schedule_with_warmup=get_polynomial_decay_schedule_with_warmup)
    d = {}
    for tag, over, trainer in (
            ("pretrain", dict(loss_names={"mlm": 1, "itm": 1}, learning_rate=1e-5, lr_mult_head=5, lr_mult_cross_modal=5,
                              warmup_steps=0.1), dict(max_steps=200, max_epochs=None, accumulate_grad_batches=1, n_batches=0)),
            ("vqa", dict(loss_names={"vqa": 1}, learning_rate=2e-5, lr_mult_head=50, lr_mult_cross_modal=5, warmup_steps=0.1,
                         vqav2_label_size=17), dict(max_steps=None, max_epochs=10, accumulate_grad_batches=2, n_batches=37))):
        cfg = dict(cases.TINY, weight_decay=0.01, end_lr=0, decay_power=1, optim_type="adamw", **over)
        m = FiberRef(cfg)
        m.hparams = types.SimpleNamespace(config=cfg)
        dl = list(range(trainer["n_batches"]))
        m.trainer = types.SimpleNamespace(max_steps=trainer["max_steps"], max_epochs=trainer["max_epochs"],
                                          accumulate_grad_batches=trainer["accumulate_grad_batches"],
                                          datamodule=types.SimpleNamespace(train_dataloader=lambda dl=dl: dl))
        (opt,), (sch,) = fn["set_schedule"](m)
        names = {id(p): n for n, p in m.named_parameters()}
        for gi, g in enumerate(opt.param_groups):
            d[f"{tag}/group{gi}/names"] = np.array(sorted(names[id(p)] for p in g["params"]))
            d[f"{tag}/group{gi}/lr"], d[f"{tag}/group{gi}/wd"] = np.float64(g["initial_lr"]), np.float64(g["weight_decay"])
        lrs = []
        for _ in range(60):
            lrs.append([g["lr"] for g in opt.param_groups])
            opt.step()
            sch["scheduler"].step()
        d[f"{tag}/lrs"] = np.array(lrs)
        d[f"{tag}/all_names"] = np.array(sorted(n for n, _ in m.named_parameters()))
    save("schedule", d)


def gen_adapt():
    """swin_helpers.swin_adapt_position_encoding run on a seeded fake state dict."""
    hp = sys.modules["_fiber_reference_modules.swin_helpers"]
    ac = cases.ADAPT_CASE
    side = 2 * (ac["before"] // 32) - 1
    sd = {
        "vit_model.layers.0.blocks.0.attn.relative_position_bias_table": cases.randn("adapt.t0", (side * side, ac["heads"])),
        "vit_model.layers.2.blocks.3.attn.relative_position_bias_table": cases.randn("adapt.t1", (side * side, 2 * ac["heads"])),
        "vit_model.layers.0.blocks.0.attn.relative_position_index": torch.zeros(4, 4, dtype=torch.long),
        "vit_model.layers.0.blocks.1.attn_mask": torch.zeros(2, 4, 4),
        "vit_model.layers.0.blocks.0.norm1.weight": torch.ones(8),
    }
    out = hp.swin_adapt_position_encoding(dict(sd), before=ac["before"], after=ac["after"])
    d = {"keys": np.array(sorted(out.keys()))}
    for k, v in out.items():
        if k.endswith("relative_position_bias_table"):
            d["table/" + k] = v.numpy()
    save("adapt_pos", d)


def main():
    torch.set_num_threads(8)
    sw, rb = shim.load_reference()
    heads = shim._load("heads", os.path.join(shim.MODS, "heads.py"), "_fiber_reference_modules")
    only = set(sys.argv[1:])
    if not only or "blocks" in only:
        gen_blocks(sw)
    if not only or "merge" in only:
        gen_merge(sw)
    if not only or "embed" in only:
        gen_patch_embed()
    if not only or "roberta" in only:
        gen_roberta(rb)
    if not only or "paths" in only:
        gen_paths(sw, rb, heads)
    if not only or "vqa" in only:
        gen_vqa(sw, rb, heads)
    if not only or "itc" in only:
        gen_itc(sw, rb, heads)
    if not only or "schedule" in only:
        gen_schedule()
    if not only or "adapt" in only:
        gen_adapt()


if __name__ == "__main__":
    main()
