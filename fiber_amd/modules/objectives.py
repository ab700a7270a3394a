"""Loss functions on the metric path: compute_mlm / compute_itm / compute_itm_hardneg / compute_itc / compute_vqa /
init_weights (reference coarse_grained/fiber/modules/objectives.py:17-213, 502-510).  Same call signatures and return keys."""
import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import ops


def _mlm_ce(logits, labels, head=None, owner=None):
    """F.cross_entropy(logits, labels, ignore_index=-100) (objectives.py:24-28) on the HIP kernel for bf16 device logits.

    The 50265-way logits are bf16 (2 B instead of 8 MB per sample of HBM traffic); log-sum-exp averages their rounding away,
    but the LABEL logit enters the loss directly, and 2^-9 relative rounding of a logit of magnitude 5-10 is 1e-2..2e-2 per
    token -- the dominant term of the loss-curve gap to the fp32 reference (tests/test_hip_modules.py, 50-step curve).  With
    `head` (the MLMHead, after its forward) the label logit is recomputed in fp32 from the head's hidden state -- a gathered
    dot product per token -- and substituted into the loss VALUE; gradients are unchanged (they are the same function)."""
    x16 = logits.to(torch.bfloat16)
    loss = ops.cross_entropy(x16, labels, -100)
    if owner is not None and getattr(x16, "_fiber_argmax", None) is not None:
        owner._fiber_argmax = x16._fiber_argmax              # the arg max of the labelled rows rides on the logits the metric will be given
    if head is not None and head.last_hidden is not None:
        with torch.no_grad():
            h = head.last_hidden.reshape(-1, head.last_hidden.shape[-1])
            valid = labels != -100
            lab = labels.clamp(min=0)
            z16 = logits.gather(1, lab[:, None]).squeeze(1).float()
            z32 = (h.float() * head.decoder.weight[lab]).sum(-1) + head.bias[lab]
            corr = ((z16 - z32) * valid).sum() / valid.sum().clamp(min=1)
        loss = loss + corr                                  # a constant w.r.t. autograd: value correction only
    return loss


def _mlm_head(pl_module, text_feats, labels):
    """MLM head + loss (objectives.py:17-33: `mlm_score(text_feats)`, cross entropy with ignore_index -100).

    85 % of the rows carry the ignore label: the reference still pushes them through the 50265-way decoder (forward, dX, dW) and
    writes a zero gradient row for each.  With `mlm_compact_rows` (default OFF: the reference's output shapes) only the rows that carry a label go through the head:
    same loss (the mean runs over labelled rows either way), same gradients (an ignored row's logit gradient is exactly zero), same
    accuracy (the metric drops ignored rows).  The row count is data dependent, so the selection costs ONE host sync per step; it is
    skipped while a hipGraph is being captured (static shapes) and when no row is labelled.  Returns (loss, logits, labels) --
    the logits / labels of the labelled rows only ([n, V] / [n]) on the compact path, the full tensors otherwise."""
    cfg = pl_module.hparams.config
    V = cfg["vocab_size"]
    flat = labels.reshape(-1)
    if cfg.get("mlm_compact_rows", False) and text_feats.is_cuda and not torch.cuda.is_current_stream_capturing():
        keep = (flat != -100).nonzero(as_tuple=True)[0]
        if keep.numel() > 0:
            feats = text_feats.reshape(-1, text_feats.shape[-1]).index_select(0, keep)
            lab = flat.index_select(0, keep)
            logits = pl_module.mlm_score(feats)
            return _mlm_ce(logits, lab, pl_module.mlm_score, owner=logits), logits, lab
    logits = pl_module.mlm_score(text_feats)
    return _mlm_ce(logits.view(-1, V), flat, pl_module.mlm_score, owner=logits), logits, labels


def compute_mlm(pl_module, batch):
    infer = pl_module.infer(batch, mask_text=True, mask_image=False)
    mlm_loss, mlm_logits, mlm_labels = _mlm_head(pl_module, infer["text_feats"], infer["text_labels"])
    ret = {"mlm_loss": mlm_loss, "mlm_logits": mlm_logits, "mlm_labels": mlm_labels, "mlm_ids": infer["text_ids"]}
    phase = "train" if pl_module.training else "val"
    loss = getattr(pl_module, f"{phase}_mlm_loss")(ret["mlm_loss"])
    acc = getattr(pl_module, f"{phase}_mlm_accuracy")(ret["mlm_logits"], ret["mlm_labels"])
    pl_module.log(f"mlm/{phase}/loss", loss)
    pl_module.log(f"mlm/{phase}/accuracy", acc)
    return ret


def compute_itm(pl_module, batch, itm_labels=None):
    """`itm_labels` (optional) pins the true/false permutation for parity tests; by default it is drawn with
    torch.randperm exactly as the reference does (objectives.py:47-48)."""
    pos_len = len(batch["text"]) // 2
    neg_len = len(batch["text"]) - pos_len
    if itm_labels is None:
        itm_labels = torch.cat([torch.ones(pos_len), torch.zeros(neg_len)]).to(pl_module.device)
        itm_labels = itm_labels[torch.randperm(itm_labels.size(0))]
    else:
        itm_labels = itm_labels.to(pl_module.device).float()
    sel = itm_labels.view(-1, 1, 1, 1) == 1
    itm_images = [torch.where(sel, bti, bfi) for bti, bfi in zip(batch["image"], batch["false_image_0"])]
    batch = {k: v for k, v in batch.items()}
    batch["image"] = itm_images
    infer = pl_module.infer(batch, mask_text=False, mask_image=False)
    itm_logits = pl_module.itm_score(infer["cls_feats"])
    itm_loss = F.cross_entropy(itm_logits, itm_labels.long())
    ret = {"itm_loss": itm_loss, "itm_logits": itm_logits, "itm_labels": itm_labels}
    phase = "train" if pl_module.training else "val"
    loss = getattr(pl_module, f"{phase}_itm_loss")(ret["itm_loss"])
    acc = getattr(pl_module, f"{phase}_itm_accuracy")(ret["itm_logits"], ret["itm_labels"])
    pl_module.log(f"itm/{phase}/loss", loss)
    pl_module.log(f"itm/{phase}/accuracy", acc)
    return ret


def compute_mlm_itm_fused(pl_module, batch, itm_labels=None):
    """compute_mlm + compute_itm (objectives.py:17-75) evaluated in ONE fused-backbone pass over the 2B concatenated
    samples [masked-text pairs ; true/false-image pairs].  Every op on the path is per-sample (LayerNorm, window /
    cross attention, per-sample DropPath), so the two halves are exactly the two separate infer() calls of the
    reference; one pass halves the kernel-launch count and doubles the GEMM M dimension.  Same return keys."""
    B = len(batch["text"])
    pos_len = B // 2
    if itm_labels is None:
        itm_labels = torch.cat([torch.ones(pos_len), torch.zeros(B - pos_len)]).to(pl_module.device)
        itm_labels = itm_labels[torch.randperm(itm_labels.size(0))]
    else:
        itm_labels = itm_labels.to(pl_module.device).float()
    true_img, false_img = batch["image"][0], batch["false_image_0"][0]
    fused = {
        # [true ; where(label == 1, true, false)] as its two sources: the patch embedding gathers from them (ops.ImagePair)
        "image": [ops.ImagePair(true_img, false_img, itm_labels == 1)],
        "text_ids": torch.cat([batch["text_ids_mlm"], batch["text_ids"]], 0),
        "text_labels": torch.cat([batch["text_labels_mlm"], batch["text_labels"]], 0),
        "text_masks": torch.cat([batch["text_masks"], batch["text_masks"]], 0),
    }
    infer = pl_module.infer(fused, mask_text=False, mask_image=False)
    mlm_loss, mlm_logits, mlm_labels = _mlm_head(pl_module, infer["text_feats"][:B], batch["text_labels_mlm"])
    itm_logits = pl_module.itm_score(infer["cls_feats"][B:])
    itm_loss = F.cross_entropy(itm_logits, itm_labels.long())
    ret = {"mlm_loss": mlm_loss, "mlm_logits": mlm_logits, "mlm_labels": mlm_labels, "mlm_ids": batch["text_ids_mlm"],
           "itm_loss": itm_loss, "itm_logits": itm_logits, "itm_labels": itm_labels}
    phase = "train" if pl_module.training else "val"
    for task in ("mlm", "itm"):
        loss = getattr(pl_module, f"{phase}_{task}_loss")(ret[f"{task}_loss"])
        acc = getattr(pl_module, f"{phase}_{task}_accuracy")(ret[f"{task}_logits"], ret[f"{task}_labels"])
        pl_module.log(f"{task}/{phase}/loss", loss)
        pl_module.log(f"{task}/{phase}/accuracy", acc)
    return ret


def compute_itc(pl_module, batch, neg_override=None):
    """Image-text contrastive loss against the local batch + the feature queues, and the hard-negative draw that feeds
    compute_itm_hardneg (objectives.py:119-180, after ALBEF).  The reference draws one negative per row with a Python loop
    of `torch.multinomial(...).item()` (2B host syncs); here both draws are one batched multinomial each and the gathers stay
    on the device -- the same distribution, a different consumption of the RNG stream.  `neg_override = (image_idx, text_idx)`
    pins the draw for parity tests."""
    with torch.no_grad():
        pl_module.temp.clamp_(0.001, 1.0)
    infer_image = pl_module.infer(batch, mask_image=False, mask_text=False, image_only=True)
    infer_text = pl_module.infer(batch, mask_image=False, mask_text=False, text_only=True)
    image_feat, text_feat = infer_image["cls_feats"].float(), infer_text["cls_feats"].float()
    image_feat_all = torch.cat([image_feat.t().detach(), pl_module.image_queue.detach().float()], dim=1)
    text_feat_all = torch.cat([text_feat.t().detach(), pl_module.text_queue.detach().float()], dim=1)
    sim_i2t = ops.lib_linear(image_feat, text_feat_all.t().contiguous()) / pl_module.temp
    sim_t2i = ops.lib_linear(text_feat, image_feat_all.t().contiguous()) / pl_module.temp
    bs = image_feat.size(0)
    diag = torch.arange(bs, device=sim_i2t.device)
    loss_i2t = -F.log_softmax(sim_i2t, dim=1)[diag, diag].mean()         # sim_targets = identity on the first bs columns
    loss_t2i = -F.log_softmax(sim_t2i, dim=1)[diag, diag].mean()
    loss_itc = (loss_i2t + loss_t2i) / 2.0

    total = int(pl_module.queue_total)      # NB as in the reference this may exceed queue_size once the queue has wrapped
    pool = min(bs + total, sim_i2t.shape[1])
    tot_image = torch.cat([batch["image"][0], pl_module.image_input_queue[:total].to(batch["image"][0].dtype)], dim=0)
    tot_text = torch.cat([batch["text_ids"], pl_module.text_input_queue[:total]], dim=0)
    tot_text_mask = torch.cat([batch["text_masks"], pl_module.text_input_mask_queue[:total]], dim=0)
    if neg_override is not None:
        img_idx, txt_idx = (torch.as_tensor(v, device=sim_i2t.device) for v in neg_override)
    else:
        with torch.no_grad():
            w_i2t = F.softmax(sim_i2t[:, :pool], dim=1)
            w_t2i = F.softmax(sim_t2i[:, :pool], dim=1)
            w_i2t[diag, diag] = 0
            w_t2i[diag, diag] = 0
            img_idx = torch.multinomial(w_t2i + 1e-9, 1).squeeze(1)
            txt_idx = torch.multinomial(w_i2t + 1e-9, 1).squeeze(1)
    image_neg, text_neg, text_mask_neg = tot_image[img_idx], tot_text[txt_idx], tot_text_mask[txt_idx]

    if pl_module.training:
        pl_module._dequeue_and_enqueue(infer_image["cls_feats"].detach().clone(), infer_text["cls_feats"].detach().clone(),
                                       batch["image"][0].clone(), batch["text_ids"].clone(), batch["text_masks"].clone())
    ret = {"itc_loss": loss_itc}
    phase = "train" if pl_module.training else "val"
    loss = getattr(pl_module, f"{phase}_itc_loss")(ret["itc_loss"])
    pl_module.log(f"itc/{phase}/loss", loss)
    return ret, image_neg, text_neg, text_mask_neg


def compute_itm_hardneg(pl_module, batch, image_neg, text_neg, text_mask_neg):
    """ITM over 3B pairs: the B true pairs, (image, hard-negative text) and (hard-negative image, text)
    (objectives.py:78-116).  The reference overwrites batch["image"/"text_ids"/"text_masks"] in place; a shallow copy is
    used here so later objectives still see the loader's batch."""
    pos_len = len(batch["text"])
    itm_labels = torch.cat([torch.ones(pos_len), torch.zeros(2 * pos_len)]).to(pl_module.device)
    img = batch["image"][0]
    b3 = {k: v for k, v in batch.items()}
    b3["image"] = [torch.cat([img, img, image_neg.to(img.dtype)], dim=0)]
    b3["text_ids"] = torch.cat([batch["text_ids"], text_neg, batch["text_ids"]], dim=0)
    b3["text_masks"] = torch.cat([batch["text_masks"], text_mask_neg, batch["text_masks"]], dim=0)
    b3["text_labels"] = torch.cat([batch["text_labels"]] * 3, dim=0)
    infer = pl_module.infer(b3, mask_text=False, mask_image=False)
    itm_logits = pl_module.itm_score(infer["cls_feats"])
    itm_loss = F.cross_entropy(itm_logits, itm_labels.long())
    ret = {"itm_loss": itm_loss, "itm_logits": itm_logits, "itm_labels": itm_labels}
    phase = "train" if pl_module.training else "val"
    loss = getattr(pl_module, f"{phase}_itm_loss")(ret["itm_loss"])
    acc = getattr(pl_module, f"{phase}_itm_accuracy")(ret["itm_logits"], ret["itm_labels"])
    pl_module.log(f"itm/{phase}/loss", loss)
    pl_module.log(f"itm/{phase}/accuracy", acc)
    return ret


def compute_mlm_itm_hardneg_fused(pl_module, batch, image_neg, text_neg, text_mask_neg):
    """compute_mlm + compute_itm_hardneg (objectives.py:17-33, 78-116) in ONE fused-backbone pass over 4B samples
    [B masked-text pairs ; B true pairs ; B (image, hard-negative text) ; B (hard-negative image, text)] -- the same
    per-sample computation as the reference's two infer() calls, with a quarter fewer launches and larger GEMMs."""
    B = len(batch["text"])
    img = batch["image"][0]
    fused = {
        "image": [torch.cat([img, img, img, image_neg.to(img.dtype)], 0)],
        "text_ids": torch.cat([batch["text_ids_mlm"], batch["text_ids"], text_neg, batch["text_ids"]], 0),
        "text_labels": torch.cat([batch["text_labels_mlm"]] + [batch["text_labels"]] * 3, 0),
        "text_masks": torch.cat([batch["text_masks"], batch["text_masks"], text_mask_neg, batch["text_masks"]], 0),
    }
    infer = pl_module.infer(fused, mask_text=False, mask_image=False)
    mlm_loss, mlm_logits, mlm_labels = _mlm_head(pl_module, infer["text_feats"][:B], batch["text_labels_mlm"])
    itm_labels = torch.cat([torch.ones(B), torch.zeros(2 * B)]).to(pl_module.device)
    itm_logits = pl_module.itm_score(infer["cls_feats"][B:])
    itm_loss = F.cross_entropy(itm_logits, itm_labels.long())
    ret = {"mlm_loss": mlm_loss, "mlm_logits": mlm_logits, "mlm_labels": mlm_labels, "mlm_ids": batch["text_ids_mlm"],
           "itm_loss": itm_loss, "itm_logits": itm_logits, "itm_labels": itm_labels}
    phase = "train" if pl_module.training else "val"
    for task in ("mlm", "itm"):
        loss = getattr(pl_module, f"{phase}_{task}_loss")(ret[f"{task}_loss"])
        acc = getattr(pl_module, f"{phase}_{task}_accuracy")(ret[f"{task}_logits"], ret[f"{task}_labels"])
        pl_module.log(f"{task}/{phase}/loss", loss)
        pl_module.log(f"{task}/{phase}/accuracy", acc)
    return ret


def compute_vqa(pl_module, batch):
    """VQAv2 fine-tune head (objectives.py:182-213): soft-target BCE over the answer vocabulary, scaled by its size."""
    infer = pl_module.infer(batch, mask_text=False, mask_image=False)
    vqa_logits = pl_module.vqa_classifier(infer["cls_feats"])
    n_ans = pl_module.hparams.config["vqav2_label_size"]
    vqa_targets = torch.zeros(len(vqa_logits), n_ans, device=vqa_logits.device)
    vqa_labels, vqa_scores = batch["vqa_labels"], batch["vqa_scores"]
    for i, (_label, _score) in enumerate(zip(vqa_labels, vqa_scores)):
        for lab, sc in zip(_label, _score):
            vqa_targets[i, lab] = sc
    vqa_loss = F.binary_cross_entropy_with_logits(vqa_logits.float(), vqa_targets) * vqa_targets.shape[1]
    ret = {"vqa_loss": vqa_loss, "vqa_logits": vqa_logits, "vqa_targets": vqa_targets, "vqa_labels": vqa_labels,
           "vqa_scores": vqa_scores}
    phase = "train" if pl_module.training else "val"
    loss = getattr(pl_module, f"{phase}_vqa_loss")(ret["vqa_loss"])
    score = getattr(pl_module, f"{phase}_vqa_score")(ret["vqa_logits"], ret["vqa_targets"])
    pl_module.log(f"vqa/{phase}/loss", loss)
    pl_module.log(f"vqa/{phase}/score", score)
    return ret


def vqa_test_step(pl_module, batch, output):
    """objectives.py:513-524: arg-max answer strings for a test batch (answer vocabulary from the datamodule)."""
    dicts = pl_module.trainer.datamodule.dm_dicts
    id2answer = (dicts["vqa_trainval"] if "vqa_trainval" in dicts else dicts["vqa"]).id2answer
    picks = output["vqa_logits"].argmax(dim=-1).tolist()
    return {"qids": batch["qid"], "preds": [id2answer[i] for i in picks]}


def vqa_test_wrapup(outs, model_name, out_dir="result"):
    """objectives.py:531-557: every rank writes its shard, rank 0 merges the shards into result/vqa_submit_<model>.json."""
    import glob
    import json
    import os
    import torch.distributed as dist
    multi = dist.is_available() and dist.is_initialized()
    rank = dist.get_rank() if multi else 0
    records = [{"question_id": q, "answer": a} for out in outs for q, a in zip(out["qids"], out["preds"])]
    shard = f"vqa_submit_{rank}.json"
    with open(shard, "w") as fp:
        json.dump(records, fp, indent=4)
    if multi:
        dist.barrier()
    if rank == 0:
        merged = []
        for path in sorted(glob.glob("vqa_submit_*.json")):
            with open(path) as fp:
                merged += json.load(fp)
        os.makedirs(out_dir, exist_ok=True)
        with open(os.path.join(out_dir, f"vqa_submit_{model_name}.json"), "w") as fp:
            json.dump(merged, fp, indent=4)
    if multi:
        dist.barrier()
    os.remove(shard)


def init_weights(module):
    if isinstance(module, (nn.Linear, nn.Embedding)):
        module.weight.data.normal_(mean=0.0, std=0.02)
    elif isinstance(module, nn.LayerNorm):
        module.bias.data.zero_()
        module.weight.data.fill_(1.0)
    if isinstance(module, nn.Linear) and module.bias is not None:
        module.bias.data.zero_()
