"""Deterministic, platform-independent weight / batch generator.  TEST INFRASTRUCTURE ONLY.

Every tensor is drawn from a counter-based Philox stream keyed by the CRC32 of its
state-dict name, so the build container (golden generation from the reference) and the
GPU box (parity tests) regenerate bit-identical fp32 weights without shipping them.
"""
import zlib

import numpy as np
import torch


def _rng(name, seed):
    return np.random.Generator(np.random.Philox(key=[zlib.crc32(name.encode()), seed]))


def tensor_for(name, shape, seed=0, alpha=0.5):
    """fp32 value for parameter `name` (statistics chosen so every branch carries signal)."""
    g = _rng(name, seed)
    n = lambda std: torch.from_numpy((g.standard_normal(shape) * std).astype(np.float32))
    leaf = name.rsplit(".", 1)[-1]
    if "alpha_" in name:
        return torch.full(shape, alpha, dtype=torch.float32)
    if leaf == "temp":                                   # ITC temperature keeps its initial value (fiber_module.py:58)
        return torch.full(shape, 0.07, dtype=torch.float32)
    if "relative_position_bias_table" in name:
        return n(0.5)
    is_norm = ("norm" in name.lower()) and leaf in ("weight", "bias")
    if is_norm:
        return 1.0 + n(0.1) if leaf == "weight" else n(0.05)
    if leaf == "bias" or name.endswith("mlm_score.bias"):
        return n(0.02)
    return n(0.02)


def fill_(module, seed=0, alpha=0.5):
    """In-place deterministic fill of every floating parameter of `module`; returns module."""
    with torch.no_grad():
        for name, p in module.named_parameters():
            p.copy_(tensor_for(name, tuple(p.shape), seed, alpha))
    return module


def state_dict_for(module, seed=0, alpha=0.5):
    return {k: tensor_for(k, tuple(v.shape), seed, alpha) for k, v in module.named_parameters()}


def synth_batch(B, image_size=384, max_text_len=40, vocab=50265, seed=0, min_len=8):
    """Synthetic image-text batch per SURVEY.md section 8(d) (schema: base_dataset.py:172-245)."""
    g = _rng("batch", seed)
    img = torch.from_numpy(g.standard_normal((B, 3, image_size, image_size)).astype(np.float32))
    fimg = torch.from_numpy(g.standard_normal((B, 3, image_size, image_size)).astype(np.float32))
    ids = np.full((B, max_text_len), 1, dtype=np.int64)
    lens = g.integers(min(min_len, max_text_len), max_text_len + 1, size=B)
    for b, L in enumerate(lens):
        ids[b, 0] = 0
        ids[b, 1:L - 1] = g.integers(3, vocab - 2, size=L - 2)
        ids[b, L - 1] = 2
    masks = (ids != 1).astype(np.int64)
    special = (ids == 0) | (ids == 1) | (ids == 2)
    pick = (g.random((B, max_text_len)) < 0.15) & ~special
    for b in range(B):                                   # at least one masked position per row
        if not pick[b].any():
            pick[b, 1] = True
    labels_mlm = np.where(pick, ids, -100)
    r = g.random((B, max_text_len))
    ids_mlm = ids.copy()
    ids_mlm[pick & (r < 0.8)] = vocab - 1                # <mask> = 50264
    rnd = pick & (r >= 0.8) & (r < 0.9)
    ids_mlm[rnd] = g.integers(3, vocab - 2, size=int(rnd.sum()))
    itm = np.concatenate([np.ones(B // 2), np.zeros(B - B // 2)])
    itm = itm[g.permutation(B)].astype(np.float32)
    t = torch.from_numpy
    return {
        "image": [img], "false_image_0": [fimg], "text": ["x"] * B,
        "text_ids": t(ids), "text_masks": t(masks), "text_labels": torch.full((B, max_text_len), -100, dtype=torch.long),
        "text_ids_mlm": t(ids_mlm), "text_labels_mlm": t(labels_mlm), "itm_labels": t(itm),
    }


def synth_vqa(B, n_answers=3129, seed=0):
    """Ragged soft VQA targets (schema: vqav2_dataset.py `vqa_labels` / `vqa_scores`): per question 0..4 distinct answer
    ids with scores from the VQAv2 soft-accuracy set; question 1 (when present) has no answer at all."""
    g = _rng("vqa", seed)
    labels, scores = [], []
    for b in range(B):
        k = 0 if b == 1 else int(g.integers(1, 5))
        labels.append([int(v) for v in g.choice(n_answers, size=k, replace=False)])
        scores.append([float(v) for v in g.choice(np.array([0.3, 0.6, 0.9, 1.0]), size=k)])
    return {"vqa_labels": labels, "vqa_scores": scores}
