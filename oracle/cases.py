"""Shared parity-case definitions (inputs only).  TEST INFRASTRUCTURE ONLY.

Each case names a module configuration and deterministic inputs (oracle/detgen.py).  The same
definitions drive (a) oracle/gen_golden.py, which runs the REFERENCE modules in the build
container and stores outputs under tests/golden/, (b) the CPU tests that pin oracle/fiber_ref.py
to those outputs, and (c) the GPU tests that compare the HIP path with the oracle.
"""
import numpy as np
import torch

from . import detgen

BLOCK_CASES = {
    # name: dim, res, heads, ws, shift, B, dim_text, S
    "blk_tiny": dict(dim=64, res=(8, 8), heads=2, ws=4, shift=0, B=2, dim_text=None, S=0),
    "blk_tiny_shift": dict(dim=64, res=(8, 8), heads=2, ws=4, shift=2, B=2, dim_text=None, S=0),
    "blk_tiny_shift_text": dict(dim=64, res=(8, 8), heads=2, ws=4, shift=2, B=2, dim_text=96, S=6),
    "blk_w7_shift": dict(dim=96, res=(14, 14), heads=3, ws=7, shift=3, B=2, dim_text=None, S=0),
    "blk_s0_shift": dict(dim=128, res=(96, 96), heads=4, ws=12, shift=6, B=1, dim_text=None, S=0),
    "blk_s2_shift_text": dict(dim=512, res=(24, 24), heads=16, ws=12, shift=6, B=2, dim_text=768, S=40),
    "blk_s3_text": dict(dim=1024, res=(12, 12), heads=32, ws=12, shift=6, B=2, dim_text=768, S=40),
}

MERGE_CASES = {
    "merge_tiny": dict(dim=32, res=(8, 8), B=2),
    "merge_s2": dict(dim=512, res=(24, 24), B=1),
}

EMBED_CASES = {
    "pe_tiny": dict(img=32, dim=32, B=2),
    "pe_384": dict(img=384, dim=128, B=1),
}

ROBERTA_LAYER_CASES = {
    # name: layer_index, image tokens L / dim (None = plain), last_norm
    "rl_plain": dict(layer_index=0, L=0, dim_img_tok=None, last_norm=True),
    "rl_fused512": dict(layer_index=6, L=576, dim_img_tok=512, last_norm=True),
    "rl_fused1024": dict(layer_index=10, L=144, dim_img_tok=1024, last_norm=True),
    "rl_fused1024_nonorm": dict(layer_index=11, L=144, dim_img_tok=1024, last_norm=False),
}

# Full fused-path configurations (config dict overrides for FiberRef / FIBERTransformerSS)
TINY = dict(
    image_size=96, swin_arch=(32, (2, 2, 18, 2), (1, 2, 4, 8)), swin_dim_text=128, vocab_size=1000,
    hidden_size=64, num_heads=2, num_layers=12, mlp_ratio=4, input_image_embed_size=256,
    input_text_embed_size=128, max_text_len=12, max_position_embeddings=40, text_dropout=0.0, drop_path_rate=0.0,
)
SWIN_T = dict(
    image_size=224, vit="swin_tiny_patch4_window7_224", input_image_embed_size=768, max_text_len=40,
    text_dropout=0.0, drop_path_rate=0.0,
)
SWIN_B = dict(image_size=384, vit="swin_base_patch4_window12_384_in22k", text_dropout=0.0, drop_path_rate=0.0)

PATH_CASES = {
    "path_tiny": dict(config=TINY, B=2, grads=True),
    "path_swin_t": dict(config=SWIN_T, B=2, grads=True),
    "path_swin_b": dict(config=SWIN_B, B=1, grads=True),
}


# VQAv2 fine-tune head at 576^2 (BASELINE.json configs[3]; reference config.py:134-150): 18x18 windows (N = 324), 50 text
# tokens, BCE over the answer vocabulary.  `vqa_tiny576` keeps the 576^2 geometry with shrunken widths so it runs anywhere.
VQA_CASES = {
    "vqa_tiny576": dict(config=dict(TINY, image_size=576, max_text_len=20, vqav2_label_size=97,
                                    loss_names={"vqa": 1}), B=2),
    "vqa_swin_b_576": dict(config=dict(SWIN_B, image_size=576, max_text_len=50, loss_names={"vqa": 1}), B=1),
}
# task_pretrain_mlm_itm_itc (reference config.py:95-110): MLM + ITC against feature queues + ITM on
# hard negatives drawn from batch + raw-input queues.  Two consecutive training steps on different batches, so the second
# step sees a non-empty queue, a wrapped queue pointer (queue_size 6 < 2 * B) and queue-drawn negatives.
ITC_CASES = {
    "itc_tiny": dict(config=dict(TINY, loss_names={"mlm": 1, "itm": 1, "itc": 1}, itc_queue_size=6, draw_false_image=0), B=4),
}
ADAPT_CASE = dict(before=384, after=576, heads=4)       # swin_adapt_position_encoding: 23^2 -> 35^2 rows


def randn(name, shape, seed=0, std=1.0):
    g = detgen._rng("input:" + name, seed)
    return torch.from_numpy((g.standard_normal(shape) * std).astype(np.float32))


def text_mask(B, S, name, seed=0):
    """int64 (B,S) padding mask with ragged lengths (first row full) and its additive (B,1,1,S) form."""
    g = detgen._rng("mask:" + name, seed)
    lens = g.integers(max(1, S // 3), S + 1, size=B)
    lens[0] = S
    m = (np.arange(S)[None, :] < lens[:, None]).astype(np.int64)
    m = torch.from_numpy(m)
    return m, (1.0 - m[:, None, None, :].float()) * -10000.0


def block_inputs(name):
    c = BLOCK_CASES[name]
    L = c["res"][0] * c["res"][1]
    x = randn(name + ".x", (c["B"], L, c["dim"]))
    g = randn(name + ".g", (c["B"], L, c["dim"]))
    y = ext = None
    if c["dim_text"]:
        y = randn(name + ".y", (c["B"], c["S"], c["dim_text"]))
        _, ext = text_mask(c["B"], c["S"], name)
    return x, y, ext, g


def roberta_layer_inputs(name, B=2, S=40, hidden=768):
    c = ROBERTA_LAYER_CASES[name]
    h = randn(name + ".h", (B, S, hidden))
    g = randn(name + ".g", (B, S, hidden))
    _, ext = text_mask(B, S, name)
    img = randn(name + ".img", (B, c["L"], c["dim_img_tok"])) if c["dim_img_tok"] else None
    return h, ext, img, g


def summarize(t, cap=4096):
    """Compact fingerprint of a tensor: strided sample + l2 norm + sum (keeps fixtures small)."""
    t = t.detach().float().cpu().reshape(-1)
    n = t.numel()
    stride = max(1, n // cap)
    return {"sub": t[::stride][:cap].numpy().copy(), "norm": np.float64(t.double().norm().item()),
            "sum": np.float64(t.double().sum().item()), "numel": np.int64(n)}


def flatten_summary(prefix, t, out):
    for k, v in summarize(t).items():
        out[f"{prefix}/{k}"] = v


def check_summary(prefix, t, gold, rtol, atol, norm_rtol=None):
    """Assert tensor `t` matches the stored fingerprint `gold[prefix/...]`; returns max abs err."""
    s = summarize(t)
    assert int(gold[f"{prefix}/numel"]) == int(s["numel"]), f"{prefix}: numel {s['numel']} vs {gold[prefix + '/numel']}"
    ref = gold[f"{prefix}/sub"]
    err = np.abs(s["sub"] - ref)
    tol = atol + rtol * np.abs(ref)
    assert (err <= tol).all(), f"{prefix}: max err {err.max():.3e} (tol {tol.min():.1e}), worst ref {ref[err.argmax()]:.4e}"
    gn = float(gold[f"{prefix}/norm"])
    nr = norm_rtol if norm_rtol is not None else 10 * rtol
    assert abs(float(s["norm"]) - gn) <= nr * gn + atol, f"{prefix}: norm {s['norm']:.6e} vs {gn:.6e}"
    return float(err.max())
