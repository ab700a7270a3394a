"""Checkpoint adaptation for a change of input resolution (reference coarse_grained/fiber/modules/swin_helpers.py:20-44).

A FIBER checkpoint pre-trained at 384^2 has a (2*12-1)^2 x heads relative-position-bias table per Swin block; fine-tuning
at 576^2 uses 18x18 windows and so needs (2*18-1)^2 rows.  The reference resizes every table bicubically and drops the
resolution-dependent buffers (`attn_mask`, `relative_position_index`) so the freshly built ones are kept."""
import torch
import torch.nn.functional as F

_STALE_BUFFERS = ("attn_mask", "relative_position_index")


def swin_adapt_position_encoding(model, before=384, patch_size=32, after=384, suffix="relative_position_bias_table"):
    """`model` is a state dict; returns it with every `*relative_position_bias_table` resized for the new window.
    Window side = resolution // 32 whatever `patch_size` says (kept in the signature for call compatibility)."""
    if before == after:
        return model
    side0, side1 = 2 * (before // 32) - 1, 2 * (after // 32) - 1
    tables = [k for k in model if k.endswith(suffix)]
    if not tables:
        raise AssertionError(f"no '*{suffix}' entries in the state dict")
    for k in tables:
        t = model[k]                                            # [side0^2, heads]
        heads = t.shape[1]
        grid = t.t().reshape(1, heads, side0, side0)
        grid = F.interpolate(grid, size=(side1, side1), mode="bicubic")
        model[k] = grid[0].reshape(heads, side1 * side1).t().contiguous()
    for k in [k for k in model if k.endswith(_STALE_BUFFERS)]:
        del model[k]
    return model
