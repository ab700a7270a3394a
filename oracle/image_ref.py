"""TEST INFRASTRUCTURE ONLY (never imported by fiber_amd/): CPU restatement of the reference's input transforms for the
on-device input pipeline.

1. `albef_transform` (reference coarse_grained/fiber/transforms/transform.py:10-17) = torchvision Resize((S,S), BICUBIC) on a
   PIL image + ToTensor + Normalize.  The arithmetic lives in third-party Pillow (src/libImaging/Resample.c, functions
   precompute_coeffs / normalize_coeffs_8bpc / ImagingResampleHorizontal_8bpc / ImagingResampleVertical_8bpc, bicubic_filter
   with a = -0.5; unchanged across Pillow 3.x-12.x; the reference pins only `Pillow` transitively through torchvision).  It
   is restated below in numpy (float64 coefficient windows, 22-bit fixed point, 8-bit rounding after each pass) and PINNED
   against PIL itself in tests/test_input_pipeline.py (bit-exact on random and real-size cases) plus the committed fixtures
   tests/golden/resize_*.npz that PIL produced (oracle/gen_input_golden.py).
2. `mask_tokens` of transformers 4.6.0 DataCollatorForLanguageModeling (reference datamodule_base.py:52): the selection
   rule (15 % of non-special tokens; 80 / 10 / 10) restated on the counter-based hash the HIP kernel uses, so the GPU result
   is checked bit-exactly and the distribution against the collator's specification.
"""
import numpy as np

PRECISION_BITS = 32 - 8 - 2
MEAN = (0.485, 0.456, 0.406)
STD = (0.229, 0.224, 0.225)


def _bicubic(x):
    a = -0.5
    x = abs(x)
    if x < 1.0:
        return ((a + 2.0) * x - (a + 3.0)) * x * x + 1
    if x < 2.0:
        return (((x - 5) * x + 8) * x - 4) * a
    return 0.0


def precompute_coeffs(in_size, out_size):
    """Pillow precompute_coeffs + normalize_coeffs_8bpc: (bounds [out,2] int, coefficients [out,ksize] int32)."""
    scale = in_size / out_size
    filterscale = max(scale, 1.0)
    support = 2.0 * filterscale
    ksize = int(np.ceil(support)) * 2 + 1
    bounds = np.zeros((out_size, 2), np.int64)
    kk = np.zeros((out_size, ksize), np.int64)
    ss = 1.0 / filterscale
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        xmin = max(int(center - support + 0.5), 0)          # C cast: truncation toward zero (argument may be negative)
        xmax = min(int(center + support + 0.5), in_size) - xmin
        w = [_bicubic((x + xmin - center + 0.5) * ss) for x in range(xmax)]
        ww = 0.0
        for v in w:
            ww += v
        for x, v in enumerate(w):
            if ww != 0.0:
                v = v / ww
            kk[xx, x] = int(-0.5 + v * (1 << PRECISION_BITS)) if v < 0 else int(0.5 + v * (1 << PRECISION_BITS))
        bounds[xx] = (xmin, xmax)
    return bounds, kk


def _pass(img, out_size, axis):
    """One resampling pass along `axis` (0 = vertical, 1 = horizontal) of a uint8 [H, W, C] image."""
    in_size = img.shape[axis]
    if in_size == out_size:
        return img                                          # ImagingResample: need_horizontal / need_vertical false
    bounds, kk = precompute_coeffs(in_size, out_size)
    src = np.moveaxis(img, axis, 0).astype(np.int64)        # [in, other, C]
    out = np.empty((out_size,) + src.shape[1:], np.uint8)
    for xx in range(out_size):
        xmin, xmax = bounds[xx]
        acc = (1 << (PRECISION_BITS - 1)) + np.tensordot(kk[xx, :xmax], src[xmin:xmin + xmax], axes=(0, 0))
        out[xx] = np.clip(acc >> PRECISION_BITS, 0, 255).astype(np.uint8)
    return np.moveaxis(out, 0, axis)


def resize_bicubic_u8(img, size):
    """PIL `Image.resize((size, size), Image.BICUBIC)` of a uint8 [H, W, 3] array: horizontal pass, then vertical."""
    return _pass(_pass(np.ascontiguousarray(img), size, 1), size, 0)


def albef_transform(img, size, mean=MEAN, std=STD):
    """uint8 [H, W, 3] -> float32 [3, size, size]: Resize(BICUBIC) -> ToTensor (x / 255) -> Normalize, each op rounded to fp32."""
    r = resize_bicubic_u8(img, size).astype(np.float32)
    t = np.transpose(r / np.float32(255.0), (2, 0, 1))
    m = np.asarray(mean, np.float32)[:, None, None]
    s = np.asarray(std, np.float32)[:, None, None]
    return ((t - m) / s).astype(np.float32)


# ---- MLM masking -------------------------------------------------------------------------------------------------------
_M64 = (1 << 64) - 1


def hash_u32(seed, idx):
    """common.h hash_u32 (splitmix64 finaliser of seed + idx * golden ratio, high 32 bits)."""
    z = (seed + idx * 0x9E3779B97F4A7C15) & _M64
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & _M64
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & _M64
    z = z ^ (z >> 31)
    return z >> 32


def mlm_mask(ids, seed, mlm_probability=0.15, mask_id=50264, vocab=50265, special=(0, 2)):
    """(ids_mlm, labels): transformers 4.6.0 mask_tokens rule on the kernel's draws (special = inclusive id range)."""
    flat = np.asarray(ids, np.int64).reshape(-1)
    out, lab = flat.copy(), np.full_like(flat, -100)
    p_sel = int(mlm_probability * (1 << 32))
    for i, tok in enumerate(flat.tolist()):
        if special[0] <= tok <= special[1] or hash_u32(seed, 4 * i) >= p_sel:
            continue
        lab[i] = tok
        if hash_u32(seed, 4 * i + 1) < 3435973836:
            out[i] = mask_id
        elif hash_u32(seed, 4 * i + 2) < 2147483648:
            out[i] = hash_u32(seed, 4 * i + 3) % vocab
    return out.reshape(np.shape(ids)), lab.reshape(np.shape(ids))
