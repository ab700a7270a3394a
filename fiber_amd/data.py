"""On-device input pipeline: the reference's per-sample CPU transforms, executed on the GPU after
the H2D copy of the RAW sample bytes.

  * `DeviceImageTransform` = `albef_transform(size)` of coarse_grained/fiber/transforms/transform.py:10-17 (PIL bicubic Resize ->
    ToTensor -> Normalize), bit-identical to PIL + torchvision on uint8 RGB input (csrc/input.hip restates Pillow's resampler);
  * `mlm_mask` = the masking rule of `DataCollatorForLanguageModeling(mlm_probability)` used by BaseDataModule
    (datamodules/datamodule_base.py:52) -> `text_ids_mlm`, `text_labels_mlm` of the batch schema (base_dataset.py:223-243).

What stays on the host: JPEG decode (`Image.open(...).convert("RGB")`, base_dataset.py:97-103) and tokenisation -- the loader
workers hand over uint8 [H, W, 3] arrays and int64 token ids; everything after that is stream-ordered device work, so at
~800 images/s per GPU the loader no longer resizes and normalises 2 x 384^2 fp32 images per sample on CPU cores.
"""
import numpy as np
import torch

from . import lib, ops

IMAGENET_MEAN = (0.485, 0.456, 0.406)
IMAGENET_STD = (0.229, 0.224, 0.225)

_DESC = np.dtype([("src", "<i8"), ("H", "<i4"), ("W", "<i4"), ("src_stride", "<i4"), ("ksize_h", "<i4"), ("ksize_v", "<i4"),
                  ("tmp_off", "<i4"), ("coef_h_off", "<i4"), ("coef_v_off", "<i4")])
assert _DESC.itemsize == 40


class DeviceImageTransform:
    """images: list of uint8 [H, W, 3] tensors on a HIP device (ragged sizes) -> fp32 [B, 3, size, size]."""

    def __init__(self, size=384, mean=IMAGENET_MEAN, std=IMAGENET_STD):
        self.size = int(size)
        self.mean = (lib.C.c_float * 3)(*mean)
        self.std = (lib.C.c_float * 3)(*std)

    def __call__(self, images):
        S, n = self.size, len(images)
        dev = images[0].device
        desc = np.zeros(n, _DESC)
        tmp_off = coef_off = 0
        for i, im in enumerate(images):
            if im.dtype != torch.uint8 or im.dim() != 3 or im.shape[2] != 3 or im.stride(2) != 1 or im.stride(1) != 3:
                raise ValueError("DeviceImageTransform needs uint8 [H, W, 3] images with packed pixels")
            H, W = int(im.shape[0]), int(im.shape[1])
            kh, kv = lib.plain("fiber_resample_ksize", W, S), lib.plain("fiber_resample_ksize", H, S)
            desc[i] = (lib.ptr(im), H, W, int(im.stride(0)), kh, kv, tmp_off, coef_off, coef_off + S * (2 + kh))
            tmp_off += (H * S * 3 + 15) // 16 * 16
            coef_off += S * (2 + kh) + S * (2 + kv)
        if tmp_off >= 2 ** 31:
            raise ValueError("batch too large for one call (intermediate images exceed 2 GiB): split it")
        d_desc = torch.from_numpy(desc.view(np.uint8).copy()).to(dev, non_blocking=True)
        coef = torch.empty(coef_off, dtype=torch.int32, device=dev)
        tmp = torch.empty(tmp_off, dtype=torch.uint8, device=dev)
        out = torch.empty((n, 3, S, S), dtype=torch.float32, device=dev)
        lib.call("fiber_resize_bicubic_norm_u8", lib.ptr(d_desc), n, lib.ptr(coef), lib.ptr(tmp), lib.ptr(out), S,
                 int(desc["H"].max()), self.mean, self.std)
        self._keep = (images, d_desc)          # the sources / table must outlive the (asynchronous) kernels of this call
        return out


def mlm_mask(ids, seed=None, mlm_probability=0.15, mask_id=50264, vocab=50265, special=(0, 2)):
    """ids int64 [B, S] (device) -> (text_ids_mlm, text_labels_mlm).  `special`: inclusive id range never masked (RoBERTa
    <s> = 0, <pad> = 1, </s> = 2).  `seed`: 64-bit key of this batch (default: the next key of the input pipeline's own stream,
    ops.collate_seed(): by value and advancing per batch whether or not the training step is a captured hipGraph)."""
    ids = ids.contiguous()
    out, lab = torch.empty_like(ids), torch.empty_like(ids)
    if seed is None:
        seed = ops.collate_seed()
    lib.call("fiber_mlm_mask_i64", lib.ptr(ids), lib.ptr(out), lib.ptr(lab), ids.numel(), int(seed) & (2 ** 64 - 1),
             int(mlm_probability * 2 ** 32), int(mask_id), int(vocab), int(special[0]), int(special[1]))
    return out, lab


def device_collate(samples, transform, seed=None, max_text_len=40, pad_id=1, draw_false_image=0):
    """The device-side half of BaseDataset.collate (base_dataset.py:172-245) for RAW samples: each sample is a dict with
    `image` (uint8 [H, W, 3] device tensor), optional `false_image_0`, and `text_ids` (1-D int64 device tensor, already
    tokenised with <s> ... </s>).  Returns the batch dict FIBERTransformerSS.forward consumes."""
    dev = samples[0]["image"].device
    B = len(samples)
    ids = torch.full((B, max_text_len), pad_id, dtype=torch.int64, device=dev)
    for i, s in enumerate(samples):
        t = s["text_ids"][:max_text_len]
        ids[i, :t.numel()] = t
    ids_mlm, labels_mlm = mlm_mask(ids, seed)
    batch = {"image": [transform([s["image"] for s in samples])], "text": ["" for _ in samples], "text_ids": ids,
             "text_masks": (ids != pad_id).long(), "text_labels": torch.full_like(ids, -100), "text_ids_mlm": ids_mlm,
             "text_labels_mlm": labels_mlm}
    for k in range(draw_false_image):
        batch[f"false_image_{k}"] = [transform([s[f"false_image_{k}"] for s in samples])]
    return batch
