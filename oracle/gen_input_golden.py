"""Golden vectors for the on-device input pipeline, produced by the REAL third-party code the reference calls:
PIL `Image.resize((S, S), Image.BICUBIC)` (what torchvision's Resize does on a PIL image, transforms/transform.py:12) followed
by ToTensor / Normalize restated in fp32 (torchvision is not installed here).  Inputs are small synthetic uint8 images with
structure (gradients + noise + saturated patches) regenerated from a seed by the tests; the fixtures hold PIL's resized uint8
image and the normalised fp32 tensor.  Run in the build container: python -m oracle.gen_input_golden"""
import os

import numpy as np
from PIL import Image

from . import image_ref

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
CASES = {"resize_down": (97, 131, 48), "resize_up": (20, 30, 64), "resize_mixed": (150, 40, 64), "resize_same_w": (100, 64, 64)}


def synth_image(H, W, seed):
    g = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:H, 0:W]
    img = np.stack([(xx * 255 // max(1, W - 1)), (yy * 255 // max(1, H - 1)), ((xx + yy) % 256)], -1).astype(np.int64)
    img = img + g.integers(-40, 41, (H, W, 3))
    img[: H // 5, : W // 4] = 255                      # saturated patches exercise the clip after negative bicubic lobes
    img[-(H // 6):, -(W // 3):] = 0
    return np.clip(img, 0, 255).astype(np.uint8)


def main():
    for name, (H, W, S) in CASES.items():
        img = synth_image(H, W, seed=H * 1000 + W)
        r = np.asarray(Image.fromarray(img, "RGB").resize((S, S), Image.BICUBIC))
        t = np.transpose(r.astype(np.float32) / np.float32(255.0), (2, 0, 1))
        t = (t - np.asarray(image_ref.MEAN, np.float32)[:, None, None]) / np.asarray(image_ref.STD, np.float32)[:, None, None]
        np.savez_compressed(os.path.join(OUT, name + ".npz"), resized=r, normalized=t.astype(np.float32),
                            shape=np.array([H, W, S]))
        print("wrote", name, r.shape)


if __name__ == "__main__":
    main()
