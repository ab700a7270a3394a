"""On-device input pipeline: oracle/image_ref.py against PIL and the PIL-produced fixtures (CPU), and the
HIP kernels (csrc/input.hip through the C ABI) against the oracle -- BIT-EXACT: this is integer / byte arithmetic plus three
correctly rounded fp32 operations per value."""
import numpy as np
import pytest
import torch

from oracle import image_ref as R
from oracle.gen_input_golden import CASES, synth_image


# ---------------------------------------------------------------------------------------------------- CPU: pin the oracle
@pytest.mark.parametrize("name", list(CASES))
def test_oracle_matches_pil_fixture(name, golden):
    gold = golden(name)
    H, W, S = (int(v) for v in gold["shape"])
    img = synth_image(H, W, seed=H * 1000 + W)
    assert np.array_equal(R.resize_bicubic_u8(img, S), gold["resized"])
    assert np.array_equal(R.albef_transform(img, S), gold["normalized"])


def test_oracle_matches_pil_live():
    """Random images at real sizes (incl. the 384 target, up- and down-scaling, an unchanged axis) against PIL itself."""
    Image = pytest.importorskip("PIL.Image")
    rng = np.random.default_rng(3)
    for H, W, S in [(480, 640, 384), (333, 500, 384), (384, 384, 384), (600, 384, 384), (90, 120, 384), (1, 7, 16), (700, 20, 224)]:
        img = rng.integers(0, 256, (H, W, 3), dtype=np.uint8)
        ref = np.asarray(Image.fromarray(img, "RGB").resize((S, S), Image.BICUBIC))
        assert np.array_equal(R.resize_bicubic_u8(img, S), ref), (H, W, S)


def test_mlm_mask_oracle_statistics():
    """transformers 4.6.0 mask_tokens: 15 % of non-special tokens selected; 80 % -> <mask>, 10 % random, 10 % unchanged;
    special tokens never touched; labels = original id at selected positions, -100 elsewhere."""
    rng = np.random.default_rng(0)
    ids = rng.integers(3, 50264, (64, 40))
    ids[:, 0] = 0
    ids[:, -5:] = 1
    ids[:, -6] = 2
    out, lab = R.mlm_mask(ids, seed=1234)
    special = ids <= 2
    sel = lab != -100
    assert not (sel & special).any() and np.array_equal(out[special], ids[special])
    assert np.array_equal(lab[sel], ids[sel]) and np.array_equal(out[~sel], ids[~sel])
    n = (~special).sum()
    assert abs(sel.sum() / n - 0.15) < 0.03
    masked = (out == 50264) & sel
    kept = (out == ids) & sel
    assert abs(masked.sum() / sel.sum() - 0.8) < 0.08 and abs(kept.sum() / sel.sum() - 0.1) < 0.06
    out2, lab2 = R.mlm_mask(ids, seed=1235)
    assert not np.array_equal(lab, lab2)                      # another seed, another mask
    assert np.array_equal(R.mlm_mask(ids, seed=1234)[0], out)  # pure function of (ids, seed)


# ---------------------------------------------------------------------------------------------------- GPU: kernels vs oracle
@pytest.mark.gpu
def test_device_transform_bit_exact():
    """One ragged batch (down-scale, up-scale, mixed, unchanged axis, a row-strided view) through csrc/input.hip: the resized
    uint8 image is not observable, the fp32 output must equal the oracle's bit for bit."""
    from fiber_amd import data, lib
    lib.load()
    rng = np.random.default_rng(5)
    shapes = [(97, 131), (20, 30), (150, 40), (100, 64), (64, 64), (480, 640)]
    S = 64
    imgs = [synth_image(H, W, seed=H * 1000 + W) if i < 4 else rng.integers(0, 256, (H, W, 3), dtype=np.uint8)
            for i, (H, W) in enumerate(shapes)]
    dev = [torch.from_numpy(im).cuda() for im in imgs]
    wide = torch.from_numpy(rng.integers(0, 256, (50, 90, 3), dtype=np.uint8)).cuda()
    dev.append(wide[:, 10:70])                                 # packed pixels, row stride > 3 * W
    imgs.append(wide[:, 10:70].cpu().numpy())
    out = data.DeviceImageTransform(S)(dev).cpu().numpy()
    for i, im in enumerate(imgs):
        assert np.array_equal(out[i], R.albef_transform(im, S)), (i, im.shape)


@pytest.mark.gpu
@pytest.mark.parametrize("name", list(CASES))
def test_device_transform_matches_pil_fixture(name, golden):
    from fiber_amd import data
    gold = golden(name)
    H, W, S = (int(v) for v in gold["shape"])
    img = torch.from_numpy(synth_image(H, W, seed=H * 1000 + W)).cuda()
    out = data.DeviceImageTransform(S)([img]).cpu().numpy()[0]
    assert np.array_equal(out, gold["normalized"])


@pytest.mark.gpu
def test_device_transform_full_size_properties():
    """BASELINE size (384^2 from camera-sized sources): bit-exact against the oracle on one image, and a size-independent
    property on a batch: a constant image stays constant (the coefficient windows sum to 1 after quantisation + rounding)."""
    from fiber_amd import data
    rng = np.random.default_rng(9)
    img = rng.integers(0, 256, (480, 640, 3), dtype=np.uint8)
    const = np.full((333, 500, 3), 77, np.uint8)
    out = data.DeviceImageTransform(384)([torch.from_numpy(img).cuda(), torch.from_numpy(const).cuda()]).cpu().numpy()
    assert np.array_equal(out[0], R.albef_transform(img, 384))
    want = (np.float32(77) / np.float32(255) - np.asarray(R.MEAN, np.float32)) / np.asarray(R.STD, np.float32)
    assert np.array_equal(out[1], np.broadcast_to(want[:, None, None], (3, 384, 384)))


@pytest.mark.gpu
def test_device_mlm_mask_and_collate():
    from fiber_amd import data
    rng = np.random.default_rng(1)
    ids = rng.integers(3, 50264, (32, 40))
    ids[:, 0] = 0
    ids[:, -3:] = 1
    ids[:, -4] = 2
    d_ids = torch.from_numpy(ids).cuda()
    out, lab = data.mlm_mask(d_ids, seed=0xDEADBEEF12345)
    ro, rl = R.mlm_mask(ids, seed=0xDEADBEEF12345)
    assert np.array_equal(out.cpu().numpy(), ro) and np.array_equal(lab.cpu().numpy(), rl)
    # collate: raw samples -> the batch schema of base_dataset.py:172-245
    samples = [{"image": torch.from_numpy(rng.integers(0, 256, (40 + 3 * i, 50, 3), dtype=np.uint8)).cuda(),
                "false_image_0": torch.from_numpy(rng.integers(0, 256, (30, 45 + i, 3), dtype=np.uint8)).cuda(),
                "text_ids": torch.tensor([0] + list(range(10, 15 + i)) + [2]).cuda()} for i in range(3)]
    b = data.device_collate(samples, data.DeviceImageTransform(32), seed=7, max_text_len=12, draw_false_image=1)
    assert b["image"][0].shape == (3, 3, 32, 32) and b["false_image_0"][0].shape == (3, 3, 32, 32)
    assert b["text_ids"].shape == (3, 12) and b["text_ids"][0, :7].tolist() == [0, 10, 11, 12, 13, 14, 2]
    assert b["text_masks"].sum(1).tolist() == [7, 8, 9] and (b["text_labels"] == -100).all()
    sel = b["text_labels_mlm"] != -100
    assert (b["text_ids"][sel] == b["text_labels_mlm"][sel]).all() and (b["text_ids_mlm"][~sel] == b["text_ids"][~sel]).all()
