"""Frame input transform: the CPU oracle and the product's coefficient tables against fixtures recorded
from PIL itself (tools/gen_golden_frames.py) -- what torchvision's Resize / crop / flip in the reference's
base/transforms3D.py execute.  Runs without a GPU."""
import numpy as np

from helpers import golden
from oracle.frames import center_crop_offset, frames_transform, precompute_coeffs, resize_bilinear_u8


def _frames():
    rng = np.random.default_rng(2024)
    smooth = (np.add.outer(np.arange(256), np.arange(256)) % 256).astype(np.uint8)
    return np.stack([rng.integers(0, 256, (256, 256, 3), dtype=np.uint8),
                     np.stack([smooth, smooth.T, 255 - smooth], -1),
                     rng.integers(0, 256, (256, 256, 3), dtype=np.uint8) // 8 * 8])


def test_oracle_resize_is_bit_exact_vs_pil_fixture():
    g = golden("frames_transform.npz")
    frames = _frames()
    for f, ref in zip(frames, g["resized"]):
        assert np.array_equal(resize_bilinear_u8(f, 48), ref)


def test_oracle_crop_flip_normalise_vs_pil_fixture():
    g = golden("frames_transform.npz")
    frames = _frames()
    x1, y1 = [int(v) for v in g["crop"]]
    train = ((g["train_u8"].astype(np.float32) / 255.0 - 0.5) / 0.5).transpose(0, 3, 1, 2)
    assert np.abs(frames_transform(frames, 48, 40, x1, y1, True) - train).max() < 1e-6
    off = center_crop_offset(48, 40)
    assert off == 4
    center = ((g["center_u8"].astype(np.float32) / 255.0 - 0.5) / 0.5).transpose(0, 3, 1, 2)
    assert np.abs(frames_transform(frames) - center).max() < 1e-6


def test_product_tables_equal_the_oracle_tables():
    from feature_vs_text_compound_emotion_amd.frames import resample_tables
    for a, b in [(256, 48), (100, 48), (77, 48), (48, 48), (40, 48), (224, 112), (256, 40)]:
        b1, k1 = resample_tables(a, b)
        b2, k2 = precompute_coeffs(a, b)
        assert np.array_equal(b1, b2) and np.array_equal(k1, k2), (a, b)
        assert (k1.sum(1) - (1 << 22)).__abs__().max() <= k1.shape[1]  # rows sum to ~2^22


def test_draw_follows_the_reference_call_order():
    import random
    from feature_vs_text_compound_emotion_amd.frames import FrameTransform
    random.seed(7)
    exp = []
    for _ in range(3):
        x1 = random.randint(0, 8)
        y1 = random.randint(0, 8)
        exp.append((x1, y1, 1 if random.random() < 0.5 else 0))
    random.seed(7)
    got = FrameTransform(48, 40, train=True).draw(3)
    assert [tuple(int(v) for v in r) for r in got] == exp
    assert FrameTransform(48, 40, train=False).draw(2).tolist() == [[4, 4, 0], [4, 4, 0]]
