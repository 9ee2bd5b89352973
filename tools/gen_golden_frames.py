"""Golden fixture for the frame input transform, from PIL itself (what torchvision's Resize /
CenterCrop / crop / FLIP_LEFT_RIGHT in the reference's base/transforms3D.py execute).

    python tools/gen_golden_frames.py
"""
import os
import sys

import numpy as np
from PIL import Image

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle.frames import frames_transform, resize_bilinear_u8  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")


def main():
    rng = np.random.default_rng(2024)
    smooth = (np.add.outer(np.arange(256), np.arange(256)) % 256).astype(np.uint8)
    frames = np.stack([rng.integers(0, 256, (256, 256, 3), dtype=np.uint8),
                       np.stack([smooth, smooth.T, 255 - smooth], -1),
                       rng.integers(0, 256, (256, 256, 3), dtype=np.uint8) // 8 * 8])
    resized = []
    for f in frames:
        pil = Image.fromarray(f).convert("RGB").resize((48, 48), Image.BILINEAR)
        ref = np.asarray(pil)
        mine = resize_bilinear_u8(f, 48)
        assert np.array_equal(ref, mine), np.abs(ref.astype(int) - mine.astype(int)).max()
        resized.append(ref)
    # non-square / other sizes exercise the coefficient tables
    odd = rng.integers(0, 256, (100, 77, 3), dtype=np.uint8)
    ref = np.asarray(Image.fromarray(odd).resize((48, 48), Image.BILINEAR))
    assert np.array_equal(ref, resize_bilinear_u8(odd, 48))
    # train-style crop + flip through PIL ops
    x1, y1 = 3, 7
    pil = [Image.fromarray(f).convert("RGB").resize((48, 48), Image.BILINEAR).crop((x1, y1, x1 + 40, y1 + 40))
           .transpose(Image.FLIP_LEFT_RIGHT) for f in frames]
    train = np.stack([np.asarray(p) for p in pil]).astype(np.float32) / 255.0
    train = ((train - 0.5) / 0.5).transpose(0, 3, 1, 2)
    assert np.abs(frames_transform(frames, 48, 40, x1, y1, True) - train).max() < 1e-6
    center = np.stack(resized)[:, 4:44, 4:44]
    np.savez_compressed(os.path.join(OUT, "frames_transform.npz"), seed=np.array([2024]), resized=np.stack(resized),
                        train_u8=np.stack([np.asarray(p) for p in pil]), center_u8=center, crop=np.array([x1, y1]))
    print("frames_transform.npz", os.path.getsize(os.path.join(OUT, "frames_transform.npz")))


if __name__ == "__main__":
    main()
