"""HIP frame input transform vs the PIL fixtures (bit exact on the uint8 stage) and the CPU oracle."""
import numpy as np
import pytest
import torch

from helpers import golden
from oracle.frames import frames_transform, resize_bilinear_u8

pytestmark = pytest.mark.gpu


def _frames():
    rng = np.random.default_rng(2024)
    smooth = (np.add.outer(np.arange(256), np.arange(256)) % 256).astype(np.uint8)
    return np.stack([rng.integers(0, 256, (256, 256, 3), dtype=np.uint8),
                     np.stack([smooth, smooth.T, 255 - smooth], -1),
                     rng.integers(0, 256, (256, 256, 3), dtype=np.uint8) // 8 * 8])


def test_matches_pil_fixture_bit_exactly():
    from feature_vs_text_compound_emotion_amd.frames import FrameTransform
    g = golden("frames_transform.npz")
    frames = torch.from_numpy(_frames()).cuda()[None]
    x1, y1 = [int(v) for v in g["crop"]]
    out, u8 = FrameTransform(48, 40, train=True)(frames, crop_xyf=[[x1, y1, 1]], return_u8=True)
    assert np.array_equal(u8[0].cpu().numpy(), g["train_u8"])
    ref = ((g["train_u8"].astype(np.float32) / np.float32(255.0) - np.float32(0.5)) / np.float32(0.5)).transpose(0, 3, 1, 2)
    assert np.array_equal(out[0].cpu().numpy(), ref)  # the float stage is the same three IEEE operations
    out, u8 = FrameTransform(48, 40, train=False)(frames, return_u8=True)
    assert np.array_equal(u8[0].cpu().numpy(), g["center_u8"])


@pytest.mark.parametrize("b,l,h,w,size,crop", [(2, 5, 256, 256, 48, 40), (3, 2, 100, 77, 48, 40), (1, 3, 40, 52, 48, 48),
                                               (2, 2, 224, 224, 112, 100)])
def test_matches_oracle_on_random_clips(b, l, h, w, size, crop):
    from feature_vs_text_compound_emotion_amd.frames import FrameTransform
    rng = np.random.default_rng(b * 1000 + h)
    frames = rng.integers(0, 256, (b, l, h, w, 3), dtype=np.uint8)
    cx = np.stack([rng.integers(0, size - crop + 1, b), rng.integers(0, size - crop + 1, b), rng.integers(0, 2, b)], 1)
    out = FrameTransform(size, crop, train=True)(torch.from_numpy(frames).cuda(), crop_xyf=cx).cpu().numpy()
    for i in range(b):
        ref = frames_transform(frames[i], size, crop, int(cx[i, 0]), int(cx[i, 1]), bool(cx[i, 2]))
        assert np.array_equal(out[i], ref), (i, np.abs(out[i] - ref).max())


def test_full_size_batch_properties():
    """B=32 clips x 32 frames of 256x256 (the on-disk trial format): flip(flip) == identity on the crop window,
    a constant image stays constant, outputs stay in [-1, 1]."""
    from feature_vs_text_compound_emotion_amd.frames import FrameTransform
    g = torch.Generator().manual_seed(3)
    frames = torch.randint(0, 256, (32, 32, 256, 256, 3), dtype=torch.uint8, generator=g).cuda()
    frames[5] = 77
    ft = FrameTransform(48, 40, train=True)
    cx = np.tile(np.array([[2, 6, 0]], np.int32), (32, 1))
    a = ft(frames, crop_xyf=cx)
    cx[:, 2] = 1
    bflip = ft(frames, crop_xyf=cx)
    assert torch.equal(a, bflip.flip(-1))
    assert a.abs().max().item() <= 1.0
    assert torch.equal(a[5], torch.full_like(a[5], (np.float32(77) / np.float32(255) - np.float32(0.5)) / np.float32(0.5)))
    assert a.shape == (32, 32, 3, 40, 40)


def test_cpu_tensor_is_rejected():
    from feature_vs_text_compound_emotion_amd.frames import FrameTransform
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        FrameTransform()(torch.zeros(1, 1, 8, 8, 3, dtype=torch.uint8))


def test_trial_dataset_batch_through_the_gpu_transform(tmp_path):
    """.npy trials -> TrialDataset (raw uint8 video) -> collate_to_device -> FrameTransform == the CPU oracle."""
    from feature_vs_text_compound_emotion_amd.frames import FrameTransform
    from feature_vs_text_compound_emotion_amd.trial_dataset import (TrialDataset, calculate_mean_std, collate_to_device,
                                                                      windowed_trial_list)
    rng = np.random.default_rng(0)
    trials = []
    for name, n in (("a", 9), ("b", 3)):
        d = tmp_path / name
        d.mkdir()
        np.save(d / "video.npy", rng.integers(0, 256, (n, 64, 64, 3), dtype=np.uint8))
        np.save(d / "vggish.npy", rng.normal(size=(n, 128)).astype(np.float32))
        np.save(d / "EXPR_continuous_label.npy", rng.integers(0, 7, n))
        trials.append([str(d), name, n])
    mods = ["video", "vggish", "EXPR_continuous_label"]
    data = windowed_trial_list(trials, 4, 3)
    ds = TrialDataset(data, mods, {m: 1 for m in mods}, {"video": (64, 64, 3), "vggish": (128,), "EXPR_continuous_label": (1,)},
                      4, "train", mean_std=calculate_mean_std(data, ("vggish",)))
    batch = [ds[i] for i in range(len(ds))]
    ft = FrameTransform(48, 40, train=True)
    cx = ft.draw(len(batch))
    x, names, lengths, index = collate_to_device(batch, "cuda", ft, crop_xyf=cx)
    assert tuple(x["video"].shape) == (len(batch), 4, 3, 40, 40) and tuple(x["vggish"].shape) == (len(batch), 1, 4, 128)
    assert tuple(x["EXPR_continuous_label"].shape) == (len(batch), 4, 1) and names[-1] == "b"
    for i, b in enumerate(batch):
        ref = frames_transform(b[0]["video"].numpy(), 48, 40, int(cx[i, 0]), int(cx[i, 1]), bool(cx[i, 2]))
        assert np.array_equal(x["video"][i].cpu().numpy(), ref)
