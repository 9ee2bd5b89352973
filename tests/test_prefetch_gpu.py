"""Pinned staging + prefetch: batches that come out of ``DevicePrefetcher`` equal the ones the plain
``collate_to_device`` path builds from the same trials, on synthetic ``.npy`` trial directories in the reference's on-disk
format (abaw5_pre_processing/dlib/compact_face_images.py:278-289)."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

MODS = ["video", "vggish", "bert", "EXPR_continuous_label"]
DIMS = {"video": (64, 64, 3), "vggish": (128,), "bert": (768,), "EXPR_continuous_label": (1,)}


def _make_trials(root, lengths, seed=0):
    rng = np.random.default_rng(seed)
    trials = []
    for i, n in enumerate(lengths):
        d = os.path.join(root, f"trial{i}")
        os.makedirs(d)
        np.save(os.path.join(d, "video.npy"), rng.integers(0, 256, (n, 64, 64, 3), dtype=np.uint8))
        np.save(os.path.join(d, "vggish.npy"), rng.standard_normal((n, 128)).astype(np.float32))
        np.save(os.path.join(d, "bert.npy"), rng.standard_normal((n, 768)).astype(np.float32))
        np.save(os.path.join(d, "EXPR_continuous_label.npy"), np.full((n,), i % 7, dtype=np.float32))
        trials.append((d, f"trial{i}", n))
    return trials


def _dataset(tmp_path):
    from feature_vs_text_compound_emotion_amd import trial_dataset as td
    trials = _make_trials(str(tmp_path), [40, 9, 16, 33, 21])
    windows = td.windowed_trial_list(trials, 16, 8)
    stats = td.calculate_mean_std(windows)
    ds = td.TrialDataset(windows, MODS, {m: 1 for m in MODS}, DIMS, 16, "train", mean_std=stats)
    return ds, td


@pytest.mark.parametrize("workers,depth", [(1, 1), (3, 2), (6, 3)])
def test_prefetched_batches_equal_the_plain_collate_path(tmp_path, workers, depth):
    from feature_vs_text_compound_emotion_amd.frames import FrameTransform
    from feature_vs_text_compound_emotion_amd.prefetch import DevicePrefetcher
    ds, td = _dataset(tmp_path)
    order = list(range(len(ds)))
    batches = [order[i:i + 3] for i in range(0, len(order), 3)]
    ft = FrameTransform(size=48, crop=40, train=False)      # centre crop, no flip: no random draws to race over
    pf = DevicePrefetcher(ds, batches, "cuda", ft, num_workers=workers, depth=depth)
    got = list(pf)
    pf.close()
    assert len(got) == len(batches)
    assert all(t.is_pinned() for slot in [pf._last] if slot for t in slot["pinned"].values())
    for idxs, (out, trials, lengths, indices) in zip(batches, got):
        ref, rt, rl, ri = td.collate_to_device([ds[i] for i in idxs], "cuda", ft, crop_xyf=ft.draw(len(idxs)))
        assert trials == rt and torch.equal(lengths, rl) and torch.equal(indices, ri)
        assert set(out) == set(ref)
        for k in ref:
            assert out[k].is_cuda and out[k].shape == ref[k].shape, k
            assert torch.equal(out[k], ref[k]), k
        assert out["video"].dtype == torch.float32 and tuple(out["video"].shape[2:]) == (3, 40, 40)


def test_prefetcher_train_transform_and_error_propagation(tmp_path):
    from feature_vs_text_compound_emotion_amd.frames import FrameTransform
    from feature_vs_text_compound_emotion_amd.prefetch import DevicePrefetcher
    ds, td = _dataset(tmp_path)
    pf = DevicePrefetcher(ds, [[0, 1], [2, 3]], "cuda", FrameTransform(train=True), num_workers=2, depth=2)
    for out, trials, lengths, indices in pf:
        assert torch.isfinite(out["video"]).all() and out["video"].abs().max() <= 1.0 + 1e-6
    pf.close()
    bad = DevicePrefetcher(ds, [[0, 10 ** 6]], "cuda", None, num_workers=1, depth=1)
    with pytest.raises(IndexError):
        next(bad)
