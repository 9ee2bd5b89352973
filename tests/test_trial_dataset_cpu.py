"""Trial-format Dataset mirror (base/dataset.py:188-270,433-631 restated on synthetic .npy trials; the reference module
itself needs torchvision, absent here: parity unpinned against the import, pinned by these rule-by-rule checks)."""
import random

import numpy as np
import pytest
import torch

from feature_vs_text_compound_emotion_amd.trial_dataset import (TRAINSET, VALIDSET, TrialDataset, calculate_mean_std,
                                                                  windowed_trial_list, windowing)

MODS = ["video", "vggish", "bert", "EXPR_continuous_label"]
MULT = {m: 1 for m in MODS}
DIMS = {"video": (16, 16, 3), "vggish": (128,), "bert": (768,), "EXPR_continuous_label": (1,)}


def _make_trial(root, name, n, seed):
    rng = np.random.default_rng(seed)
    d = root / name
    d.mkdir()
    np.save(d / "video.npy", rng.integers(0, 256, (n, 16, 16, 3), dtype=np.uint8))
    np.save(d / "vggish.npy", rng.normal(2.0, 3.0, (n, 128)).astype(np.float32))
    np.save(d / "bert.npy", rng.normal(-1.0, 0.5, (n, 768)).astype(np.float32))
    np.save(d / "EXPR_continuous_label.npy", rng.integers(0, 7, n).astype(np.int64))
    return [str(d), name, n]


def test_windowing_rules():
    x = np.arange(10)
    assert [w.tolist() for w in windowing(x, 6, 4)] == [[0, 1, 2, 3, 4, 5], [4, 5, 6, 7, 8, 9]]
    # a tail the hops do not reach -> one more window aligned to the end
    assert [w.tolist() for w in windowing(np.arange(11), 6, 4)][-1] == [5, 6, 7, 8, 9, 10]
    assert len(windowing(np.arange(11), 6, 4)) == 3
    # '>' (dataset.py:438): a sequence of exactly the window length, or shorter, stays whole
    assert [w.tolist() for w in windowing(np.arange(6), 6, 4)] == [list(range(6))]
    assert [w.tolist() for w in windowing(np.arange(3), 6, 4)] == [[0, 1, 2]]


def test_windowed_trial_list_eval_switch(tmp_path):
    t = [_make_trial(tmp_path, "a", 10, 0)]
    assert len(windowed_trial_list(t, 6, 4, split=TRAINSET)) == 2
    whole = windowed_trial_list(t, 6, 4, split=VALIDSET, window_eval=False)
    assert len(whole) == 1 and whole[0][3].tolist() == list(range(10))
    assert len(windowed_trial_list(t, 6, 4, windowing_on=False)) == 1


def test_getitem_layout_padding_and_standardisation(tmp_path):
    trials = [_make_trial(tmp_path, "long", 10, 1), _make_trial(tmp_path, "short", 4, 2)]
    data = windowed_trial_list(trials, 6, 4)
    assert [d[1] for d in data] == ["long", "long", "short"]
    ms = calculate_mean_std(data)
    allv = np.concatenate([np.load(tmp_path / "long" / "vggish.npy")] * 2 + [np.load(tmp_path / "short" / "vggish.npy")])
    assert np.allclose(ms["vggish"]["mean"], allv.sum(0) / (len(allv) + 1e-10))
    assert np.allclose(ms["vggish"]["std"], allv.std(0, ddof=1), rtol=1e-5)
    ds = TrialDataset(data, MODS, MULT, DIMS, window_length=6, mode=TRAINSET, mean_std=ms)
    ex, trial, length, index = ds[1]
    assert trial == "long" and length == 10 and index.tolist() == [4, 5, 6, 7, 8, 9]
    assert ex["video"].dtype == torch.uint8 and tuple(ex["video"].shape) == (6, 16, 16, 3)
    assert np.array_equal(ex["video"].numpy(), np.load(tmp_path / "long" / "video.npy")[4:10])
    assert tuple(ex["vggish"].shape) == (1, 6, 128) and tuple(ex["bert"].shape) == (1, 6, 768)   # [B,1,L,C] after collate
    ref = (np.load(tmp_path / "long" / "vggish.npy")[4:10] - ms["vggish"]["mean"]) / ms["vggish"]["std"]
    assert np.abs(ex["vggish"][0].numpy() - ref).max() < 1e-5
    assert ex["EXPR_continuous_label"].dtype == np.float32 and ex["EXPR_continuous_label"].shape == (6, 1)
    # short trial: padded to the window by repeating the last element, index becomes arange(window)
    ex, trial, length, index = ds[2]
    assert trial == "short" and length == 4 and index.tolist() == list(range(6))
    v = np.load(tmp_path / "short" / "video.npy")
    assert np.array_equal(ex["video"].numpy()[:4], v) and np.array_equal(ex["video"].numpy()[4], v[3])
    assert np.array_equal(ex["video"].numpy()[5], v[3])
    lab = np.load(tmp_path / "short" / "EXPR_continuous_label.npy")
    assert ex["EXPR_continuous_label"][:, 0].tolist() == lab.tolist() + [lab[-1]] * 2


def test_one_random_draw_per_modality_in_order(tmp_path):
    data = windowed_trial_list([_make_trial(tmp_path, "a", 8, 3)], 6, 4)
    ds = TrialDataset(data, MODS, MULT, DIMS, 6, TRAINSET, mean_std=calculate_mean_std(data))
    random.seed(5)
    ds[0]
    got = random.getstate()
    random.seed(5)
    for _ in MODS:
        random.randint(0, 0)
    assert got == random.getstate()


def test_missing_label_file_gives_zero_dummies(tmp_path):
    t = _make_trial(tmp_path, "test_trial", 7, 4)
    (tmp_path / "test_trial" / "EXPR_continuous_label.npy").unlink()
    data = windowed_trial_list([t], 6, 4)
    ds = TrialDataset(data, MODS, MULT, DIMS, 6, "test", mean_std=calculate_mean_std(data))
    assert float(np.abs(ds[0][0]["EXPR_continuous_label"]).max()) == 0.0
