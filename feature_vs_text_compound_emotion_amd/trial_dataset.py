"""Host-side mirror of the reference's Dataset over the on-disk ``.npy`` trial format, feeding the GPU input
pipeline.

Reference: ``base/dataset.py`` -- ``DataArranger.windowing`` (:433-452), ``generate_partitioned_trial_list`` (:188-270),
``calculate_mean_std`` (:272-325), ``GenericDataset`` (:456-631).  A trial is a directory
``<dataset>/features/<data_folder>/<trial>/`` holding ``video.npy [N,256,256,3] uint8``, ``vggish.npy [N,128]``,
``bert.npy [N,768]`` and ``EXPR_continuous_label.npy [N]`` (abaw5_pre_processing/dlib/compact_face_images.py:278-289).

What differs from the reference, on purpose: the video modality is returned as RAW uint8 frames ``[L,H,W,3]``;
resize / crop / flip / normalise then run on the GPU for the whole batch (``frames.FrameTransform``, bit-identical to
the PIL path) instead of per frame in DataLoader worker processes.  The reference's ``np.int8`` pad buffer for short
trials (:570-574) wraps values >= 128 and un-wraps them again in ``astype('uint8')`` (transforms3D.py:27), so keeping
uint8 yields the same pixels.  Parity of this module is pinned by tests that restate the reference's rules on synthetic
trials -- the reference's own ``base/dataset.py`` needs torchvision, which this image lacks ("parity unpinned" against
the imported reference).
"""
import os
import random

import numpy as np
import torch

TRAINSET, VALIDSET, TESTSET = "train", "valid", "test"  # constants.py of the reference


def windowing(x, window_length, hop_length):
    """dataset.py:433-452: hop over ``x``; a last full window is appended when the hops leave a tail; a sequence that
    is not LONGER than the window stays whole (note ``>``; trainer.py:898 uses ``>=`` for the inference windows)."""
    n = len(x)
    if n > window_length:
        steps = (n - window_length) // hop_length + 1
        out = [x[i * hop_length:i * hop_length + window_length] for i in range(steps)]
        if out[-1][-1] < n - 1:
            out.append(x[-window_length:])
        return out
    return [x]


def windowed_trial_list(trials, window_length, hop_length, split=TRAINSET, windowing_on=True, window_eval=True):
    """dataset.py:188-270 for one split: ``trials`` = [(path, trial, length)] -> [[path, trial, length, index]]."""
    out = []
    for path, trial, length in trials:
        wl = window_length
        if not windowing_on or (split in (TESTSET, VALIDSET) and not window_eval):
            wl = length
        for index in windowing(np.arange(length), wl, hop_length):
            out.append([path, trial, length, index])
    return out


def load_npy(path, feature):
    return np.load(os.path.join(path, feature + ".npy"), mmap_mode="c")


def calculate_mean_std(data, features=("vggish", "bert")):
    """dataset.py:272-325: per-component mean (sum / (n + 1e-10)) and unbiased std over the given window list
    (train + valid; every WINDOW entry counts its whole trial, like the reference)."""
    stats = {}
    for feature in features:
        n, sums = 0, 0
        for path, _, _, _ in data:
            s = np.asarray(load_npy(path, feature))
            n += s.shape[0]
            sums = sums + s.sum(axis=0)
        mean = sums / (n + 1e-10)
        n, sq = 0, 0
        for path, _, _, _ in data:
            s = np.asarray(load_npy(path, feature))
            n += s.shape[0]
            sq = sq + ((s - mean) ** 2).sum(axis=0)
        stats[feature] = {"mean": mean, "std": np.sqrt(sq / (n - 1))}
    return stats


class TrialDataset(torch.utils.data.Dataset):
    """``GenericDataset`` (dataset.py:456-631) with raw uint8 video.  ``__getitem__ -> (examples, trial, length, index)``."""

    def __init__(self, data_list, modality, multiplier, feature_dimension, window_length, mode, mean_std=None,
                 task="CLASSIFICATION", continuous_label_dim=None, time_delay=0):
        assert time_delay == 0, time_delay  # dataset.py:588
        self.data_list, self.modality, self.multiplier = data_list, list(modality), multiplier
        self.feature_dimension, self.window_length, self.mode = feature_dimension, window_length, mode
        self.mean_std, self.task, self.continuous_label_dim = mean_std, task, continuous_label_dim

    def __len__(self):
        return len(self.data_list)

    def _load(self, path, indices, feature):
        filename = os.path.join(path, feature + ".npy")
        data = np.zeros((len(indices),) + tuple(self.feature_dimension[feature]), dtype=np.float32)  # test set: dummy labels
        if os.path.isfile(filename):
            data = np.load(filename, mmap_mode="c")[indices]
            if "continuous_label" in feature:
                if self.task == "CLASSIFICATION":
                    assert data.ndim == 1, data.ndim
                    data = data[:, None]
                else:
                    data = data[:, self.continuous_label_dim]
                    if data.ndim == 1:
                        data = data[:, None]
        return data

    def _example(self, path, length, index, feature):
        x = random.randint(0, self.multiplier[feature] - 1)  # one draw per feature, in modality order (dataset.py:562)
        idx = index * self.multiplier[feature] + x
        if length < self.window_length:
            dtype = np.uint8 if feature == "video" else np.float32
            ex = np.zeros((self.window_length,) + tuple(self.feature_dimension[feature]), dtype=dtype)
            ex[index] = self._load(path, idx, feature)
            ex[index.size:] = ex[index[-1]]  # repeat the last element (dataset.py:578-582)
        else:
            ex = self._load(path, idx, feature)
        if "continuous_label" in feature:
            return np.asarray(ex, dtype=np.float32)
        if feature == "video":
            return torch.from_numpy(np.ascontiguousarray(ex, dtype=np.uint8))  # [L,H,W,3]; transformed on the GPU
        t = torch.from_numpy(np.asarray(ex, dtype=np.float32))[None]  # transforms.ToTensor on a 2-D array: [1,L,C]
        if "logmel" in feature:
            return t
        avg = torch.from_numpy(np.asarray(self.mean_std[feature]["mean"], dtype=np.float64).reshape(1, -1))
        std = torch.from_numpy(np.asarray(self.mean_std[feature]["std"], dtype=np.float64).reshape(1, -1))
        return (t - avg.to(t.dtype)) / std.to(t.dtype)  # transforms.Normalize(mean=[1,C], std=[1,C])

    def __getitem__(self, i):
        path, trial, length, index = self.data_list[i]
        examples = {f: self._example(path, length, index, f) for f in self.modality}
        if len(index) < self.window_length:
            index = np.arange(self.window_length)
        return examples, trial, length, index


def collate_to_device(batch, device, frame_transform=None, crop_xyf=None):
    """Default-collate semantics for the (examples, trial, length, index) tuples, then the batch goes to ``device`` and
    the video modality through the GPU ``FrameTransform``: returns (inputs dict for the model incl. labels, trials,
    lengths, indices)."""
    examples = [b[0] for b in batch]
    out = {}
    for k in examples[0]:
        vals = [e[k] if torch.is_tensor(e[k]) else torch.from_numpy(e[k]) for e in examples]
        out[k] = torch.stack(vals).to(device, non_blocking=True)
    if "video" in out and frame_transform is not None:
        out["video"] = frame_transform(out["video"], crop_xyf=crop_xyf)
    return out, [b[1] for b in batch], torch.tensor([b[2] for b in batch]), torch.from_numpy(np.stack([b[3] for b in batch]))
