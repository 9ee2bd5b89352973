"""Host loop around the hot path: one optimisation step, whole-video inference, window stitching.

Mirror of the reference's ``Trainer`` for the parts that call the model (trainer.py:315-434
``train_one_epoch``, :436-523 ``inference``, :788-892 ``window_input`` /
``inference_forward_windows``, :894-912 ``windowing``) plus the dataset-side window rule
(base/dataset.py:434-453, which uses ``>`` where the trainer uses ``>=``).  Logging, checkpoints,
LR-scheduler bookkeeping and best-model selection stay with the reference's own trainer; this class
is what ``bench.py`` and the parity tests drive, and what a maintainer can diff against.
"""
from collections import Counter

import numpy as np
import torch

from . import metrics
from .lfan import cross_entropy_loss

VIDEO, VGGISH, BERT = "video", "vggish", "bert"
EXPR = "EXPR_continuous_label"


def windowing(x, window_length, hop_length):
    """trainer.py:894-912 (inclusive ``>=``): inference-time windows of a whole video."""
    n = len(x)
    if n >= window_length:
        steps = (n - window_length) // hop_length + 1
        out = [x[i * hop_length:i * hop_length + window_length] for i in range(steps)]
        if out[-1][-1] < n - 1:
            out.append(x[-window_length:])
        return out
    return [x]


def dataset_windowing(x, window_length, hop_length):
    """base/dataset.py:434-453 (strict ``>``): training-time windows of a trial."""
    n = len(x)
    if n > window_length:
        steps = (n - window_length) // hop_length + 1
        out = [x[i * hop_length:i * hop_length + window_length] for i in range(steps)]
        if out[-1][-1] < n - 1:
            out.append(x[-window_length:])
        return out
    return [x]


def _num_frames(modality, t):
    if modality == VIDEO:
        return t.shape[1]
    if modality in (VGGISH, BERT):
        return t.shape[2]
    raise NotImplementedError(modality)


def _take(modality, t, wd):
    return t[:, wd, ...] if modality == VIDEO else t[:, :, wd, ...]


class Trainer:
    def __init__(self, model, optimizer=None, criterion=None, device="cuda", window_length=300, hop_length=200,
                 model_name="LFAN", train_batch_size=2, number_classes=7, data_parallel=None, ignore_classes=(None,)):
        self.model, self.optimizer, self.device = model, optimizer, device
        self.criterion = criterion if criterion is not None else cross_entropy_loss
        self.window_length, self.hop_length, self.model_name = window_length, hop_length, model_name
        self.train_batch_size, self.number_classes = train_batch_size, number_classes
        self.ddp, self.ignore_classes = data_parallel, ignore_classes

    # ------------------------------------------------------------------ training
    def _split(self, X):
        inputs = {k: v.to(self.device) for k, v in X.items()}
        labels = inputs.pop("continuous_label", None)
        if labels is None:
            labels = inputs.pop(EXPR, None)
        return inputs, labels

    def train_step(self, X, indices=None):
        """One iteration of trainer.py:345-391.  Returns the (detached) loss tensor."""
        inputs, labels = self._split(X)
        if labels.numel() == self.train_batch_size:  # the reference's "todo : fix this." label hack (:360-363)
            n = len(indices[0]) if indices is not None else labels.shape[1]
            labels = torch.zeros((self.train_batch_size, n, 1), dtype=torch.float32, device=self.device)
        if self.ddp is not None:
            self.ddp.zero_grad()
        else:
            self.optimizer.zero_grad(set_to_none=True)
        outputs = self.model(inputs)
        bsz, nfms, d = labels.shape
        assert d == 1, d
        assert outputs.ndim == 3 and tuple(outputs.shape) == (bsz, nfms, self.number_classes), tuple(outputs.shape)
        loss = self.criterion(outputs.contiguous().view(bsz * nfms, -1), labels.contiguous().view(bsz * nfms).long())  # trainer.py:380-383
        loss.backward()
        if self.ddp is not None:
            self.ddp.all_reduce_gradients()
        self.optimizer.step()
        return loss.detach()

    def train_one_epoch(self, dataloader):
        self.model.train()
        running, count = 0.0, 0
        for X, trials, lengths, indices in dataloader:
            running = running + self.train_step(X, indices)
            count += 1
        return float(running / max(count, 1))

    # ------------------------------------------------------------------ inference
    def window_input(self, data):
        sizes = [[t.shape[0], _num_frames(m, t)] for m, t in data.items()]
        for s in sizes:
            assert s == sizes[0], f"{s} | {sizes[0]}"
        windows = windowing(np.arange(sizes[0][1]), self.window_length, self.hop_length)
        return [[{m: _take(m, t, wd) for m, t in data.items()}, wd] for wd in windows]

    def inference_forward_windows(self, data, aggregate="device"):
        """Forward a video longer than the model's window (trainer.py:832-892).  ``aggregate="device"`` (default): all
        windows go through the model as ONE batch (eval mode: clips are independent), then one kernel scatter-adds them
        in window order and divides by the overlap counts (``eval_device.stitch_windows``).  ``aggregate="host"`` is the
        reference's own sequence -- one forward per window, indexed adds, Counter of the frame indices -- kept as the
        checker of the device path and for the CPU tests of the window rules."""
        total = _num_frames(*next(iter(data.items())))
        chunks = self.window_input(data)
        last = int(chunks[-1][1][-1])
        assert total == last + 1, f"{total} | {last + 1}"
        if aggregate == "device":
            from .eval_device import stitch_windows
            batch = {m: torch.cat([c[m] for c, _ in chunks], dim=0).contiguous() for m in chunks[0][0]}
            out = self.model(batch)                                   # [n_windows, window_length, n_cls]
            assert out.ndim == 3, out.ndim
            return stitch_windows(out, [int(wd[0]) for _, wd in chunks], total).unsqueeze(0)
        results = []
        for chunk, wd in chunks:
            out = self.model({m: t.contiguous() for m, t in chunk.items()})
            assert out.ndim == 3, out.ndim
            results.append((out, wd))
        final = torch.zeros((results[-1][0].shape[0], total, results[-1][0].shape[2]), device=results[-1][0].device,
                            dtype=results[-1][0].dtype)
        idx = []
        for out, wd in results:
            final[:, wd, ...] = final[:, wd, ...] + out
            idx += wd.tolist()
        counts = sorted(Counter(idx).items())
        freqs = torch.tensor([c for _, c in counts], dtype=final.dtype, device=final.device).view(1, -1, 1)
        where = np.asarray([i for i, _ in counts], dtype=np.int64)
        final[:, where, ...] = final[:, where, ...] / freqs
        return final

    @torch.no_grad()
    def inference(self, dataloader, keep_logits=False, aggregate="device"):
        """Trainer.inference (trainer.py:436-523).  ``aggregate="device"`` (default): logits stay on the GPU, each video is
        folded into device-side confusion counts (``DeviceEvalAccumulator``) and the scores come from one small copy at
        the end; ``keep_logits=True`` additionally returns the reference's per-video ``{labels, logits}`` dictionary (one
        copy per video).  ``aggregate="host"``: the reference's own flow -- per-video copies, numpy scores
        (``metrics.compute_perf``) -- the checker of the device path."""
        self.model.eval()
        per_video = {}
        acc = None
        if aggregate == "device":
            from .eval_device import DeviceEvalAccumulator
            acc = DeviceEvalAccumulator(self.number_classes, self.ignore_classes, device=self.device)
        elif aggregate != "host":
            raise ValueError(aggregate)
        for X, trials, lengths, indices in dataloader:
            inputs, labels = self._split(X)
            nframes = 0
            for m, t in inputs.items():
                assert t.shape[0] == 1, f"{t.shape[0]} | {m}"
                nframes = _num_frames(m, t)
            if labels.numel() == self.train_batch_size:
                labels = torch.zeros((self.train_batch_size, len(indices[0]), 1), dtype=torch.float32, device=self.device)
            if nframes > self.window_length and self.model_name == "LFAN":
                outputs = self.inference_forward_windows(inputs, aggregate)
            else:
                outputs = self.model(inputs)
            bsz, nfms, d = labels.shape
            assert d == 1 and tuple(outputs.shape) == (bsz, nfms, self.number_classes), tuple(outputs.shape)
            if acc is not None:
                acc.add(outputs.contiguous().view(bsz * nfms, -1), labels.contiguous().view(bsz * nfms))
            if keep_logits or acc is None:
                per_video[trials[0]] = {"labels": labels.contiguous().view(bsz * nfms).long().cpu().numpy().flatten(),
                                        "logits": outputs.contiguous().view(bsz * nfms, -1).cpu().numpy()}
        if acc is not None:
            return acc.compute(), per_video
        return metrics.compute_perf(per_video, self.ignore_classes), per_video
