"""Evaluation scores from confusion counts gathered ON THE DEVICE.

``DeviceEvalAccumulator`` receives each video's logits and labels as GPU tensors (nothing is copied to the host per
video), accumulates the frame-level and the three video-level confusion count matrices with ``cer_eval_accumulate`` and
turns them into the reference's score dictionary (trainer.py:525-605 / metrics.py:148-193: macro / weighted F1 with
sklearn's "classes present in targets or predictions" convention, accuracy, row-normalised confusion matrix) after ONE
[4, C, C] device-to-host copy per evaluation and ignore-class setting.
"""
import numpy as np
import torch

from . import _lib
from ._lib import check, current_stream, ptr
from .metrics import (CFUSE_MATRIX, CL_ACC, FRAME_LEVEL, FRM_AVG_LOGITS, FRM_AVG_PROBS, FRM_VOTE, MACRO_F1, VIDEO_LEVEL, W_F1)


def stitch_windows(win_out, starts, total):
    """win_out [nw, Lw, C] (GPU), starts: list of window start frames -> [total, C]; trainer.py:832-892 in one launch."""
    if not (isinstance(win_out, torch.Tensor) and win_out.is_cuda and win_out.dtype == torch.float32 and win_out.dim() == 3):
        raise ValueError("stitch_windows: expected a [n_windows, window_length, n_classes] float32 GPU tensor")
    nw, lw, c = win_out.shape
    starts = [int(s) for s in starts]
    if len(starts) != nw:
        raise ValueError(f"stitch_windows: {nw} windows but {len(starts)} start frames (one video per call)")
    if nw and (min(starts) < 0 or max(starts) + lw > total):
        raise ValueError(f"stitch_windows: a window of {lw} frames starting at {max(starts)} leaves the {total}-frame video")
    st = torch.tensor(list(starts), dtype=torch.int32, device=win_out.device)
    out = torch.empty((total, c), device=win_out.device, dtype=torch.float32)
    check(_lib.load().cer_window_stitch(ptr(win_out.contiguous()), ptr(st), nw, lw, c, total, ptr(out), current_stream()),
          "cer_window_stitch")
    return out


def scores_from_confusion(cm):
    """Counts [C, C] (rows = targets, columns = predictions) -> the four entries the reference derives with sklearn."""
    cm = np.asarray(cm, dtype=np.float64)
    present = (cm.sum(0) + cm.sum(1)) > 0            # sklearn: labels = sorted union of targets and predictions
    sub = cm[np.ix_(present, present)]
    tp = np.diag(sub)
    fp, fn = sub.sum(0) - tp, sub.sum(1) - tp
    den = 2 * tp + fp + fn
    f1 = np.divide(2 * tp, den, out=np.zeros_like(tp), where=den > 0)
    support = sub.sum(1)
    total = sub.sum()
    rows = sub.sum(1, keepdims=True)
    return {"f1_per_class": f1, "macro_f1": float(f1.mean()) if f1.size else 0.0,
            "weighted_f1": float((f1 * support).sum() / max(support.sum(), 1.0)),
            "accuracy": float(tp.sum() / total * 100.0) if total > 0 else 0.0,
            "confusion": np.divide(sub, rows, out=np.zeros_like(sub), where=rows > 0)}


class DeviceEvalAccumulator:
    def __init__(self, n_classes, ignore_classes=(None,), device="cuda", keep_video_predictions=False):
        self.c, self.ignore_classes, self.device = n_classes, tuple(ignore_classes), device
        self.cm = {ic: torch.zeros((4, n_classes, n_classes), dtype=torch.int64, device=device) for ic in self.ignore_classes}
        self.bad = torch.zeros((1,), dtype=torch.int32, device=device)
        self.keep = keep_video_predictions
        self.video_predictions = []

    def add(self, logits, labels, video_offsets=None):
        """logits [R, C] float32 GPU, labels [R] (float or long) GPU; ``video_offsets`` = row offsets [V+1] when several
        videos are concatenated (default: one video)."""
        if not (logits.is_cuda and logits.dtype == torch.float32 and logits.dim() == 2 and logits.shape[1] == self.c):
            raise ValueError("logits: expected a [R, n_classes] float32 GPU tensor")
        if not (isinstance(labels, torch.Tensor) and labels.is_cuda):
            raise ValueError("labels: expected a GPU tensor (one label per logits row)")
        logits = logits.contiguous()
        labels = labels.reshape(-1).float().contiguous()
        r = logits.shape[0]
        if labels.numel() != r:
            raise ValueError("one label per logits row")
        off = torch.tensor([0, r] if video_offsets is None else list(video_offsets), dtype=torch.int32, device=logits.device)
        v = off.numel() - 1
        lib = _lib.load()
        for ic, cm in self.cm.items():
            vp = torch.empty((v, 3), dtype=torch.int32, device=logits.device) if self.keep else None
            check(lib.cer_eval_accumulate(ptr(logits), ptr(labels), ptr(off), v, r, self.c, -1 if ic is None else int(ic),
                                          ptr(cm[0]), ptr(cm[1:]), ptr(vp), ptr(self.bad), current_stream()), "cer_eval_accumulate")
            if self.keep:
                self.video_predictions.append((ic, vp))

    def compute(self):
        """One device -> host copy per ignore-class setting; the reference's nested score dictionary."""
        if int(self.bad.item()) != 0:
            raise AssertionError(f"{int(self.bad.item())} frame(s) / video(s) with labels outside [0, n_classes) or mixed labels "
                                 "inside one video (the reference asserts len(unique) == 1, metrics.py:104-105)")
        out = {}
        for ic, cm in self.cm.items():
            counts = cm.cpu().numpy()
            perf = {m: {FRAME_LEVEL: None, VIDEO_LEVEL: {}} for m in (MACRO_F1, W_F1, CL_ACC, CFUSE_MATRIX)}

            def put(level, key, s):
                entries = {MACRO_F1: {"master": s["macro_f1"], "per_cl": s["f1_per_class"]},
                           W_F1: {"master": s["weighted_f1"], "per_cl": s["f1_per_class"]},
                           CL_ACC: {"master": s["accuracy"], "per_cl": s["accuracy"]},
                           CFUSE_MATRIX: {"master": s["confusion"], "per_cl": s["confusion"]}}
                for m, e in entries.items():
                    if level == FRAME_LEVEL:
                        perf[m][FRAME_LEVEL] = e
                    else:
                        perf[m][VIDEO_LEVEL][key] = e
            put(FRAME_LEVEL, None, scores_from_confusion(counts[0]))
            for i, k in enumerate((FRM_VOTE, FRM_AVG_LOGITS, FRM_AVG_PROBS)):
                put(VIDEO_LEVEL, k, scores_from_confusion(counts[1 + i]))
            out[ic] = perf
        return out
