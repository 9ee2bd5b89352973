"""Frame- and video-level scores (host side, numpy).

Mirror of the parts of the reference's metrics.py the hot path feeds (metrics.py:43-193):
prediction formatting (frame level; video level by majority vote / mean logits / mean
probabilities, with the C-EXPR-DB 'Other' class option), per-class / macro / weighted F1 (the
definitions sklearn.metrics.f1_score uses, restated so the GPU box needs no sklearn), accuracy and
the row-normalised confusion matrix.
"""
from collections import Counter

import numpy as np

FRM_VOTE, FRM_AVG_LOGITS, FRM_AVG_PROBS = "FRAMES_MAJORITY_VOTING", "FRAMES_AVERAGE_LOGITS", "FRAMES_AVERAGE_PROBS"
MACRO_F1, W_F1, CL_ACC, CFUSE_MATRIX = "MACRO_F1", "W_F1", "CL_ACC", "CONFUSION_MATRIX"
FRAME_LEVEL, VIDEO_LEVEL = "FRAME_LEVEL", "VIDEO_LEVEL"


def softmax(x):
    assert x.ndim == 2, x.ndim
    e = np.exp(x)  # like the reference: no max subtraction
    return e / e.sum(axis=1, keepdims=True)


def format_trg_pred_frames(data, ignore_class=None):
    limited = isinstance(ignore_class, int)
    preds, trgs = [], []
    for vid in data:
        labels, logits = data[vid]["labels"].tolist(), data[vid]["logits"]
        if limited:
            logits = logits[:, :-1]
        p = np.argmax(logits, axis=1).flatten().tolist()
        assert len(p) == len(labels), f"{len(p)} | {len(labels)}"
        for i, l in enumerate(labels):
            if limited and l == ignore_class:
                continue
            trgs.append(l)
            preds.append(p[i])
    return preds, trgs


def format_trg_pred_video(data, ignore_class=None):
    limited = isinstance(ignore_class, int)
    preds, trgs = [], []
    for vid in data:
        labels = data[vid]["labels"]
        uniq = np.unique(labels).tolist()
        assert len(uniq) == 1, len(uniq)
        if limited and uniq[0] == ignore_class:
            continue
        logits = data[vid]["logits"]
        if limited:
            logits = logits[:, :-1]
        frame_preds = np.argmax(logits, axis=1).flatten().tolist()
        preds.append({FRM_VOTE: Counter(frame_preds).most_common(1)[0][0],
                      FRM_AVG_LOGITS: int(np.argmax(logits.mean(axis=0))),
                      FRM_AVG_PROBS: int(np.argmax(softmax(logits).mean(axis=0)))})
        trgs.append(uniq[0])
    return preds, trgs


def per_class_f1(trgs, preds):
    """F1 per class over the sorted union of labels present in trgs or preds (sklearn's default)."""
    t, p = np.asarray(trgs).astype(np.int64), np.asarray(preds).astype(np.int64)
    classes = np.unique(np.concatenate([t, p]))
    f1, support = [], []
    for c in classes:
        tp = float(np.sum((t == c) & (p == c)))
        fp = float(np.sum((t != c) & (p == c)))
        fn = float(np.sum((t == c) & (p != c)))
        den = 2 * tp + fp + fn
        f1.append(2 * tp / den if den > 0 else 0.0)
        support.append(float(np.sum(t == c)))
    return np.array(f1), np.array(support), classes


def compute_f1_score(trgs, preds, f1_type):
    f1s, support, _ = per_class_f1(trgs, preds)
    if f1_type == MACRO_F1:
        return f1s, float(np.mean(f1s))
    if f1_type == W_F1:
        return f1s, float(np.sum(f1s * support) / max(np.sum(support), 1.0))
    raise NotImplementedError(f1_type)


def compute_class_acc(trgs, preds):
    return float((np.asarray(trgs, dtype=np.float32) == np.asarray(preds, dtype=np.float32)).mean() * 100.0)


def compute_confusion_matrix(trgs, preds):
    t, p = np.asarray(trgs).astype(np.int64), np.asarray(preds).astype(np.int64)
    classes = np.unique(np.concatenate([t, p]))
    idx = {c: i for i, c in enumerate(classes)}
    m = np.zeros((len(classes), len(classes)))
    for a, b in zip(t, p):
        m[idx[a], idx[b]] += 1
    rows = m.sum(axis=1, keepdims=True)
    return np.divide(m, rows, out=np.zeros_like(m), where=rows > 0)


def compute_perf(data, ignore_classes=(None,)):
    """{ignore_class: {metric: {FRAME_LEVEL: {...}, VIDEO_LEVEL: {agg: {...}}}}} like Trainer.compute_perf
    (trainer.py:525-605)."""
    out = {}
    for ic in ignore_classes:
        perf = {m: {FRAME_LEVEL: None, VIDEO_LEVEL: {}} for m in (MACRO_F1, W_F1, CL_ACC, CFUSE_MATRIX)}
        preds, trgs = format_trg_pred_frames(data, ic)
        f1s, macro = compute_f1_score(trgs, preds, MACRO_F1)
        perf[MACRO_F1][FRAME_LEVEL] = {"master": macro, "per_cl": f1s}
        perf[W_F1][FRAME_LEVEL] = {"master": compute_f1_score(trgs, preds, W_F1)[1], "per_cl": f1s}
        acc = compute_class_acc(trgs, preds)
        perf[CL_ACC][FRAME_LEVEL] = {"master": acc, "per_cl": acc}
        cm = compute_confusion_matrix(trgs, preds)
        perf[CFUSE_MATRIX][FRAME_LEVEL] = {"master": cm, "per_cl": cm}
        vpreds, vtrgs = format_trg_pred_video(data, ic)
        for k in (FRM_VOTE, FRM_AVG_LOGITS, FRM_AVG_PROBS):
            pk = [item[k] for item in vpreds]
            f1s, macro = compute_f1_score(vtrgs, pk, MACRO_F1)
            perf[MACRO_F1][VIDEO_LEVEL][k] = {"master": macro, "per_cl": f1s}
            perf[W_F1][VIDEO_LEVEL][k] = {"master": compute_f1_score(vtrgs, pk, W_F1)[1], "per_cl": f1s}
            acc = compute_class_acc(vtrgs, pk)
            perf[CL_ACC][VIDEO_LEVEL][k] = {"master": acc, "per_cl": acc}
            cm = compute_confusion_matrix(vtrgs, pk)
            perf[CFUSE_MATRIX][VIDEO_LEVEL][k] = {"master": cm, "per_cl": cm}
        out[ic] = perf
    return out
