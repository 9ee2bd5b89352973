"""Oracle: LFAN assembly, loss and optimiser step (test infrastructure only).

Follows /root/reference/models/model.py:487-526 (LFAN.forward),
/root/reference/trainer.py:365-391 (one optimisation step),
/root/reference/experiment.py:133 (CrossEntropyLoss, mean) and
/root/reference/instantiators.py:74-79 (SGD built WITHOUT lr -> torch default 1e-3).
"""
import torch
import torch.nn.functional as F

from .fusion import lfan_fusion_forward
from .ir50 import BN_EPS, BN_MOMENTUM, ir50_forward
from .tcn import tcn_forward


def _bn1d(x, sd, key, train, new_buffers):
    """BatchNorm1d on [B,C,L]."""
    w, b = sd[key + ".weight"], sd[key + ".bias"]
    if train:
        mean = x.mean((0, 2))
        var = x.var((0, 2), unbiased=False)
        if new_buffers is not None:
            n = x.shape[0] * x.shape[2]
            new_buffers[key + ".running_mean"] = (1 - BN_MOMENTUM) * sd[key + ".running_mean"] + BN_MOMENTUM * mean
            new_buffers[key + ".running_var"] = ((1 - BN_MOMENTUM) * sd[key + ".running_var"]
                                                 + BN_MOMENTUM * var * n / max(n - 1, 1))
    else:
        mean, var = sd[key + ".running_mean"], sd[key + ".running_var"]
    inv = torch.rsqrt(var + BN_EPS)
    return (x - mean.view(1, -1, 1)) * (inv * w).view(1, -1, 1) + b.view(1, -1, 1)


def lfan_forward(inputs, sd, modalities, modal_dim=32, num_heads=2, train=False,
                 backbone_train=None, masks=None, new_buffers=None):
    """inputs: dict with keys in ``modalities`` order:
    video [B,L,3,H,W]; vggish [B,1,L,128]; bert [B,1,L,768]; logmel [B,64,L,96] (model.py:500-508: one 96 x 64
    log-mel example per clip frame, run through VGGish inside forward)  ->  [B,L,n_cls].

    ``masks`` (train only, optional) = {"head": [B*L,512,h,w], "tcn": {m: [(m1,m2)..]},
    "fusion": [B,L,modal_dim*M]} pre-scaled dropout masks; missing -> identity.
    ``backbone_train`` defaults to ``train`` (the reference calls model.train() on
    the whole module, frozen backbone included -- trainer.py:318).
    """
    if backbone_train is None:
        backbone_train = train
    masks = masks or {}
    feats = {}
    for m in modalities:
        x = inputs[m]
        if m == "video":
            bsz, length = x.shape[:2]
            x = ir50_forward(x.reshape(-1, *x.shape[2:]), sd, "spatial.visual.backbone.",
                             train=backbone_train, head_dropout_mask=masks.get("head"),
                             new_buffers=new_buffers)
            x = x.view(bsz, length, -1)
        elif m == "logmel":
            from .vggish import vggish_forward
            bsz, height, length, width = x.shape
            x = vggish_forward(x.permute(0, 2, 3, 1).reshape(-1, width, height), sd, "spatial.audio.backbone.")
            x = x.view(bsz, length, -1)
        else:
            x = x.squeeze(1)
        x = x.transpose(1, 2)  # [B,C,L]
        x = tcn_forward(x, sd, f"temporal.{m}.", (masks.get("tcn") or {}).get(m))
        x = _bn1d(x, sd, f"bn.{m}", train, new_buffers)
        feats[m] = x.transpose(1, 2)  # [B,L,C]
    follower = lfan_fusion_forward(feats, sd, modalities, modal_dim, num_heads,
                                   dropout_mask=masks.get("fusion"))
    z = torch.cat((feats[modalities[0]], follower), dim=-1)
    return F.linear(z, sd["regressor.weight"], sd["regressor.bias"])


def cross_entropy_mean(logits, labels):
    """logits [B,L,C], labels [B,L,1] float (as the reference ships them) -> scalar."""
    b, l, c = logits.shape
    return F.cross_entropy(logits.reshape(b * l, c), labels.reshape(b * l).long())


def sgd_nesterov_step(params, grads, bufs, lr=1e-3, momentum=0.9, weight_decay=1e-4):
    """torch.optim.SGD(nesterov=True, dampening=0) single step, functional.
    ``bufs`` entries may be None on the first step.  Returns (new_params, new_bufs)."""
    new_p, new_b = [], []
    for p, g, buf in zip(params, grads, bufs):
        g = g + weight_decay * p
        buf = g.clone() if buf is None else momentum * buf + g
        g = g + momentum * buf
        new_p.append(p - lr * g)
        new_b.append(buf)
    return new_p, new_b
