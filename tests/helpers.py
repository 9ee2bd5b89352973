"""Shared helpers for the parity tests (oracle-side plumbing only)."""
import os

import numpy as np
import torch

from feature_vs_text_compound_emotion_amd import synth

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
MODS = ["video", "vggish", "bert"]


def golden(name):
    return np.load(os.path.join(GOLDEN, name))


def masks_from_golden(g, mods=MODS, p_head=0.4, p=0.1):
    tcn = {m: [(torch.from_numpy(g[f"mask_tcn_{m}_{i}_0"]).float() / (1 - p),
                torch.from_numpy(g[f"mask_tcn_{m}_{i}_1"]).float() / (1 - p)) for i in range(4)] for m in mods}
    return {"head": torch.from_numpy(g["mask_head"]).float() / (1 - p_head), "tcn": tcn,
            "fusion": torch.from_numpy(g["mask_fusion"]).float() / (1 - p)}


def oracle_train_steps(sd, mods, steps, batch, length, hw, seed, backbone_train):
    """Restates trainer.py:365-391 + instantiators.py:74-79 on the oracle (dropout off)."""
    from oracle.lfan import cross_entropy_mean, lfan_forward, sgd_nesterov_step
    alias = synth.lfan_spec(mods)[1]
    names = trainable_names(sd, alias)
    osd = {k: v.clone() for k, v in sd.items()}
    bufs = [None] * len(names)
    out = []
    for step in range(steps):
        xs, ls = synth.make_clip_batch(mods, batch, length, hw=hw, seed=seed + step)
        params = [osd[n].clone().requires_grad_(True) for n in names]
        sds = dict(osd)
        sds.update(zip(names, params))
        for a, s in alias.items():
            sds[a] = sds[s]
        nb = {}
        logits = lfan_forward(xs, sds, mods, train=True, backbone_train=backbone_train, new_buffers=nb)
        loss = cross_entropy_mean(logits, ls)
        grads = torch.autograd.grad(loss, params)
        newp, bufs = sgd_nesterov_step([p.detach() for p in params], list(grads), bufs)
        for n, p in zip(names, newp):
            osd[n] = p
        for k, v in nb.items():
            osd[k] = v.detach()
        for a, s in alias.items():
            osd[a] = osd[s]
        out.append({"loss": loss.item(), "logits": logits.detach(), "grads": dict(zip(names, grads))})
    return out, osd, names


def trainable_names(sd, alias):
    """named_parameters() with requires_grad in the reference: everything except the frozen
    visual encoder, buffers and the duplicate net.N aliases (models/model.py:432-433)."""
    return [k for k in sd if not k.startswith("spatial.") and k not in alias
            and not k.endswith(("running_mean", "running_var", "num_batches_tracked"))]
