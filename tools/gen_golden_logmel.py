"""Fixture generator (build container only: imports /root/reference read-only): the reference's LFAN with the 'logmel'
modality -- the one place it runs VGGish INSIDE forward (models/model.py:458-461,500-508) -- on seeded synthetic weights and
inputs.  Checks oracle.lfan_forward against it and writes tests/golden/lfan_logmel.npz (logits + the seeds; weights and
inputs are re-drawn from the seeds by the tests).

    python tools/gen_golden_logmel.py
"""
import os
import sys

import numpy as np
import torch

sys.modules.setdefault("triton", None)
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, "/root/reference")
sys.path.insert(0, ROOT)

import models.model as ref_model  # noqa: E402
import oracle  # noqa: E402
from feature_vs_text_compound_emotion_amd import synth  # noqa: E402

MODS, B, L, N_CLS, WSEED, DSEED = ["logmel", "vggish"], 2, 6, 7, 3, 9


def main():
    spec, alias = synth.lfan_spec(MODS, n_cls=N_CLS)
    sd = synth.make_state_dict(spec, alias, seed=WSEED)
    # init() reads <root_dir>/<audio_state_dict>.pth; the synthetic weights arrive through load_state_dict instead
    ref_model.LFAN.load_audio_backbone = lambda self, backbone_settings: ref_model.AudioBackbone()
    m = ref_model.LFAN(backbone_settings={}, output_dim=N_CLS, task="CLASSIFICATION", modality=MODS, example_length=L,
                       kernel_size=5, tcn_channel=synth.TCN_CHANNELS, root_dir="", device="cpu")
    m.init()
    m.load_state_dict(sd, strict=True)
    m.eval()
    x, _ = synth.make_clip_batch(MODS, B, L, seed=DSEED)
    with torch.no_grad():
        caller = {k: v.clone() for k, v in x.items()}
        ref = m(caller)
        # the caller's dict afterwards (model.py:511-515): per-modality features [B, L, C_m]
        left = {k: v.numpy() for k, v in caller.items()}
        # another key order gives the same logits (the fusion walks the model's modality list)
        ref_swapped = m({k: x[k].clone() for k in reversed(MODS)})
        out = oracle.lfan_forward(x, sd, MODS)
    assert (ref - ref_swapped).abs().max().item() == 0.0
    err = (out - ref).abs().max().item()
    print(f"oracle vs reference LFAN(logmel, vggish): {err:.2e}")
    assert err < 5e-6
    np.savez(os.path.join(ROOT, "tests", "golden", "lfan_logmel.npz"), logits=ref.numpy(),
             meta=np.asarray([B, L, N_CLS, WSEED, DSEED]), left_logmel=left["logmel"], left_vggish=left["vggish"])


if __name__ == "__main__":
    main()
