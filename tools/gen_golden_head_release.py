"""Golden fixture for the first group of the reference's gradual release (base/parameter_control.py:55-103: parameters
4..9 of the visual encoder = the output layer), produced by the REFERENCE's VisualBackbone in train mode.

    PYTHONPATH=/root/reference PYTHONDONTWRITEBYTECODE=1 python tools/gen_golden_head_release.py
"""
import os
import sys

sys.modules["triton"] = None
import numpy as np  # noqa: E402
import torch  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(1, "/root/reference")

from feature_vs_text_compound_emotion_amd import synth  # noqa: E402
from models.backbone import VisualBackbone  # noqa: E402  (reference)

OUT = os.path.join(ROOT, "tests", "golden")


def main():
    torch.set_num_threads(8)
    n, hw, wseed, dseed = 6, 40, 21, 77
    vsd = synth.make_state_dict(synth.visual_backbone_spec("", hw // 8), seed=wseed)
    g = torch.Generator().manual_seed(dseed)
    frames = torch.randn(n, 3, hw, hw, generator=g)
    G = torch.randn(n, 512, generator=g)
    vb = VisualBackbone(use_pretrained=False)
    vb.load_state_dict(vsd, strict=True)
    params = list(vb.parameters())
    for p in params:
        p.requires_grad = False
    released = params[4:10]  # the reference's first visual group: np.arange(4, 10)
    names = [k for k, _ in vb.named_parameters()][4:10]
    assert names == ["backbone.output_layer.0.weight", "backbone.output_layer.0.bias", "backbone.output_layer.3.weight",
                     "backbone.output_layer.3.bias", "backbone.output_layer.4.weight", "backbone.output_layer.4.bias"], names
    for p in released:
        p.requires_grad = True
    vb.train()
    masks = {}

    def hook(mod, inp, out):
        x = inp[0]
        masks["m"] = torch.where(x != 0, out / torch.where(x != 0, x, torch.ones_like(x)),
                                 torch.full_like(x, 1.0 / (1.0 - mod.p))).detach().clone()
    h = vb.backbone.output_layer[1].register_forward_hook(hook)
    emb = vb(frames)
    h.remove()
    (emb * G).sum().backward()
    keep = (masks["m"] > 0).numpy().astype(np.uint8)  # [n,512,5,5]; scale 1/(1-0.4)
    sd_after = vb.state_dict()
    out = {"meta": np.array([n, hw, wseed, dseed]), "emb": emb.detach().numpy(), "keep": keep}
    for name, p in zip(names, released):
        gr = p.grad.numpy()
        key = name.replace("backbone.output_layer.", "g")
        out[key + "_norm"] = np.array([np.linalg.norm(gr.astype(np.float64))])
        out[key] = gr if gr.size < 20000 else gr[:8].copy()  # the FC weight gradient: first 8 rows + its norm
    for k in ("backbone.output_layer.0.running_mean", "backbone.output_layer.0.running_var",
              "backbone.output_layer.4.running_mean", "backbone.output_layer.4.running_var"):
        out["after_" + k.replace("backbone.output_layer.", "")] = sd_after[k].numpy()
    np.savez_compressed(os.path.join(OUT, "head_release_step.npz"), **out)
    print({k: (v.shape if hasattr(v, "shape") else v) for k, v in out.items()})
    print("size", os.path.getsize(os.path.join(OUT, "head_release_step.npz")))


if __name__ == "__main__":
    main()
