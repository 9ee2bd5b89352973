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


def run(groups, fname, n=6):
    torch.set_num_threads(8)
    hw, wseed, dseed = 40, 21, 77
    vsd = synth.make_state_dict(synth.visual_backbone_spec("", hw // 8), seed=wseed)
    g = torch.Generator().manual_seed(dseed)
    frames = torch.randn(n, 3, hw, hw, generator=g)
    G = torch.randn(n, 512, generator=g)
    vb = VisualBackbone(use_pretrained=False)
    vb.load_state_dict(vsd, strict=True)
    params = list(vb.parameters())
    for p in params:
        p.requires_grad = False
    idx = [i for a, b in groups for i in range(a, b)]  # ResnetParamControl.init_param_group: np.arange(*group)
    released = [params[i] for i in idx]
    all_names = [k for k, _ in vb.named_parameters()]
    names = [all_names[i] for i in idx]
    assert names[:6] == ["backbone.output_layer.0.weight", "backbone.output_layer.0.bias", "backbone.output_layer.3.weight",
                         "backbone.output_layer.3.bias", "backbone.output_layer.4.weight", "backbone.output_layer.4.bias"], names
    if len(groups) > 1:
        assert names[6].startswith("backbone.body.21.") and names[-1] == "backbone.body.23.res_layer.4.bias", (names[6], names[-1])
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
        if name.startswith("backbone.output_layer."):
            key = name.replace("backbone.output_layer.", "g")
            out[key] = gr if gr.size < 20000 else gr[:8].copy()  # the FC weight gradient: first 8 rows + its norm
        else:
            key = "grad:" + name[len("backbone."):]
            out[key] = gr if gr.size < 20000 else gr.reshape(-1)[:4096].copy()  # conv weights: first 4096 values + norm
        out[key + "_norm"] = np.array([np.linalg.norm(gr.astype(np.float64))])
    stats = ["backbone.output_layer.0.running_mean", "backbone.output_layer.0.running_var",
             "backbone.output_layer.4.running_mean", "backbone.output_layer.4.running_var"]
    if len(groups) > 1:
        stats += ["backbone.body.21.res_layer.0.running_var", "backbone.body.21.shortcut_layer.1.running_mean",
                  "backbone.body.23.res_layer.4.running_var"]
    for k in stats:
        out["after_" + k.replace("backbone.output_layer.", "").replace("backbone.", "")] = sd_after[k].numpy()
    np.savez_compressed(os.path.join(OUT, fname), **out)
    print(fname, len(out), "arrays, size", os.path.getsize(os.path.join(OUT, fname)))


if __name__ == "__main__":
    run([(4, 10)], "head_release_step.npz")               # module_dict["visual"][0]
    run([(4, 10), (163, 187)], "body_release_step.npz")   # + module_dict["visual"][1]: stage 4 of the IR-50
    # the same release on 32 frames: batch statistics over 800 instead of 150 values per channel in stage 4, so that the
    # comparison measures the kernels and not the conditioning of a 6-frame BatchNorm (round-1 verdict, tolerance hygiene)
    run([(4, 10), (163, 187)], "body_release_step_n32.npz", n=32)
