"""Golden fixtures for the CAN / JMT / MT heads from the reference's own classes.

    PYTHONPATH=/root/reference PYTHONDONTWRITEBYTECODE=1 python tools/gen_golden_heads.py

Eval logits and one optimisation step (loss, logits, gradient norms, a few gradients) per head with
dropout off and model.train() exactly as the reference runs it.  Checked against the oracle first.
"""
import os
import sys
import tempfile

sys.modules["triton"] = None
import numpy as np  # noqa: E402
import torch  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(1, "/root/reference")

from feature_vs_text_compound_emotion_amd import synth  # noqa: E402
from oracle.jmt import can_forward, jmt_forward  # noqa: E402
from oracle.lfan import cross_entropy_mean  # noqa: E402

from models.model import CAN, JMT  # noqa: E402  (reference)

OUT = os.path.join(ROOT, "tests", "golden")
MODS = ["video", "vggish"]


def build(name, sd):
    d = tempfile.mkdtemp()
    vb = {k[len("spatial.visual."):]: v for k, v in sd.items() if k.startswith("spatial.visual.")}
    torch.save(vb, os.path.join(d, "res50_ir_0.887.pth"))
    bs = {"visual_state_dict": "res50_ir_0.887", "audio_state_dict": "vggish"}
    if name == "CAN":
        m = CAN(task="CLASSIFICATION", modalities=MODS, tcn_settings=synth.TCN_SETTINGS, backbone_settings=bs,
                output_dim=7, root_dir=d, device="cpu")
    else:
        m = JMT(task="CLASSIFICATION", modalities=MODS, tcn_settings=synth.TCN_SETTINGS, backbone_settings=bs,
                output_dim=7, root_dir=d, device="cpu", model_name=name)
    res = m.load_state_dict(sd, strict=True)
    assert not res.missing_keys and not res.unexpected_keys
    return m


def main():
    torch.set_num_threads(8)
    B, L, HW = 3, 8, 40
    rec = {}
    for name in ("JMT", "MT", "CAN"):
        spec, alias = synth.can_spec(MODS) if name == "CAN" else synth.jmt_spec(MODS, name)
        sd = synth.make_state_dict(spec, alias, seed=50)
        fwd = (lambda x, s, **kw: can_forward(x, s, MODS, **kw)) if name == "CAN" else \
              (lambda x, s, **kw: jmt_forward(x, s, MODS, model_name=name, **kw))
        x, labels = synth.make_clip_batch(MODS, B, L, hw=HW, seed=60)
        ref = build(name, sd)
        ref.eval()
        with torch.no_grad():
            lg = ref({k: v.clone() for k, v in x.items()})
            og = fwd(x, sd)
        print(name, "eval logits oracle-vs-reference", (lg - og).abs().max().item())
        assert (lg - og).abs().max().item() < 2e-5
        rec[f"{name}_eval_logits"] = lg.numpy()
        # one optimisation step, dropout off
        ref = build(name, sd)
        ref.train()
        for mod in ref.modules():
            if isinstance(mod, torch.nn.Dropout):
                mod.p = 0.0
        names = [n for n, p in ref.named_parameters() if p.requires_grad]
        params = [p for n, p in ref.named_parameters() if p.requires_grad]
        out = ref({k: v.clone() for k, v in x.items()})
        loss = torch.nn.functional.cross_entropy(out.reshape(B * L, 7), labels.reshape(B * L).long())
        loss.backward()
        oparams = [sd[n].clone().requires_grad_(True) for n in names]
        osd = dict(sd)
        osd.update(zip(names, oparams))
        for a, s in alias.items():
            osd[a] = osd[s]
        oout = fwd(x, osd, train=True)
        oloss = cross_entropy_mean(oout, labels)
        ograds = torch.autograd.grad(oloss, oparams, allow_unused=True)
        worst = 0.0
        for n, p, g in zip(names, params, ograds):
            if p.grad is None:
                assert g is None or g.abs().max() == 0, n
                continue
            worst = max(worst, (p.grad - g).abs().max().item() / max(1.0, p.grad.abs().max().item()))
        print(name, f"train loss ref {loss.item():.6f} oracle {oloss.item():.6f} max grad diff {worst:.2e}",
              len(names), "trainable tensors")
        assert abs(loss.item() - oloss.item()) < 1e-5 and worst < 5e-5  # relative to max(1, |grad|_inf): fused vs explicit MHA
        rec[f"{name}_train_loss"] = np.array(loss.item())
        rec[f"{name}_train_logits"] = out.detach().numpy()
        rec[f"{name}_names"] = np.array(names)
        rec[f"{name}_gradnorm"] = np.array([0.0 if p.grad is None else p.grad.norm().item() for p in params])
        for n in ("fc2.weight", "fuse.final_self_attention.in_proj_weight", "fuse.CA_va.out_proj.weight",
                  "fuse.visual_encoder.layers.0.feed_forward.0.weight", "fuse.augment_audio_feats_dim.weight",
                  "fuse.attn.1.weight", "fuse.weights.weight", "temporal.vggish.network.0.conv1.weight_v", "bn.video.weight"):
            if n in names and dict(zip(names, params))[n].grad is not None:
                rec[f"{name}_grad:{n}"] = dict(zip(names, params))[n].grad.numpy()
    np.savez_compressed(os.path.join(OUT, "heads_can_jmt_mt.npz"), **rec, meta=np.array([B, L, HW, 7, 50, 60]))
    print(os.path.getsize(os.path.join(OUT, "heads_can_jmt_mt.npz")))


if __name__ == "__main__":
    main()
