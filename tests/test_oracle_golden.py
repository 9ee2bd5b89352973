"""The CPU oracle against the fixtures recorded from the reference's own model code
(tools/gen_golden.py).  Runs without a GPU and without /root/reference."""
import numpy as np
import torch

import oracle
from feature_vs_text_compound_emotion_amd import synth
from oracle.lfan import lfan_forward

from helpers import MODS, golden, masks_from_golden, oracle_train_steps

TOL = 2e-5  # fp32 CPU vs fp32 CPU (same machine class); the generator saw <= 4e-6


def _setup(g):
    b, l, hw, ncls, wseed, dseed = [int(v) for v in g["meta"]]
    sd = synth.lfan_state_dict(MODS, n_cls=ncls, head_hw=hw // 8, seed=wseed)
    x, labels = synth.make_clip_batch(MODS, b, l, hw=hw, seed=dseed)
    return sd, x, labels, (b, l, hw)


def test_eval_forward_matches_reference():
    g = golden("lfan_trimodal_eval.npz")
    sd, x, _, (b, l, hw) = _setup(g)
    with torch.no_grad():
        emb = oracle.ir50_forward(x["video"].reshape(-1, 3, hw, hw), sd, "spatial.visual.backbone.")
        logits = lfan_forward(x, sd, MODS)
    assert np.abs(emb.numpy() - g["emb"]).max() < TOL
    assert np.abs(logits.numpy() - g["logits"]).max() < TOL
    assert np.abs(np.linalg.norm(emb.numpy(), axis=1) - 1).max() < 1e-5


def test_modality_subsets_and_order():
    g = golden("lfan_modal_subsets_eval.npz")
    b, l, hw, ncls, wseed, dseed = [int(v) for v in g["meta"]]
    for mods in (["video"], ["video", "vggish"], ["vggish", "video"], ["bert", "vggish"]):
        sd = synth.lfan_state_dict(mods, n_cls=ncls, head_hw=hw // 8, seed=wseed)
        x, _ = synth.make_clip_batch(mods, b, l, hw=hw, seed=dseed)
        with torch.no_grad():
            logits = lfan_forward(x, sd, mods)
        assert np.abs(logits.numpy() - g["logits_" + "_".join(mods)]).max() < TOL


def test_train_forward_with_reference_dropout_masks():
    g = golden("lfan_trimodal_train_fwd.npz")
    sd, x, _, _ = _setup(g)
    nb = {}
    with torch.no_grad():
        logits = lfan_forward(x, sd, MODS, train=True, masks=masks_from_golden(g), new_buffers=nb)
    assert np.abs(logits.numpy() - g["logits"]).max() < 5e-5
    assert np.abs(nb["bn.video.running_mean"].numpy() - g["bn_video_running_mean"]).max() < 1e-5
    assert np.abs(nb["bn.video.running_var"].numpy() - g["bn_video_running_var"]).max() < 1e-5
    assert np.abs(nb["spatial.visual.backbone.input_layer.1.running_mean"].numpy() - g["stem_running_mean"]).max() < 1e-5


def test_two_optimisation_steps_match_reference():
    for tag, backbone_train in (("refmode", True), ("evalbackbone", False)):
        g = golden(f"lfan_trimodal_train_steps_{tag}.npz")
        sd, _, _, (b, l, hw) = _setup(g)
        steps, osd, names = oracle_train_steps(sd, MODS, 2, b, l, hw, int(g["meta"][5]), backbone_train)
        assert list(g["names"]) == names
        for s, rec in enumerate(steps):
            assert abs(rec["loss"] - float(g[f"loss{s}"])) < 1e-5
            assert np.abs(rec["logits"].numpy() - g[f"logits{s}"]).max() < 5e-5
            gn = np.array([rec["grads"][n].norm().item() for n in names])
            assert np.abs(gn - g[f"gradnorm{s}"]).max() < 1e-4
            for key in g.files:
                if key.startswith(f"grad{s}:"):
                    assert np.abs(rec["grads"][key.split(":", 1)[1]].numpy() - g[key]).max() < 1e-5
        for key in g.files:
            if key.startswith("param2:"):
                assert np.abs(osd[key.split(":", 1)[1]].numpy() - g[key]).max() < 1e-6
        assert np.abs(osd["bn.video.running_mean"].numpy() - g["bn_video_running_mean2"]).max() < 1e-5


def test_visual_backbone_alone():
    g = golden("visual_backbone_eval.npz")
    n, hw, wseed, dseed = [int(v) for v in g["meta"]]
    vsd = synth.make_state_dict(synth.visual_backbone_spec("", hw // 8), seed=wseed)
    frames = torch.randn(n, 3, hw, hw, generator=torch.Generator().manual_seed(dseed))
    with torch.no_grad():
        emb, feat = oracle.ir50_forward(frames, vsd, "backbone.", return_features=True)
    assert np.abs(emb.numpy() - g["emb"]).max() < TOL
    assert np.abs(feat.mean((2, 3)).numpy() - g["feat_mean"]).max() < 1e-4


def test_logmel_front_end_matches_reference():
    g = golden("logmel_examples.npz")
    sr, fps, seed = [int(v) for v in g["meta"]]
    pcm = synth.make_audio_int16(1.0, sr, seed=seed)
    ex = oracle.wav_int16_to_examples(pcm.numpy(), sr, 0.96, 1.0 / fps)
    assert ex.shape == (33, 96, 64)
    assert np.abs(ex.astype(np.float32) - g["examples"]).max() == 0.0
    assert np.abs(oracle.log_mel_spectrogram(np.pad(pcm.numpy() / 32768.0, (0, sr), "edge")) - g["log_mel"]).max() < 1e-12
    from oracle.vggish import example_starts
    assert example_starts(198, 96, 2.5) == list(g["starts_hop25"])  # round-half-to-even: 0, 2, 5, 8, 10 ...


def test_vggish_matches_reference():
    g = golden("vggish_eval.npz")
    n, wseed = [int(v) for v in g["meta"]]
    vsd = synth.make_state_dict(synth.vggish_spec(""), seed=wseed)
    x = golden("logmel_examples.npz")["examples"][:n]
    with torch.no_grad():
        emb = oracle.vggish_forward(x, vsd)
    assert np.abs(emb.numpy() - g["emb"]).max() < 1e-4


def test_bert_features_match_transformers():
    g = golden("bert_eval.npz")
    wseed, s1, s2 = [int(v) for v in g["meta"]]
    bsd = synth.make_state_dict(synth.bert_spec(""), seed=wseed)
    ids, mask = synth.make_token_ids(3, 24, seed=s1, pad_from=[24, 17, 9])
    with torch.no_grad():
        tok = oracle.bert_token_features(ids, mask, bsd)
    assert np.array_equal(mask.numpy(), g["mask"])
    assert np.abs(tok.numpy()[:, :, ::8] - g["tok_sum"])[mask.bool().numpy()].max() < 2e-4
    ids2, mask2 = synth.make_token_ids(2, 24, seed=s2, pad_from=[20, 12])
    with torch.no_grad():
        feats = oracle.exclude_padding(oracle.bert_token_features(ids2, mask2, bsd), mask2)
    assert np.abs(feats.numpy()[:, ::8] - g["feats_excl"]).max() < 2e-4


def test_can_jmt_mt_heads_match_reference():
    from oracle.jmt import can_forward, jmt_forward
    g = golden("heads_can_jmt_mt.npz")
    b, l, hw, ncls, wseed, dseed = [int(v) for v in g["meta"]]
    mods = ["video", "vggish"]
    x, _ = synth.make_clip_batch(mods, b, l, hw=hw, seed=dseed)
    for name in ("JMT", "MT", "CAN"):
        spec, alias = synth.can_spec(mods) if name == "CAN" else synth.jmt_spec(mods, name)
        sd = synth.make_state_dict(spec, alias, seed=wseed)
        with torch.no_grad():
            out = can_forward(x, sd, mods) if name == "CAN" else jmt_forward(x, sd, mods, model_name=name)
        assert np.abs(out.numpy() - g[f"{name}_eval_logits"]).max() < TOL


def test_lfan_with_the_on_model_vggish_matches_reference():
    """'logmel' modality: VGGish inside forward (models/model.py:458-461,500-508); tools/gen_golden_logmel.py."""
    g = golden("lfan_logmel.npz")
    b, l, ncls, wseed, dseed = [int(v) for v in g["meta"]]
    mods = ["logmel", "vggish"]
    spec, alias = synth.lfan_spec(mods, n_cls=ncls)
    sd = synth.make_state_dict(spec, alias, seed=wseed)
    x, _ = synth.make_clip_batch(mods, b, l, seed=dseed)
    with torch.no_grad():
        out = lfan_forward(x, sd, mods)
    assert np.abs(out.numpy() - g["logits"]).max() < TOL
