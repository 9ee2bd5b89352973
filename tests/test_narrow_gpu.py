"""Narrow-storage encoder mode (precision = "bf16" / "fp16": one 16-bit plane per tensor, one MFMA per product, fp32
accumulate) vs the fp32 oracle: BASELINE cfg5 ("bf16 storage / fp32 accumulate", 64-frame clips, C-EXPR-DB classes) and the
arithmetic of the reference's own --amp recipe (fp16 autocast, trainer.py:341,367).

Bars.  SURVEY section 7 sets cfg5's bar: |logit error| <= 2e-2 and argmax agreement; north_star's 1e-3 is the fp32-parity bar
(met by precision = "bf16x3" / "fp32", tests/test_backbone_gpu.py, tests/test_tail_gpu.py).  The tests below assert the
cfg5 bar for both storage types and, for fp16, additionally a 4e-3 bound (measured ~1e-3: its 11-bit mantissa sits right
at north_star's fp32 bar; bf16's 8-bit mantissa measures ~6e-3 on logits).  Measured values are printed (pytest -s) and
recorded in DESIGN.md.
"""
import pytest
import torch

pytestmark = pytest.mark.gpu

from helpers import MODS  # noqa: E402

DT = {"bf16": torch.bfloat16, "fp16": torch.float16}
EMB_BAR = {"bf16": 2.5e-2, "fp16": 4e-3}     # unit-norm 512-d embeddings (elements ~0.044)
LOGIT_BAR = {"bf16": 2e-2, "fp16": 4e-3}


def _vb(sd, head_hw, precision):
    from feature_vs_text_compound_emotion_amd.visual_backbone import VisualBackbone
    vb = VisualBackbone(use_pretrained=False, head_hw=head_hw)
    vb.load_state_dict(sd, strict=True)
    vb.backbone.precision = precision
    return vb.cuda().eval()


@pytest.mark.parametrize("precision", ["bf16", "fp16"])
@pytest.mark.parametrize("n,hw", [(1, 40), (37, 40), (2, 224), (3, 64)])
def test_narrow_embedding_vs_oracle(n, hw, precision):
    import oracle
    from feature_vs_text_compound_emotion_amd import synth
    vsd = synth.make_state_dict(synth.visual_backbone_spec("", hw // 8), seed=11)
    frames = torch.randn(n, 3, hw, hw, generator=torch.Generator().manual_seed(n + hw))
    with torch.no_grad():
        ref = oracle.ir50_forward(frames, vsd, "backbone.")
        emb = _vb(vsd, hw // 8, precision)(frames.cuda()).cpu()
    err = (emb - ref).abs().max().item()
    cos = torch.nn.functional.cosine_similarity(emb, ref, dim=1).min().item()
    print(f"\n[narrow {precision}] IR-50 eval n={n} hw={hw}: max |emb err| {err:.2e}, min cosine {cos:.6f}")
    assert err < EMB_BAR[precision]
    assert cos > (0.995 if precision == "bf16" else 0.9999)
    assert (emb.norm(dim=1) - 1).abs().max().item() < 1e-5


@pytest.mark.parametrize("precision", ["bf16", "fp16"])
def test_narrow_train_mode_batch_statistics_vs_oracle(precision):
    """model.train() semantics (batch-statistics BatchNorm in all 54 layers, running buffers updated, injected Dropout mask)."""
    import oracle
    from feature_vs_text_compound_emotion_amd import synth
    n, hw = 6, 40
    vsd = synth.make_state_dict(synth.visual_backbone_spec("", hw // 8), seed=13)
    g = torch.Generator().manual_seed(n * hw)
    frames = torch.randn(n, 3, hw, hw, generator=g)
    mask = synth.dropout_mask((n, 512, hw // 8, hw // 8), 0.4, g)
    nb = {}
    with torch.no_grad():
        ref = oracle.ir50_forward(frames, vsd, "backbone.", train=True, head_dropout_mask=mask, new_buffers=nb)
    vb = _vb(vsd, hw // 8, precision).train()
    with torch.no_grad():
        emb = vb(frames.cuda(), mask.permute(0, 2, 3, 1).contiguous().cuda()).cpu()
    err = (emb - ref).abs().max().item()
    sd_after = vb.state_dict()
    worst = max(((sd_after[k].cpu() - v).abs() / v.abs().clamp_min(1.0)).max().item() for k, v in nb.items())
    print(f"\n[narrow {precision}] IR-50 train-mode n={n}: max |emb err| {err:.2e}, worst running-stat rel err {worst:.2e}")
    # 6 frames: the batch statistics of the deep layers are over 150 values, which amplifies the storage rounding
    assert err < 2 * EMB_BAR[precision]
    assert worst < (5e-2 if precision == "bf16" else 6e-3)
    assert int(sd_after["backbone.input_layer.1.num_batches_tracked"]) == 1


def _lfan(sd, n_cls, length, hw, precision):
    from feature_vs_text_compound_emotion_amd import synth
    from feature_vs_text_compound_emotion_amd.lfan import LFAN
    model = LFAN(backbone_settings={}, output_dim=n_cls, task="CLASSIFICATION", modality=MODS, example_length=length,
                 kernel_size=5, tcn_channel=synth.TCN_CHANNELS, root_dir="", device="cuda", head_hw=hw // 8)
    model.init(load_backbone=False)
    model.load_state_dict(sd, strict=True)
    model.spatial["visual"].backbone.precision = precision
    return model.cuda()


@pytest.mark.parametrize("precision", ["bf16", "fp16"])
def test_cfg5_trimodal_64_frame_clips_8_classes(precision):
    """BASELINE cfg5: tri-modal LFAN (video + vggish + bert), 64-frame clips, C-EXPR-DB's 7 classes + 'Other'
    (experiment.py:55-57), narrow storage in the encoder.  Eval forward and the reference's model.train() forward
    (batch statistics in the frozen encoder) vs the fp32 oracle: cfg5's bar |logit err| <= 2e-2 + argmax agreement."""
    from feature_vs_text_compound_emotion_amd import synth
    from feature_vs_text_compound_emotion_amd.lfan import cross_entropy_loss
    from oracle.lfan import cross_entropy_mean, lfan_forward
    b, length, hw, n_cls = 2, 64, 40, 8
    sd = synth.lfan_state_dict(MODS, n_cls=n_cls, head_hw=hw // 8, seed=0)
    x, labels = synth.make_clip_batch(MODS, b, length, hw=hw, seed=4321, n_cls=n_cls)
    model = _lfan(sd, n_cls, length, hw, precision).eval()
    xd = {k: v.cuda() for k, v in x.items()}
    with torch.no_grad():
        logits = model(dict(xd)).cpu()
        ref = lfan_forward(x, sd, MODS, train=False)
    assert logits.shape == (b, length, n_cls)
    err = (logits - ref).abs().max().item()
    agree = (logits.argmax(-1) == ref.argmax(-1)).float().mean().item()
    # frames whose top-2 reference logits are closer than twice the error bar may legitimately flip
    top2 = ref.topk(2, dim=-1).values
    decided = (top2[..., 0] - top2[..., 1]) > 2 * LOGIT_BAR[precision]
    agree_decided = (logits.argmax(-1) == ref.argmax(-1))[decided].float().mean().item() if decided.any() else 1.0
    print(f"\n[narrow {precision}] cfg5 eval: max |logit err| {err:.2e}, argmax agreement {agree:.4f} "
          f"({agree_decided:.4f} on the {int(decided.sum())} frames with a margin > {2 * LOGIT_BAR[precision]:g})")
    assert err < LOGIT_BAR[precision]
    assert agree_decided == 1.0 and agree > 0.97
    # train-mode forward + loss, dropout off (no mask plumbing needed), encoder BatchNorms on batch statistics
    model.train()
    for mod in model.modules():
        if isinstance(mod, torch.nn.Dropout):
            mod.p = 0.0
    for net in model.temporal.values():
        net.dropout = 0.0
    out = model(dict(xd))
    loss = cross_entropy_loss(out, labels.cuda())
    loss.backward()
    oref = lfan_forward(x, sd, MODS, train=True, backbone_train=True)
    oloss = cross_entropy_mean(oref, labels)
    terr = (out.detach().cpu() - oref).abs().max().item()
    print(f"[narrow {precision}] cfg5 train forward: max |logit err| {terr:.2e}, loss {loss.item():.6f} vs oracle {oloss.item():.6f}")
    assert terr < 2 * LOGIT_BAR[precision]
    assert abs(loss.item() - oloss.item()) < LOGIT_BAR[precision]
    assert model.regressor.weight.grad is not None and torch.isfinite(model.regressor.weight.grad).all()
