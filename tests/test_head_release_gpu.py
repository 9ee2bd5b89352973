"""First group of the reference's gradual release (base/parameter_control.py:55-103): the encoder's output layer trained on
top of the frozen body -- forward, gradients and running statistics vs a fixture recorded from the reference's own
VisualBackbone (tools/gen_golden_head_release.py)."""
import numpy as np
import pytest
import torch

from helpers import golden

pytestmark = pytest.mark.gpu

_ORACLE_CACHE = {}


def _setup():
    from feature_vs_text_compound_emotion_amd import synth
    from feature_vs_text_compound_emotion_amd.visual_backbone import VisualBackbone
    g = golden("head_release_step.npz")
    n, hw, wseed, dseed = [int(v) for v in g["meta"]]
    vsd = synth.make_state_dict(synth.visual_backbone_spec("", hw // 8), seed=wseed)
    gen = torch.Generator().manual_seed(dseed)
    frames = torch.randn(n, 3, hw, hw, generator=gen)
    G = torch.randn(n, 512, generator=gen)
    vb = VisualBackbone(use_pretrained=False, head_hw=hw // 8)
    vb.load_state_dict(vsd, strict=True)
    return g, vb.cuda(), frames, G


def test_released_head_matches_reference_step():
    g, vb, frames, G = _setup()
    params = list(vb.parameters())
    for p in params:
        p.requires_grad = False
    for p in params[4:10]:  # ResnetParamControl's first visual group
        p.requires_grad = True
    vb.train()
    mask = torch.from_numpy(g["keep"]).float().div(1 - 0.4).permute(0, 2, 3, 1).contiguous().cuda()
    emb = vb(frames.cuda(), mask)
    assert emb.requires_grad
    assert np.abs(emb.detach().cpu().numpy() - g["emb"]).max() < 1e-4
    (emb * G.cuda()).sum().backward()
    ol = vb.backbone.output_layer
    for key, p in (("g0.weight", ol[0].weight), ("g0.bias", ol[0].bias), ("g3.bias", ol[3].bias), ("g4.weight", ol[4].weight),
                   ("g4.bias", ol[4].bias)):
        ref = g[key]
        assert np.abs(p.grad.cpu().numpy() - ref).max() < 5e-5 * max(1.0, np.abs(ref).max()), key
    wg = ol[3].weight.grad.cpu().numpy()
    assert wg.shape == (512, 12800)
    # dW = de^T . hfeat carries the body's own error in hfeat (bf16x3 convs through 24 units of batch-statistics BatchNorm
    # over only 6 frames: ~1e-4 absolute on O(1) features), hence the wider bar than for the per-channel gradients
    assert np.abs(wg[:8] - g["g3.weight"]).max() < 2e-4 * max(1.0, np.abs(g["g3.weight"]).max())
    assert abs(np.linalg.norm(wg.astype(np.float64)) - g["g3.weight_norm"][0]) < 1e-4 * g["g3.weight_norm"][0]
    sd = vb.state_dict()
    for k in ("0.running_mean", "0.running_var", "4.running_mean", "4.running_var"):
        ref = g["after_" + k]
        got = sd["backbone.output_layer." + k].cpu().numpy()
        assert (np.abs(got - ref) / np.maximum(np.abs(ref), 1.0)).max() < 2e-5, k
    assert all(p.grad is None for p in params[:4] + params[10:])


def test_release_of_a_non_suffix_fails_loudly():
    g, vb, frames, G = _setup()
    for p in vb.parameters():
        p.requires_grad = False
    for p in list(vb.backbone.output_layer.parameters()) + list(vb.backbone.body[5].parameters()):
        p.requires_grad = True         # a unit in the middle of the body: gradients would have to cross frozen units above it
    vb.train()
    with pytest.raises(NotImplementedError, match="suffix of the body"):
        vb(frames.cuda())
    for p in vb.backbone.body[5].parameters():
        p.requires_grad = False
    for p in vb.backbone.input_layer.parameters():
        p.requires_grad = True         # the stem without the body above it
    with pytest.raises(NotImplementedError, match="together with the whole body"):
        vb(frames.cuda())


def test_l2norm_backward_kernel():
    from feature_vs_text_compound_emotion_amd import ops
    gen = torch.Generator().manual_seed(3)
    x = torch.randn(37, 512, generator=gen, requires_grad=True)
    dy = torch.randn(37, 512, generator=gen)
    (x / torch.norm(x, 2, 1, True)).backward(dy)
    dx = ops.l2norm_rows_bwd(dy.cuda(), x.detach().cuda())
    assert (dx.cpu() - x.grad).abs().max().item() < 1e-6


@pytest.mark.parametrize("precision,fixture,tol", [("fp32", "body_release_step.npz", 5e-5), ("bf16x3", "body_release_step.npz", 2e-3),
                                                   ("fp32", "body_release_step_n32.npz", None),
                                                   ("bf16x3", "body_release_step_n32.npz", None),
                                                   ("fp16", "body_release_step_n32.npz", None)])
def test_released_stage4_matches_reference_step(precision, fixture, tol):
    """Second release group (base/parameter_control.py: parameters 163..186 = stage 4, units 21-23) on top of the head:
    frozen units run on the bf16x3 kernels, released units on the fp32 kernels with their backward.  Two fixtures from the
    reference's own VisualBackbone.

    6 frames: every gradient ELEMENT within 5e-5 (fp32 frozen units) / 2e-3 (bf16x3 frozen units: the BatchNorm backward of
    the released units amplifies the frozen units' ~1e-5 feature error over 150 values per channel).

    32 frames (round-2 addition): elementwise max-norm is the wrong yardstick there, and not because of bf16x3 -- the
    exact-fp32 HIP path, which matches the 6-frame fixture to 3e-6, differs from the reference on single rows of
    body.23.res_layer.1.weight by 2e-2 while every kernel involved is exact at these shapes (checked one by one against
    float64) and the float64 oracle agrees with the reference to 1.4e-6: among the 409 600 PReLU pre-activations of a unit
    one lands within fp32 rounding of zero, its derivative flips between two correct fp32 evaluations, and that one pixel
    moves a whole 4 608-entry filter row (the slope gradient, a sum of x * dy with x ~ 0, stays exact -- which is how the
    flip was identified).  So the 32-frame fixture is held to per-parameter RELATIVE L2 bounds and exact gradient norms."""
    from feature_vs_text_compound_emotion_amd import synth
    from feature_vs_text_compound_emotion_amd.visual_backbone import VisualBackbone
    g = golden(fixture)
    n, hw, wseed, dseed = [int(v) for v in g["meta"]]
    vsd = synth.make_state_dict(synth.visual_backbone_spec("", hw // 8), seed=wseed)
    gen = torch.Generator().manual_seed(dseed)
    frames = torch.randn(n, 3, hw, hw, generator=gen)
    G = torch.randn(n, 512, generator=gen)
    vb = VisualBackbone(use_pretrained=False, head_hw=hw // 8)
    vb.load_state_dict(vsd, strict=True)
    vb = vb.cuda()
    vb.backbone.precision = precision  # frozen AND released units (forward, data and weight gradients) run in this mode
    params = list(vb.parameters())
    names = [k for k, _ in vb.named_parameters()]
    for p in params:
        p.requires_grad = False
    idx = list(range(4, 10)) + list(range(163, 187))
    for i in idx:
        params[i].requires_grad = True
    assert names[163].startswith("backbone.body.21.") and names[186] == "backbone.body.23.res_layer.4.bias"
    vb.train()
    mask = torch.from_numpy(g["keep"]).float().div(1 - 0.4).permute(0, 2, 3, 1).contiguous().cuda()
    emb = vb(frames.cuda(), mask)
    assert np.abs(emb.detach().cpu().numpy() - g["emb"]).max() < {"fp32": 2e-5, "bf16x3": 2e-4, "fp16": 4e-3}[precision]
    (emb * G.cuda()).sum().backward()
    worst = 0.0
    # fp16: one storage rounding (2^-11) per tensor through 24 units -- the relative-L2 bars below are 10x the bf16x3 ones
    f16 = 10.0 if precision == "fp16" else 1.0
    for i in idx:
        name = names[i]
        key = name.replace("backbone.output_layer.", "g") if "output_layer" in name else "grad:" + name[len("backbone."):]
        ref, got = g[key], params[i].grad.cpu().numpy()
        nrm = float(g[key + "_norm"][0])
        ntol = (1e-4 if precision == "fp32" else 1e-3) * (1 if tol is not None else 10) * f16   # 32 frames: the flipped derivative
        assert abs(np.linalg.norm(got.astype(np.float64)) - nrm) < ntol * max(nrm, 1e-2), name  # (floor: zero-by-construction gradients)
        part = got if got.size == ref.size else (got[:8] if "output_layer" in name else got.reshape(-1)[:4096])
        if tol is None:   # 32 frames: relative L2 of the compared part
            err = np.linalg.norm((part.reshape(ref.shape) - ref).astype(np.float64)) / max(np.linalg.norm(ref.astype(np.float64)), 1e-2)  # (floor: the FC bias in front of a train-mode BatchNorm has a mathematically zero gradient)
            assert err < (1e-2 if precision == "fp32" else 2e-2) * f16, (name, err)
        else:
            err = np.abs(part.reshape(ref.shape) - ref).max() / max(1.0, np.abs(ref).max())
            assert err < tol, (name, err)
        worst = max(worst, err)
    print(f"\n[release] {precision} {fixture}: worst gradient error ({'relative L2 per parameter' if tol is None else 'elementwise, relative to max(1, |ref|)'}) {worst:.2e}")
    sd = vb.state_dict()
    for k in ("body.21.res_layer.0.running_var", "body.21.shortcut_layer.1.running_mean", "body.23.res_layer.4.running_var",
              "0.running_mean", "4.running_var"):
        ref = g["after_" + k]
        full = "backbone." + k if k.startswith("body") else "backbone.output_layer." + k
        got = sd[full].cpu().numpy()
        assert (np.abs(got - ref) / np.maximum(np.abs(ref), 1.0)).max() < (5e-5 if precision != "fp16" else 5e-3), k
    assert all(params[i].grad is None for i in range(len(params)) if i not in idx)


def test_released_units_in_fp16_keep_unscaled_small_gradients():
    """ADVICE (round 2): with fp16 storage and NO GradScaler (this package's own Trainer), output gradients of the size a
    mean cross entropy over B*L rows produces must not flush to zero on their way down the released units.  The backward is
    linear in the incoming gradient: scaling it by 2^-24 (products of 6e-8 x O(1e-2) values: below fp16's smallest
    subnormal) has to scale every parameter gradient by exactly that factor, up to the arithmetic's relative error."""
    from feature_vs_text_compound_emotion_amd import synth
    from feature_vs_text_compound_emotion_amd.visual_backbone import VisualBackbone
    n, hw = 6, 40
    vsd = synth.make_state_dict(synth.visual_backbone_spec("", hw // 8), seed=31)
    gen = torch.Generator().manual_seed(32)
    frames = torch.randn(n, 3, hw, hw, generator=gen).cuda()
    G = torch.randn(n, 512, generator=gen).cuda()
    keep = ((torch.rand(n, hw // 8, hw // 8, 512, generator=gen) >= 0.4).float() / 0.6).cuda()

    def grads(scale):
        vb = VisualBackbone(use_pretrained=False, head_hw=hw // 8)
        vb.load_state_dict(vsd, strict=True)
        vb = vb.cuda().train()
        vb.backbone.precision = "fp16"
        params = list(vb.parameters())
        for p in params:
            p.requires_grad = False
        idx = list(range(4, 10)) + list(range(142, 187))      # head + stage 4 + the second half of stage 3
        for i in idx:
            params[i].requires_grad = True
        emb = vb(frames, keep)
        (emb * (G * scale)).sum().backward()
        return [params[i].grad.detach().double() / scale for i in idx]
    big, small = grads(1.0), grads(2.0 ** -24)
    worst = 0.0
    for a, b in zip(big, small):
        if a.norm().item() > 1e-6:
            worst = max(worst, ((a - b).norm() / a.norm()).item())
    print(f"\n[release fp16] gradients under a 2^-24 output-gradient scale: worst per-parameter relative difference {worst:.2e}")
    assert worst < 1e-3


def test_partial_unit_release_fails_loudly():
    g, vb, frames, G = _setup()
    for p in vb.parameters():
        p.requires_grad = False
    for p in vb.backbone.output_layer.parameters():
        p.requires_grad = True
    vb.backbone.body[23].res_layer[1].weight.requires_grad = True  # one conv of a unit only
    vb.train()
    with pytest.raises(NotImplementedError, match="whole units"):
        vb(frames.cuda())


@pytest.mark.parametrize("n,cin,cout,hw,k,stride", [(3, 64, 96, 9, 3, 1), (2, 128, 64, 10, 3, 2), (4, 64, 128, 9, 1, 2),
                                                    (3, 64, 64, 5, 3, 2), (2, 32, 64, 7, 3, 2), (2, 64, 32, 6, 1, 2)])
def test_conv2d_wgrad_and_dgrad_vs_autograd(n, cin, cout, hw, k, stride):
    import torch.nn.functional as F
    from feature_vs_text_compound_emotion_amd import ops
    from feature_vs_text_compound_emotion_amd.visual_backbone import _conv_dgrad
    gen = torch.Generator().manual_seed(n + cin)
    x = torch.randn(n, cin, hw, hw, generator=gen, requires_grad=True)
    w = (torch.randn(cout, cin, k, k, generator=gen) / (cin * k * k) ** 0.5).requires_grad_(True)
    y = F.conv2d(x, w, None, stride, k // 2)
    dy = torch.randn(y.shape, generator=gen)
    y.backward(dy)
    dz = dy.permute(0, 2, 3, 1).contiguous().cuda()
    dw = ops.conv2d_wgrad(dz, x.detach().permute(0, 2, 3, 1).contiguous().cuda(), k, k, stride=stride, pad=(k // 2, k // 2))
    assert (dw.cpu() - w.grad).abs().max().item() < 2e-4 * max(1.0, w.grad.abs().max().item())
    dx = _conv_dgrad(dz, w.detach().cuda(), stride, k // 2, (hw, hw))
    assert (dx.cpu().permute(0, 3, 1, 2) - x.grad).abs().max().item() < 2e-4


@pytest.mark.parametrize("precision,with_stem,n,hw,memory", [
    ("fp32", True, 8, 40, "raw"), ("bf16x3", True, 8, 40, "raw"), ("bf16x3", False, 8, 40, "raw"), ("bf16x3", True, 4, 64, "raw"),
    ("bf16x3", True, 8, 40, "recompute"), ("bf16x3", True, 4, 64, "recompute"),
    ("bf16x3", True, 8, 40, "recompute16"), ("bf16x3", True, 4, 64, "recompute16")])
def test_whole_encoder_backward_vs_float64_oracle(precision, with_stem, n, hw, memory):
    """BASELINE configs[1] ("IR-ResNet50 forward+backward"): every unit of the body (and the input layer) released -- an
    extension of the reference's schedule, which stops at half of stage 3 -- against torch autograd through the float64
    oracle (oracle/ir50.py, itself pinned to the reference's VisualBackbone) on the same frames, weights and head dropout
    mask.  Gradients are compared per parameter in relative L2 (an element-wise bound is ill posed: a PReLU pre-activation
    within rounding of zero flips its derivative, see test_released_stage4_matches_reference_step) and as one vector.
    64x64 frames (round 3): feature maps of 64 / 32 / 16 / 8 pixels -- multiples of 16, so the released units' forward and
    stride-1 data-gradient convs run on the patch / window kernels that carry the 224x224 geometry, and the stride-2 data
    gradients on the four-parity decomposition with even sizes (40x40 exercises the odd 5 -> 3 case)."""
    import oracle.ir50 as oracle_ir50
    from feature_vs_text_compound_emotion_amd import synth
    from feature_vs_text_compound_emotion_amd.visual_backbone import VisualBackbone
    vsd = synth.make_state_dict(synth.visual_backbone_spec("", hw // 8), seed=21)
    gen = torch.Generator().manual_seed(22)
    frames = torch.randn(n, 3, hw, hw, generator=gen)
    G = torch.randn(n, 512, generator=gen)
    keep = (torch.rand(n, 512, hw // 8, hw // 8, generator=gen) >= 0.4).double() / 0.6
    # float64 reference and the float32 yardstick (CPU autograd through the oracle: seconds each) are shared by the
    # parametrisations that differ only in the HIP side (precision, memory plan)
    key = (with_stem, n, hw)
    if key not in _ORACLE_CACHE:
        sd64 = {k[len("backbone."):]: v.double().clone() for k, v in vsd.items() if k.startswith("backbone.")}
        train_keys = [k for k in sd64 if not k.endswith(("running_mean", "running_var", "num_batches_tracked")) and
                      (with_stem or not k.startswith("input_layer."))]
        for k in train_keys:
            sd64[k].requires_grad_(True)
        emb64 = oracle_ir50.ir50_forward(frames.double(), sd64, train=True, head_dropout_mask=keep)
        (emb64 * G.double()).sum().backward()
        sd32 = {k: v.detach().float().clone() for k, v in sd64.items()}
        for k in train_keys:
            sd32[k].requires_grad_(True)
        (oracle_ir50.ir50_forward(frames, sd32, train=True, head_dropout_mask=keep.float()) * G).sum().backward()
        _ORACLE_CACHE[key] = (sd64, train_keys, emb64.detach(), sd32)
    sd64, train_keys, emb64, sd32 = _ORACLE_CACHE[key]
    # HIP
    vb = VisualBackbone(use_pretrained=False, head_hw=hw // 8)
    vb.load_state_dict(vsd, strict=True)
    vb = vb.cuda()
    vb.backbone.precision = precision
    # "recompute": units keep only their input and re-run their convs in the backward (the memory plan of bench.py --release 4
    # --hw 224 --batch 32): deterministic kernels -> gradients bit-identical to "raw" (checked below against a second model).
    # "recompute16": the input as one normalised fp16 plane -- 2^-11 against bf16x3's 2^-15 per operation, bar x4
    vb.backbone.activation_memory = memory
    for p_ in vb.parameters():
        p_.requires_grad = False
    named = dict(vb.backbone.named_parameters())
    for k in train_keys:
        named[k].requires_grad = True
    vb.train()
    emb = vb(frames.cuda(), keep.float().permute(0, 2, 3, 1).contiguous().cuda())
    assert (emb.detach().cpu().double() - emb64.detach()).abs().max().item() < (1e-5 if precision == "fp32" else 1e-4)
    (emb * G.cuda()).sum().backward()
    if not with_stem:
        assert all(named[k].grad is None for k in named if k.startswith("input_layer."))
    # yardstick (sd32, cached above): torch's own float32 autograd through the same oracle -- how far plain fp32 arithmetic
    # lands from float64 on this 24-unit, batch-statistics, 8-frame problem

    def errors(grad_of):
        num = den = 0.0
        worst = (0.0, "")
        for k in train_keys:
            ref, got = sd64[k].grad, grad_of(k)
            assert got.shape == ref.shape, k
            d, r = (got - ref).norm().item(), ref.norm().item()
            num, den = num + d * d, den + r * r
            if r > 1e-6:
                worst = max(worst, (d / r, k))
        return worst, (num / den) ** 0.5

    worst32, all32 = errors(lambda k: sd32[k].grad.double())
    worst, allv = errors(lambda k: named[k].grad.detach().cpu().double())
    print(f"whole-encoder backward [{precision}, {n} x {hw}x{hw}, {memory}]: worst per-parameter rel L2 {worst[0]:.2e} ({worst[1]}), all gradients {allv:.2e}; "
          f"torch fp32: {worst32[0]:.2e} ({worst32[1]}), {all32:.2e}")
    factor = 4.0 if precision == "fp32" else 16.0   # bf16x3: 2^-15 per product against fp32's 2^-24 accumulation noise
    if memory == "recompute16":
        factor *= 4.0
    assert worst[0] < factor * worst32[0] + 1e-4, (worst, worst32)
    assert allv < factor * all32 + 1e-5, (allv, all32)
    if memory == "recompute":
        got = {k: named[k].grad.detach().clone() for k in train_keys}
        vb2 = VisualBackbone(use_pretrained=False, head_hw=hw // 8)
        vb2.load_state_dict(vsd, strict=True)
        vb2 = vb2.cuda()
        vb2.backbone.precision = precision
        for p_ in vb2.parameters():
            p_.requires_grad = False
        named2 = dict(vb2.backbone.named_parameters())
        for k in train_keys:
            named2[k].requires_grad = True
        vb2.train()
        (vb2(frames.cuda(), keep.float().permute(0, 2, 3, 1).contiguous().cuda()) * G.cuda()).sum().backward()
        assert all(torch.equal(got[k], named2[k].grad) for k in train_keys), "recompute != raw"


@pytest.mark.parametrize("n,cin,cout,hw,k,stride,prec", [
    (40, 128, 128, 10, 3, 1, "bf16x3"), (300, 256, 128, 5, 3, 1, "bf16x3"), (37, 128, 256, 10, 3, 2, "bf16x3"),
    (50, 128, 256, 9, 1, 2, "bf16x3"), (3, 128, 128, 7, 3, 1, "bf16x3"), (40, 128, 128, 10, 3, 1, "fp16"),
    # channel counts below the 128 x 128 output tile (stage 1 / 2 layers, the 3 -> 4 channel stem)
    (9, 64, 64, 12, 3, 1, "bf16x3"), (9, 64, 128, 12, 3, 2, "bf16x3"), (5, 4, 64, 20, 3, 1, "bf16x3"),
    # stride-2 data gradient as four parity convs: odd sizes (the 5 -> 3 layer of the 40x40 crop), narrow storage
    (6, 256, 512, 5, 3, 2, "bf16x3"), (4, 64, 64, 9, 3, 2, "bf16x3"), (4, 128, 128, 10, 3, 2, "fp16"), (4, 64, 128, 7, 3, 2, "bf16"),
    (5, 128, 256, 7, 1, 2, "bf16x3"),
    # the 256 x 256-tile weight-gradient kernel (8 waves, both channel counts multiples of 256): one / several tiles per tap,
    # a row count that is not a multiple of the 32-pixel step, stride 2, 1x1
    (40, 256, 256, 10, 3, 1, "bf16x3"), (21, 512, 512, 5, 3, 1, "bf16x3"), (33, 256, 512, 8, 1, 2, "bf16x3"),
    # the 224x224 geometry bench.py --release 4 --hw 224 times (round-2 verdict: parity-tested at 40x40 only): 6.4 M / 1.6 M pixel
    # rows per weight-gradient reduction, patch-kernel data gradients (H, W % 16 == 0), the stride-2 layers of units 1 and 4
    (2, 64, 64, 224, 3, 1, "bf16x3"), (2, 64, 64, 224, 3, 2, "bf16x3"), (2, 128, 128, 112, 3, 2, "bf16x3"),
    (2, 64, 128, 112, 1, 2, "bf16x3")])
def test_conv2d_wgrad_b3_and_dgrad_in_the_encoder_precision(n, cin, cout, hw, k, stride, prec):
    """The matrix-core weight gradient (transposed LDS reads, split operands, pixel range split over blocks and reduced in a
    fixed order) and the data gradient on the bf16x3 / narrow conv kernels, against float64 autograd.  Error model: every
    product drops lo*lo and rounds both lo parts: <= 2^-15 sum |dz| |x| (bf16x3); the narrow data gradient rounds both
    operands once (2^-11 fp16)."""
    import torch.nn.functional as F
    from feature_vs_text_compound_emotion_amd import ops
    from feature_vs_text_compound_emotion_amd.visual_backbone import _conv_dgrad
    gen = torch.Generator().manual_seed(n + cin + cout)
    x = torch.randn(n, cin, hw, hw, generator=gen, dtype=torch.float64, requires_grad=True)
    w = (torch.randn(cout, cin, k, k, generator=gen, dtype=torch.float64) / (cin * k * k) ** 0.5).requires_grad_(True)
    y = F.conv2d(x, w, None, stride, k // 2)
    dy = torch.randn(y.shape, generator=gen, dtype=torch.float64)
    y.backward(dy)
    # magnitude sums for the bounds: the same contractions on absolute values
    xa, wa = x.detach().abs().requires_grad_(True), w.detach().abs().requires_grad_(True)
    F.conv2d(xa, wa, None, stride, k // 2).backward(dy.abs())
    dz = dy.float().permute(0, 2, 3, 1).contiguous().cuda()
    xd = x.detach().float().permute(0, 2, 3, 1).contiguous().cuda()
    dw = ops.conv2d_wgrad(dz, xd, k, k, stride=stride, pad=(k // 2, k // 2), b3=True)
    npix = n * y.shape[2] * y.shape[3]
    bound = wa.grad * (2.0 ** -15 + npix * 2.0 ** -24) + 1e-6
    err = (dw.cpu().double() - w.grad).abs()
    assert (err <= bound).all(), (err / bound).max().item()
    # operands that are split tensors already (what the released units keep): the same arithmetic, no conversion in the loader
    dw_s = ops.conv2d_wgrad(ops.split_bf16(dz), ops.split_bf16(xd), k, k, stride=stride, pad=(k // 2, k // 2), b3=True)
    if cin % 256 == 0 and cout % 256 == 0:   # split operands of this width take the 256 x 256 tile: another summation order
        err = (dw_s.cpu().double() - w.grad).abs()
        assert (err <= bound).all(), (err / bound).max().item()
    else:
        assert torch.equal(dw_s, dw)
    ref_fp32 = ops.conv2d_wgrad(dz, xd, k, k, stride=stride, pad=(k // 2, k // 2))          # the fp32-MFMA kernel agrees
    assert (ref_fp32.cpu().double() - w.grad).abs().max().item() < 1e-3 * max(1.0, w.grad.abs().max().item())
    dx = _conv_dgrad(dz, w.detach().float().cuda(), stride, k // 2, (hw, hw), prec)
    eps = {"bf16x3": 2.0 ** -15, "fp16": 2.0 ** -10, "bf16": 2.0 ** -7}[prec]
    bound = xa.grad * (eps + cout * k * k * 2.0 ** -24) + 1e-6
    err = (dx.cpu().double().permute(0, 3, 1, 2) - x.grad).abs()
    assert (err <= bound).all(), (err / bound).max().item()


def test_prelu_fwd_bwd_vs_autograd():
    import torch.nn.functional as F
    from feature_vs_text_compound_emotion_amd import ops
    gen = torch.Generator().manual_seed(8)
    x = torch.randn(50, 64, generator=gen, requires_grad=True)
    a = (torch.rand(64, generator=gen) * 0.4 - 0.1).requires_grad_(True)  # also negative slopes
    dy = torch.randn(50, 64, generator=gen)
    y = F.prelu(x.t().reshape(1, 64, 50), a)  # channel dim 1
    y.backward(dy.t().reshape(1, 64, 50))
    assert torch.equal(ops.prelu_fwd(x.detach().cuda(), a.detach().cuda()).cpu(), y.detach().reshape(64, 50).t())
    dx, da = ops.prelu_bwd(dy.cuda(), x.detach().cuda(), a.detach().cuda())
    assert (dx.cpu() - x.grad).abs().max().item() < 1e-6 and (da.cpu() - a.grad).abs().max().item() < 1e-4


def test_fused_split_passes_equal_the_two_pass_forms():
    """Round 3: PReLU forward / backward and the BatchNorm backward's apply pass write the Split tensor the matrix-core kernels
    read in the same pass (cer_prelu_split, cer_prelu_bwd_split, cer_bn_rows_bwd_split).  Same arithmetic as "fp32 result, then
    cer_split_bf16": the PReLU forms bit for bit, the BatchNorm form to fp32 rounding (the two kernels contract a*b+c differently)."""
    from feature_vs_text_compound_emotion_amd import ops
    gen = torch.Generator().manual_seed(12)
    x = torch.randn(5, 7, 9, 64, generator=gen).cuda()
    dy = torch.randn(5, 7, 9, 64, generator=gen).cuda()
    a = (torch.rand(64, generator=gen) * 0.4 - 0.1).cuda()
    ref = ops.split_bf16(ops.prelu_fwd(x, a))
    got = ops.prelu_split(x, a)
    assert torch.equal(got.hi, ref.hi) and torch.equal(got.lo, ref.lo)
    dx, da = ops.prelu_bwd(dy, x, a)
    dxs, das = ops.prelu_bwd(dy, x, a, split_out=True)
    r = ops.split_bf16(dx)
    assert torch.equal(dxs.hi, r.hi) and torch.equal(dxs.lo, r.lo) and torch.equal(da, das)
    big = torch.empty(10, 7, 9, 64, device="cuda")     # chunked calls write into slices
    ops.prelu_bwd(dy, x, a, out=big[5:])
    assert torch.equal(big[5:], dx)
    rows, c = 5 * 7 * 9, 64
    w = (torch.rand(c, generator=gen) + 0.5).cuda()
    mean, invstd = x.view(rows, c).mean(0), torch.rsqrt(x.view(rows, c).var(0, unbiased=False) + 1e-5)
    d32, dw, db = ops.bn_rows_bwd(dy.view(rows, c), x.view(rows, c), mean, invstd, w)
    ds, dws, dbs = ops.bn_rows_bwd(dy.view(rows, c), x.view(rows, c), mean, invstd, w, split_out=True)
    assert torch.equal(dw, dws) and torch.equal(db, dbs)
    # a Split carries 16 mantissa bits: compare with the split of the fp32 result (the two differ by fp32 rounding of the inputs)
    assert (ds.float() - ops.split_bf16(d32).float()).abs().max().item() < 2.0 ** -15 * max(1.0, d32.abs().max().item())
    # against torch autograd
    xt = x.view(rows, c).cpu().double().requires_grad_(True)
    wt = w.cpu().double().requires_grad_(True)
    y = torch.nn.functional.batch_norm(xt, None, None, wt, torch.zeros(c, dtype=torch.float64), True, 0.0, 1e-5)
    y.backward(dy.view(rows, c).cpu().double())
    assert (ds.float().cpu().double() - xt.grad).abs().max().item() < 2.0 ** -14 * max(1.0, xt.grad.abs().max().item())
