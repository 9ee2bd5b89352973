"""Trainable-tail kernels (forward and hand-written backward) vs torch-CPU autograd on the oracle."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from helpers import MODS, golden, masks_from_golden  # noqa: E402

TOL = 2e-5


def _ops():
    from feature_vs_text_compound_emotion_amd import ops
    return ops


def _close(a, b, tol=TOL):
    a = a.detach().cpu() if torch.is_tensor(a) else torch.as_tensor(a)
    b = b.detach().cpu() if torch.is_tensor(b) else torch.as_tensor(b)
    assert a.shape == b.shape, (a.shape, b.shape)
    err = (a - b).abs().max().item()
    assert err < tol, err


def test_weight_norm_fwd_bwd():
    from oracle.tcn import weight_norm_weight
    ops = _ops()
    g_ = torch.Generator().manual_seed(1)
    v = torch.randn(48, 64, 5, generator=g_, requires_grad=True)
    g = (torch.rand(48, 1, 1, generator=g_) + 0.5).requires_grad_(True)
    w = weight_norm_weight(g, v)
    dw = torch.randn(48, 64, 5, generator=g_)
    w.backward(dw)
    wd, norm = ops.weight_norm_fwd(v.detach().cuda(), g.detach().cuda())
    _close(wd, w)
    dv, dg = ops.weight_norm_bwd(dw.cuda(), v.detach().cuda(), g.detach().cuda(), norm)
    _close(dv, v.grad)
    _close(dg, g.grad)


@pytest.mark.parametrize("cout,cin,k", [(48, 64, 5), (32, 40, 5), (512, 768, 5), (5, 7, 3)])
def test_weight_norm_fwd_packed_equals_weight_norm_then_two_packs(cout, cin, k):
    """One launch = weight-norm + the packed forward filter + the flipped / transposed data-gradient filter, bit for bit what the
    three separate launches produce (zero padding included)."""
    ops = _ops()
    g_ = torch.Generator().manual_seed(cout + cin)
    v, g = torch.randn(cout, cin, k, generator=g_).cuda(), (torch.rand(cout, 1, 1, generator=g_) + 0.5).cuda()
    w0, n0 = ops.weight_norm_fwd(v, g)
    w, n, wp, wt = ops.weight_norm_fwd_packed(v, g)
    assert torch.equal(w, w0) and torch.equal(n, n0)
    assert torch.equal(wp, ops.pack_conv_weight(w0.view(cout, cin, k, 1)))
    assert torch.equal(wt, ops.pack_conv_weight(w0.view(cout, cin, k, 1), flip=True, transpose=True))


@pytest.mark.parametrize("cout,cin,k,dil,bsz,length", [(64, 128, 5, 4, 3, 16), (32, 32, 5, 2, 32, 32), (512, 768, 5, 1, 32, 32)])
def test_wgrad_into_weight_norm_backward_equals_the_three_launches(cout, cin, k, dil, bsz, length):
    """conv1d_wgrad_weight_norm_bwd: the weight-norm backward sums the weight-gradient kernel's split partial slabs itself, in
    the order the fold kernel used -- (dv, dg) bit-identical to conv1d_wgrad + weight_norm_bwd."""
    ops = _ops()
    g_ = torch.Generator().manual_seed(cout + cin + k)
    dz, x = torch.randn(bsz * length, cout, generator=g_).cuda(), torch.randn(bsz * length, cin, generator=g_).cuda()
    v, g = torch.randn(cout, cin, k, generator=g_).cuda(), (torch.rand(cout, 1, 1, generator=g_) + 0.5).cuda()
    _, norm = ops.weight_norm_fwd(v, g)
    dv0, dg0 = ops.weight_norm_bwd(ops.conv1d_wgrad(dz, x, length, k, dil), v, g, norm)
    dv, dg = ops.conv1d_wgrad_weight_norm_bwd(dz, x, length, k, dil, v, g, norm)
    assert torch.equal(dv, dv0) and torch.equal(dg, dg0)


@pytest.mark.parametrize("cout,cin,k,dil,bsz,length", [(64, 128, 5, 4, 3, 16), (7, 224, 1, 1, 2, 8),
                                                        (96, 32, 1, 1, 2, 8), (40, 72, 5, 1, 2, 8)])
def test_conv1d_wgrad(cout, cin, k, dil, bsz, length):
    ops = _ops()
    g = torch.Generator().manual_seed(cout + cin)
    x = torch.randn(bsz, cin, length, generator=g)
    w = torch.randn(cout, cin, k, generator=g, requires_grad=True)
    dz = torch.randn(bsz, cout, length, generator=g)
    F.conv1d(F.pad(x, ((k - 1) * dil, 0)), w, None, dilation=dil).backward(dz)
    xr = x.transpose(1, 2).reshape(bsz * length, cin).contiguous().cuda()
    dzr = dz.transpose(1, 2).reshape(bsz * length, cout).contiguous().cuda()
    _close(ops.conv1d_wgrad(dzr, xr, length, k, dil), w.grad, 1e-4)


def test_col_sum_large():
    ops = _ops()
    a = torch.randn(1500, 70, generator=torch.Generator().manual_seed(2))
    _close(ops.col_sum(a.cuda()), a.sum(0), 1e-3)


def test_bn_rows_fwd_bwd_train_and_eval():
    ops = _ops()
    g = torch.Generator().manual_seed(3)
    r, c = 96, 40
    x = torch.randn(r, c, generator=g, requires_grad=True)
    w = (torch.rand(c, generator=g) + 0.5).requires_grad_(True)
    b = torch.randn(c, generator=g).requires_grad_(True)
    rm, rv = torch.randn(c, generator=g) * 0.1, torch.rand(c, generator=g) + 0.5
    dy = torch.randn(r, c, generator=g)
    for train in (True, False):
        rm_c, rv_c = rm.clone(), rv.clone()
        y = F.batch_norm(x, rm_c, rv_c, w, b, train, 0.1, 1e-5)
        for t in (x, w, b):
            t.grad = None
        y.backward(dy)
        rm_d, rv_d = rm.clone().cuda(), rv.clone().cuda()
        yd, sm, si = ops.bn_rows_fwd(x.detach().cuda(), w.detach().cuda(), b.detach().cuda(), rm_d, rv_d, train)
        _close(yd, y)
        _close(rm_d, rm_c)
        _close(rv_d, rv_c)
        if not train:
            sm, si = rm_d, torch.rsqrt(rv_d + 1e-5)
        dx, dw, db = ops.bn_rows_bwd(dy.cuda(), x.detach().cuda(), sm, si, w.detach().cuda(), train)
        _close(dx, x.grad)
        _close(dw, w.grad, 1e-4)
        _close(db, b.grad, 1e-4)


def test_bn_rows_fwd_large_r_statistics_are_column_sums():
    """R > 2048 (the released encoder units: 25600 .. 102400 rows): mean / variance through two deterministic column sums
    instead of the block-per-32-channels walk; same results as torch on data with a large mean (the sum x (x - mean) form
    does not cancel)."""
    ops = _ops()
    g = torch.Generator().manual_seed(13)
    r, c = 30011, 96
    x = torch.randn(r, c, generator=g, dtype=torch.float64) * (torch.rand(c, generator=g, dtype=torch.float64) + 0.2) + \
        torch.randn(c, generator=g, dtype=torch.float64) * 8
    w, b = torch.rand(c, generator=g) + 0.5, torch.randn(c, generator=g)
    rm, rv = torch.randn(c, generator=g) * 0.1, torch.rand(c, generator=g) + 0.5
    rm64, rv64 = rm.double().clone(), rv.double().clone()
    y = F.batch_norm(x, rm64, rv64, w.double(), b.double(), True, 0.1, 1e-5)
    rm_d, rv_d = rm.clone().cuda(), rv.clone().cuda()
    yd, sm, si = ops.bn_rows_fwd(x.float().cuda(), w.cuda(), b.cuda(), rm_d, rv_d, True)
    assert (yd.cpu().double() - y).abs().max().item() < 2e-4          # fp32 storage of x with |mean| ~ 8: 1e-6 * 8 / std
    assert (sm.cpu().double() - x.mean(0)).abs().max().item() < 2e-5
    assert ((si.cpu().double() - 1 / torch.sqrt(x.var(0, unbiased=False) + 1e-5)).abs() * x.std(0)).max().item() < 2e-4
    assert (rm_d.cpu().double() - rm64).abs().max().item() < 1e-5
    assert ((rv_d.cpu().double() - rv64).abs() / rv64).max().item() < 1e-4


def test_cross_entropy_fwd_bwd():
    from feature_vs_text_compound_emotion_amd.lfan import cross_entropy_loss
    g = torch.Generator().manual_seed(4)
    logits = torch.randn(3, 11, 7, generator=g, requires_grad=True)
    labels = torch.randint(0, 7, (3, 11, 1), generator=g).float()
    ref = F.cross_entropy(logits.reshape(-1, 7), labels.reshape(-1).long())
    ref.backward()
    ld = logits.detach().cuda().requires_grad_(True)
    loss = cross_entropy_loss(ld, labels.cuda())
    loss.backward()
    _close(loss, ref, 1e-6)
    _close(ld.grad, logits.grad, 1e-7)


def test_cross_entropy_label_semantics_match_torch():
    """nn.CrossEntropyLoss: ignore_index = -100 rows are skipped (mean over the others, zero gradient); any other label
    outside [0, C) raises.  The kernel never dereferences such a row; the error surfaces through the deferred counter."""
    from feature_vs_text_compound_emotion_amd import ops
    from feature_vs_text_compound_emotion_amd.lfan import cross_entropy_loss
    g = torch.Generator().manual_seed(41)
    logits = torch.randn(40, 7, generator=g, requires_grad=True)
    labels = torch.randint(0, 7, (40,), generator=g)
    labels[[3, 17, 39]] = -100
    ref = F.cross_entropy(logits, labels)
    ref.backward()
    ld = logits.detach().cuda().requires_grad_(True)
    loss = cross_entropy_loss(ld, labels.cuda())       # long labels, as the reference's trainer passes them
    loss.backward()
    _close(loss, ref, 1e-6)
    _close(ld.grad, logits.grad, 1e-7)
    ops.flush_label_check()
    for bad_value in (7.0, -1.0, float("nan")):   # 7 = the 'Other' class with its column dropped, -1 = ABAW invalid frame
        bad = labels.float().clone()
        bad[5] = bad_value
        with pytest.raises(IndexError, match="out of bounds"):
            out = cross_entropy_loss(ld.detach(), bad.cuda())
            assert torch.isnan(out).item()            # the step fails visibly even before the counter arrives
            ops.flush_label_check()
    ops.flush_label_check()                           # nothing pending: later calls are clean
    _close(cross_entropy_loss(ld.detach(), labels.cuda()), ref, 1e-6)


def test_dropout_mask_statistics_and_determinism():
    ops = _ops()
    m1 = ops.dropout_mask((1 << 16,), 0.1, 7, 0, "cuda")
    m2 = ops.dropout_mask((1 << 16,), 0.1, 7, 0, "cuda")
    m3 = ops.dropout_mask((1 << 16,), 0.1, 8, 0, "cuda")
    assert torch.equal(m1, m2) and not torch.equal(m1, m3)
    keep = (m1 != 0).float().mean().item()
    assert abs(keep - 0.9) < 0.01
    assert torch.allclose(m1[m1 != 0], torch.tensor(1 / 0.9, device="cuda"))


def _tcn_pair(cin, channels, k, seed):
    from feature_vs_text_compound_emotion_amd import synth
    from feature_vs_text_compound_emotion_amd.temporal_convnet import TemporalConvNet
    spec, alias = synth.tcn_spec("", cin, channels, k)
    sd = synth.make_state_dict(spec, alias, seed)
    net = TemporalConvNet(cin, channels, kernel_size=k, dropout=0.1)
    assert set(net.state_dict()) == set(sd)
    net.load_state_dict(sd, strict=True)
    return net.cuda(), sd, alias


@pytest.mark.parametrize("train", [False, True])
def test_tcn_forward_backward_vs_oracle(train):
    from feature_vs_text_compound_emotion_amd import synth
    from oracle.tcn import tcn_forward
    cin, channels, k, bsz, length = 128, [64, 64, 32, 32], 5, 3, 16
    net, sd, alias = _tcn_pair(cin, channels, k, 21)
    g = torch.Generator().manual_seed(22)
    x = torch.randn(bsz, cin, length, generator=g, requires_grad=True)
    masks = None
    if train:
        masks = [(synth.dropout_mask((bsz, c, length), 0.1, g), synth.dropout_mask((bsz, c, length), 0.1, g))
                 for c in channels]
    names = [n for n in sd if n not in alias]
    params = {n: sd[n].clone().requires_grad_(True) for n in names}
    full = dict(params)
    for a, s in alias.items():
        full[a] = full[s]
    y = tcn_forward(x, full, "", masks)
    dy = torch.randn(y.shape, generator=g)
    y.backward(dy)
    net.train(train)
    xr = x.detach().transpose(1, 2).reshape(bsz * length, cin).contiguous().cuda().requires_grad_(True)
    dmasks = None
    if train:
        dmasks = [tuple(m.transpose(1, 2).reshape(bsz * length, -1).contiguous().cuda() for m in pair) for pair in masks]
    yd = net.forward_rows(xr, bsz, length, masks=dmasks)
    _close(yd.view(bsz, length, -1).transpose(1, 2), y)
    yd.backward(dy.transpose(1, 2).reshape(bsz * length, -1).contiguous().cuda())
    _close(xr.grad.view(bsz, length, cin).transpose(1, 2), x.grad)
    got = dict(net.named_parameters())
    assert set(got) == set(names)
    for n in names:
        _close(got[n].grad, params[n].grad, 5e-5)


def test_tcn_reference_layout_entry_point():
    from oracle.tcn import tcn_forward
    net, sd, _ = _tcn_pair(32, [32, 16], 3, 5)
    x = torch.randn(2, 32, 10, generator=torch.Generator().manual_seed(6))
    net.eval()
    _close(net(x.cuda()), tcn_forward(x, sd, ""))


def _build_lfan(mods, sd, length, n_cls=7, head_hw=5):
    from feature_vs_text_compound_emotion_amd import synth
    from feature_vs_text_compound_emotion_amd.lfan import LFAN
    m = LFAN(backbone_settings={}, output_dim=n_cls, task="CLASSIFICATION", modality=mods, example_length=length,
             kernel_size=5, tcn_channel=synth.TCN_CHANNELS, modal_dim=32, num_heads=2, root_dir="", device="cuda",
             head_hw=head_hw)
    m.init(load_backbone=False)
    assert set(m.state_dict()) == set(sd), set(m.state_dict()) ^ set(sd)
    m.load_state_dict(sd, strict=True)
    return m.cuda()


def _golden_setup(g, mods=MODS):
    from feature_vs_text_compound_emotion_amd import synth
    b, l, hw, ncls, wseed, dseed = [int(v) for v in g["meta"]]
    sd = synth.lfan_state_dict(mods, n_cls=ncls, head_hw=hw // 8, seed=wseed)
    x, labels = synth.make_clip_batch(mods, b, l, hw=hw, seed=dseed)
    return sd, x, labels, (b, l, hw)


def test_lfan_eval_logits_match_reference_fixture():
    g = golden("lfan_trimodal_eval.npz")
    sd, x, _, (b, l, hw) = _golden_setup(g)
    model = _build_lfan(MODS, sd, l).eval()
    with torch.no_grad():
        logits = model({k: v.cuda() for k, v in x.items()})
    assert np.abs(logits.cpu().numpy() - g["logits"]).max() < 1e-4  # fp32, different summation order


def test_lfan_modality_subsets_and_order():
    from feature_vs_text_compound_emotion_amd import synth
    g = golden("lfan_modal_subsets_eval.npz")
    b, l, hw, ncls, wseed, dseed = [int(v) for v in g["meta"]]
    for mods in (["video"], ["video", "vggish"], ["vggish", "video"], ["bert", "vggish"]):
        sd = synth.lfan_state_dict(mods, n_cls=ncls, head_hw=hw // 8, seed=wseed)
        x, _ = synth.make_clip_batch(mods, b, l, hw=hw, seed=dseed)
        model = _build_lfan(mods, sd, l).eval()
        with torch.no_grad():
            logits = model({k: v.cuda() for k, v in x.items()})
        assert np.abs(logits.cpu().numpy() - g["logits_" + "_".join(mods)]).max() < 1e-4


def test_lfan_logmel_modality_key_order_and_dict_write_back():
    """Boundary details of LFAN.forward (models/model.py:487-526) against a fixture from the reference
    (tools/gen_golden_logmel.py): the 'logmel' modality runs VGGish inside forward; the caller's dict may come in any key
    order (the fusion walks the model's modality list); afterwards it holds the per-modality features [B, L, C_m]; a key the
    model was not built for is a KeyError."""
    from feature_vs_text_compound_emotion_amd import synth
    g = golden("lfan_logmel.npz")
    b, l, ncls, wseed, dseed = [int(v) for v in g["meta"]]
    mods = ["logmel", "vggish"]
    spec, alias = synth.lfan_spec(mods, n_cls=ncls)
    sd = synth.make_state_dict(spec, alias, seed=wseed)
    x, _ = synth.make_clip_batch(mods, b, l, seed=dseed)
    model = _build_lfan(mods, sd, l, n_cls=ncls).eval()
    with torch.no_grad():
        caller = {k: v.cuda() for k, v in x.items()}
        logits = model(caller)
        swapped = model({k: x[k].cuda() for k in reversed(mods)})
    assert np.abs(logits.cpu().numpy() - g["logits"]).max() < 1e-4
    assert torch.equal(logits, swapped)
    for m in mods:      # what the reference leaves in the dict it was handed
        assert tuple(caller[m].shape) == g["left_" + m].shape
        assert np.abs(caller[m].cpu().numpy() - g["left_" + m]).max() < 1e-4 * max(1.0, np.abs(g["left_" + m]).max()), m
    with pytest.raises(KeyError):
        model({"logmel": x["logmel"].cuda(), "vggish": x["vggish"].cuda(), "bert": torch.zeros(b, 1, l, 768).cuda()})
    with pytest.raises(KeyError):
        model({"logmel": x["logmel"].cuda()})


@pytest.mark.parametrize("tag", ["refmode", "evalbackbone"])
def test_lfan_two_training_steps_match_reference_fixture(tag):
    """trainer.py:365-391 on the HIP path: zero_grad, forward, CE, backward, Nesterov SGD (lr 1e-3).
    Fixtures recorded from the reference with dropout off: "refmode" = model.train() exactly as the
    reference runs it (frozen encoder BatchNorms in batch-statistics mode); "evalbackbone" = encoder
    kept in eval mode (the HIP path's ``bn_mode = "frozen"``)."""
    from feature_vs_text_compound_emotion_amd import synth
    from feature_vs_text_compound_emotion_amd.lfan import cross_entropy_loss
    g = golden(f"lfan_trimodal_train_steps_{tag}.npz")
    sd, _, _, (b, l, hw) = _golden_setup(g)
    model = _build_lfan(MODS, sd, l).train()
    model.spatial["visual"].backbone.bn_mode = "reference" if tag == "refmode" else "frozen"
    for mod in model.modules():
        if isinstance(mod, torch.nn.Dropout):
            mod.p = 0.0
    for net in model.temporal.values():
        net.dropout = 0.0
    names = [n for n, p in model.named_parameters() if p.requires_grad]
    assert names == list(g["names"])
    params = [p for _, p in model.named_parameters() if p.requires_grad]
    opt = torch.optim.SGD(params=params, momentum=0.9, dampening=0.0, weight_decay=1e-4, nesterov=True)
    for step in range(2):
        xs, ls = synth.make_clip_batch(MODS, b, l, hw=hw, seed=int(g["meta"][5]) + step)
        opt.zero_grad(set_to_none=True)
        out = model({k: v.cuda() for k, v in xs.items()})
        loss = cross_entropy_loss(out, ls.cuda())
        loss.backward()
        assert abs(loss.item() - float(g[f"loss{step}"])) < 1e-4
        assert np.abs(out.detach().cpu().numpy() - g[f"logits{step}"]).max() < 2e-4
        gn = np.array([p.grad.norm().item() for p in params])
        assert np.abs(gn - g[f"gradnorm{step}"]).max() < 2e-4 * max(1.0, np.abs(g[f"gradnorm{step}"]).max())
        named = dict(zip(names, params))
        for key in g.files:
            if key.startswith(f"grad{step}:"):
                assert np.abs(named[key.split(":", 1)[1]].grad.cpu().numpy() - g[key]).max() < 5e-5, key
        opt.step()
    named = dict(model.named_parameters())
    for key in g.files:
        if key.startswith("param2:"):
            assert np.abs(named[key.split(":", 1)[1]].detach().cpu().numpy() - g[key]).max() < 1e-6, key
    assert np.abs(model.bn["video"].running_mean.cpu().numpy() - g["bn_video_running_mean2"]).max() < 1e-5


def test_lfan_train_forward_with_reference_dropout_masks_matches_fixture():
    """Full model.train() forward of the reference with its own dropout masks (captured by hooks when
    the fixture was recorded): dropout placement in the encoder head, the TCNs and the fusion, and
    batch-statistics BatchNorm everywhere."""
    g = golden("lfan_trimodal_train_fwd.npz")
    sd, x, _, (b, l, hw) = _golden_setup(g)
    masks = masks_from_golden(g)
    model = _build_lfan(MODS, sd, l).train()
    model.test_masks = {
        "head": masks["head"].permute(0, 2, 3, 1).contiguous().cuda(),
        "tcn": {m: [tuple(t.transpose(1, 2).reshape(b * l, -1).contiguous().cuda() for t in pair) for pair in v]
                for m, v in masks["tcn"].items()},
        "fusion": masks["fusion"].cuda()}
    with torch.no_grad():
        out = model({k: v.cuda() for k, v in x.items()})
    assert np.abs(out.cpu().numpy() - g["logits"]).max() < 2e-4
    sd_after = model.state_dict()
    assert np.abs(sd_after["bn.video.running_mean"].cpu().numpy() - g["bn_video_running_mean"]).max() < 1e-5
    assert np.abs(sd_after["bn.video.running_var"].cpu().numpy() - g["bn_video_running_var"]).max() < 1e-5
    assert np.abs(sd_after["spatial.visual.backbone.input_layer.1.running_mean"].cpu().numpy()
                  - g["stem_running_mean"]).max() < 1e-5


def test_trainer_windowed_inference_on_hip_model_vs_oracle():
    """A 21-frame video through a window-8 / hop-5 LFAN: slide, forward on the GPU, stitch, average."""
    from feature_vs_text_compound_emotion_amd import synth
    from feature_vs_text_compound_emotion_amd.trainer import Trainer, windowing
    from oracle.lfan import lfan_forward
    mods = ["vggish", "bert"]
    sd = synth.lfan_state_dict(mods, n_cls=7, seed=9)
    model = _build_lfan(mods, sd, 8).eval()
    n = 21
    g = torch.Generator().manual_seed(77)
    X = {"vggish": torch.randn(1, 1, n, 128, generator=g), "bert": torch.randn(1, 1, n, 768, generator=g),
         "EXPR_continuous_label": torch.full((1, n, 1), 3.0)}
    tr = Trainer(model, device="cuda", window_length=8, hop_length=5, number_classes=7)
    perf, per_video = tr.inference([(X, ["clip0"], [n], [np.arange(n)])], keep_logits=True)
    acc = np.zeros((n, 7))
    cnt = np.zeros(n)
    for wd in windowing(np.arange(n), 8, 5):
        with torch.no_grad():
            o = lfan_forward({m: X[m][:, :, wd] for m in mods}, sd, mods)
        acc[wd] += o[0].numpy()
        cnt[wd] += 1
    assert np.abs(per_video["clip0"]["logits"] - acc / cnt[:, None]).max() < 1e-4
    assert per_video["clip0"]["labels"].tolist() == [3] * n


def test_lfan_reference_window_length_300_vs_oracle():
    """The reference's default training window (300 frames, base/dataset.py windows 300/200): TCN receptive
    field 121 frames, causal padding and dilation 8 inside a long sequence; feature modalities only."""
    from feature_vs_text_compound_emotion_amd import synth
    from feature_vs_text_compound_emotion_amd.lfan import cross_entropy_loss
    from oracle.lfan import cross_entropy_mean, lfan_forward
    mods = ["bert", "vggish"]
    sd = synth.lfan_state_dict(mods, n_cls=8, seed=17)
    x, labels = synth.make_clip_batch(mods, 2, 300, seed=18, n_cls=8)
    w = sd["regressor.weight"].clone().requires_grad_(True)
    osd = dict(sd)
    osd["regressor.weight"] = w
    ref = lfan_forward(x, osd, mods, train=True)
    rloss = cross_entropy_mean(ref, labels)
    rloss.backward()
    model = _build_lfan(mods, sd, 300, n_cls=8).train()
    for net in model.temporal.values():
        net.dropout = 0.0
    model.fusion.layers.dropout.p = 0.0
    out = model({k: v.cuda() for k, v in x.items()})
    loss = cross_entropy_loss(out, labels.cuda())
    loss.backward()
    _close(out, ref, 2e-4)
    assert abs(loss.item() - rloss.item()) < 1e-4
    _close(model.regressor.weight.grad, w.grad, 1e-4)
