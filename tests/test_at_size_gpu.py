"""Correctness AT THE SIZE bench.py times: N = 1024 (and 2048) frames of 224x224 -- 51.4 M pixels per layer, 6.6 GB per bf16
plane at Cout = 64 (past 2^31 / 2^32 bytes), where the conv kernels rely on a 64-bit per-tile base + 32-bit relative offsets
and on 0x80000000 as the out-of-range marker, and where the pickers choose the large-M tiles (bf16x3: 128x128 and 256x64;
narrow: the patch kernels, 256x256, 256x128) that no small test reaches.

Frames are independent in eval(), so the embeddings of four probe frames of the big batch are compared with the oracle run
on those four frames alone; a single 64 -> 64 layer at N = 1024 is checked on sampled output pixels of the LAST frames
(highest addresses); and the JMT / MT fusion runs at cfg3's 1024 tokens (32 clips x 32 frames) against the oracle.
"""
import pytest
import torch

pytestmark = pytest.mark.gpu

PROBES = (0, 341, 682, 1023)
HW = 224
_cache = {}


def _weights():
    if "vsd" not in _cache:
        from feature_vs_text_compound_emotion_amd import synth
        _cache["vsd"] = synth.make_state_dict(synth.visual_backbone_spec("", HW // 8), seed=11)
    return _cache["vsd"]


def _probe_frame(i):
    return torch.randn(3, HW, HW, generator=torch.Generator().manual_seed(90000 + i))


def _oracle_embeddings():
    if "ref" not in _cache:
        import oracle
        frames = torch.stack([_probe_frame(i) for i in PROBES])
        with torch.no_grad():
            _cache["ref"] = oracle.ir50_forward(frames, _weights(), "backbone.")
    return _cache["ref"]


def _big_batch(n, probes):
    g = torch.Generator(device="cuda").manual_seed(5)
    x = torch.randn(n, 3, HW, HW, device="cuda", generator=g)
    for i in probes:
        x[i] = _probe_frame(i % 1024).cuda()
    return x


@pytest.mark.parametrize("precision,n,tol", [("bf16x3", 1024, 1e-4), ("fp16", 1024, 4e-3), ("bf16", 2048, 2.5e-2)])
def test_eval_encoder_at_the_timed_batch_matches_the_oracle_on_probe_frames(precision, n, tol):
    from feature_vs_text_compound_emotion_amd.visual_backbone import VisualBackbone
    vb = VisualBackbone(use_pretrained=False, head_hw=HW // 8)
    vb.load_state_dict(_weights(), strict=True)
    vb.backbone.precision = precision
    vb = vb.cuda().eval()
    probes = list(PROBES) + ([1024 + p for p in PROBES] if n > 1024 else [])
    x = _big_batch(n, probes)
    with torch.no_grad():
        emb = vb(x)
    del x
    got = emb[probes].cpu()
    ref = _oracle_embeddings()
    ref = torch.cat([ref] * (len(probes) // len(PROBES)))
    err = (got - ref).abs().max().item()
    print(f"\n[at size] {precision} N={n} x {HW}x{HW}: max |emb err| on the probe frames {err:.2e}")
    assert err < tol
    assert torch.isfinite(emb).all() and (emb.norm(dim=1) - 1).abs().max().item() < 1e-4
    del vb, emb
    torch.cuda.empty_cache()


@pytest.mark.parametrize("mode", ["bf16x3", "bf16", "fp16"])
def test_single_64_to_64_layer_at_n1024_on_sampled_pixels_of_the_last_frames(mode):
    """64 -> 64 3x3 @224x224, N = 1024: 6.6 GB per 16-bit plane.  Reference = float64 dot products of the gathered 3x3x64
    input patches (the operands the kernel sees) for 256 sampled output pixels of the last 64 frames + image corners."""
    from feature_vs_text_compound_emotion_amd import ops
    n, c = 1024, 64
    g = torch.Generator(device="cuda").manual_seed(77)
    w = torch.randn(c, c, 3, 3, generator=torch.Generator().manual_seed(3)) / (c * 9) ** 0.5
    wp = ops.pack_conv_weight(w.cuda())
    x = torch.randn(n, HW, HW, c, device="cuda", generator=g)
    if mode == "bf16x3":
        xs, ws = ops.split_bf16(x), ops.split_bf16(wp)
        del x
        y = ops.conv2d_b3(xs, ws, 3, 3, pad=(1, 1), out_f32=False, out_split=True)["split"]
        x_val = lambda nn, yy, xx: (xs.hi[nn, yy, xx].double() + xs.lo[nn, yy, xx].double())  # noqa: E731
        y_val = lambda nn, yy, xx: (y.hi[nn, yy, xx].double() + y.lo[nn, yy, xx].double())     # noqa: E731
        w_eff = (ws.hi.double() + ws.lo.double()).cpu()
        rel = 2.0 ** -14
    else:
        dt = torch.bfloat16 if mode == "bf16" else torch.float16
        xs, ws = ops.to_n16(x, dt), ops.to_n16(wp, dt)
        del x
        y = ops.conv2d_n16(xs, ws, 3, 3, pad=(1, 1))["n16"]
        x_val = lambda nn, yy, xx: xs[nn, yy, xx].double()   # noqa: E731
        y_val = lambda nn, yy, xx: y[nn, yy, xx].double()    # noqa: E731
        w_eff = ws.double().cpu()
        rel = 2.0 ** -8 if mode == "bf16" else 2.0 ** -11    # the output is rounded once to the storage type
    gen = torch.Generator().manual_seed(9)
    pts = [(n - 1, HW - 1, HW - 1), (n - 1, 0, 0), (n - 1, HW - 1, 0), (n - 64, 0, HW - 1), (0, 0, 0), (511, 100, 223)]
    pts += [(int(n - 64 + torch.randint(0, 64, (1,), generator=gen)), int(torch.randint(0, HW, (1,), generator=gen)),
             int(torch.randint(0, HW, (1,), generator=gen))) for _ in range(250)]
    worst = 0.0
    for (nn, yy, xx) in pts:
        patch = torch.zeros(3, 3, c, dtype=torch.float64)
        for kh in range(3):
            for kw in range(3):
                iy, ix = yy + kh - 1, xx + kw - 1
                if 0 <= iy < HW and 0 <= ix < HW:
                    patch[kh, kw] = x_val(nn, iy, ix).cpu()
        ref = w_eff[:, :576] @ patch.reshape(-1)          # k = (kh*3+kw)*Cin + c
        got = y_val(nn, yy, xx).cpu()
        mag = (w_eff[:, :576].abs() @ patch.reshape(-1).abs())
        worst = max(worst, ((got - ref).abs() / (mag * rel + 1e-6)).max().item())
    print(f"\n[at size] 64->64 @224 N=1024 {mode}: worst error / bound {worst:.3f}")
    assert worst < 1.0
    del xs, y
    torch.cuda.empty_cache()


@pytest.mark.parametrize("mt", [False, True])
def test_jmt_fusion_at_cfg3_size_1024_tokens(mt):
    """BASELINE cfg3: B = 32 clips x L = 32 frames -> the final self-attention of JMT / MT runs over L*B = 1024 tokens x 6
    (2) stack slots (reference models/model.py:965-972).  Forward vs the oracle everywhere; gradients with a BULK bound
    (all but a counted handful of elements within 1e-5 + 1e-4 relative) instead of a loose max-norm: a ReLU pre-activation
    within rounding of zero flips its derivative between two correct fp32 evaluations, which moves single elements."""
    from feature_vs_text_compound_emotion_amd import synth
    from feature_vs_text_compound_emotion_amd.fusion_heads import JMTFusion, MTFusion
    from oracle.jmt import jmt_fusion
    mods = ["video", "vggish"]
    name = "MT" if mt else "JMT"
    spec, alias = synth.jmt_spec(mods, name)
    sd = synth.make_state_dict(spec, alias, seed=71)
    fsd = {k[len("fuse."):]: v for k, v in sd.items() if k.startswith("fuse.")}
    fuse = (MTFusion() if mt else JMTFusion())
    fuse.load_state_dict(fsd, strict=True)
    fuse = fuse.cuda()
    bsz, length = 32, 32
    g = torch.Generator().manual_seed(73)
    v = torch.randn(bsz, 128, length, generator=g, requires_grad=True)
    a = torch.randn(bsz, 64, length, generator=g, requires_grad=True)
    ref = jmt_fusion({"video": v, "vggish": a}, sd, "fuse.", mt=mt)  # [B, L, 128]
    go = torch.randn(ref.shape, generator=g)
    ref.backward(go)
    vr = v.detach().transpose(1, 2).reshape(bsz * length, 128).contiguous().cuda().requires_grad_(True)
    ar = a.detach().transpose(1, 2).reshape(bsz * length, 64).contiguous().cuda().requires_grad_(True)
    out = fuse.forward_rows(vr, ar, bsz, length)
    err_out = (out.detach().cpu().view(bsz, length, 128) - ref.detach()).abs().max().item()
    out.backward(go.reshape(bsz * length, 128).cuda())
    dv = vr.grad.cpu().view(bsz, length, 128).transpose(1, 2)
    da = ar.grad.cpu().view(bsz, length, 64).transpose(1, 2)
    outliers = 0
    for got, want in ((dv, v.grad), (da, a.grad)):
        bad = (got - want).abs() > (1e-5 + 1e-4 * want.abs())
        outliers += int(bad.sum())
        assert ((got - want).norm() / want.norm()).item() < 2e-4
    print(f"\n[at size] {name} fusion, 1024 tokens: max |out err| {err_out:.2e}, gradient elements outside "
          f"1e-5 + 1e-4 relative: {outliers} of {dv.numel() + da.numel()}")
    assert err_out < 2e-5
    assert outliers <= 8
