"""bf16x3 (split hi/lo) convolution on the bf16 matrix cores vs fp32 torch-CPU."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _setup(n, cin, cout, hw, k, seed):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(n, cin, hw, hw, generator=g)
    w = torch.randn(cout, cin, k, k, generator=g) / (cin * k * k) ** 0.5
    return x, w


def test_split_is_round_to_nearest_and_exact_to_16_bits():
    from feature_vs_text_compound_emotion_amd import ops
    x = torch.randn(4096, generator=torch.Generator().manual_seed(0)) * 37.0
    s = ops.split_bf16(x.cuda())
    assert torch.equal(s.hi.cpu(), x.bfloat16())
    assert torch.equal(s.lo.cpu(), (x - x.bfloat16().float()).bfloat16())
    assert ((s.float().cpu() - x).abs() / x.abs().clamp_min(1e-6)).max().item() < 2.0 ** -15


@pytest.mark.parametrize("n,cin,cout,hw,k,stride,tile", [
    (3, 64, 64, 12, 3, 1, 0), (4, 64, 128, 9, 1, 2, 0),
    # LDS-DMA staged tiles on v_mfma_f32_16x16x32_bf16 (zero padding through out-of-range buffer offsets, source-side swizzle)
    (2, 128, 256, 10, 3, 1, 41), (5, 64, 128, 9, 3, 2, 41), (3, 96, 100, 7, 3, 1, 41), (4, 64, 128, 9, 1, 2, 41),
    (3, 64, 64, 12, 3, 1, 42), (3, 64, 96, 7, 3, 1, 42), (2, 256, 256, 10, 3, 1, 44), (2, 128, 256, 10, 3, 2, 44),
    (3, 64, 96, 7, 3, 1, 45), (1, 32, 40, 5, 3, 1, 45), (7, 512, 512, 5, 3, 1, 41),
    # 1x1 stride-2 projection shortcut on the big tile (the round-1 case that missed a constant 3e-5 bound by 5 %)
    (4, 64, 256, 9, 1, 2, 41), (7, 256, 512, 5, 1, 2, 41),
    # 256x64 tile (Cout <= 64 at large M)
    (3, 64, 64, 12, 3, 1, 48), (3, 64, 40, 7, 3, 1, 48), (5, 128, 64, 9, 3, 2, 48), (4, 64, 64, 9, 1, 2, 48),
    # patch kernels (input window of a 16x16 output patch resident in LDS, ping-pong phases): borders on every side, several
    # patches per image, ragged Cout, 1 / 2 / 4 / 8 channel chunks of 32, two cout tiles per patch
    (2, 64, 64, 32, 3, 1, 59), (1, 64, 128, 48, 3, 1, 59), (2, 32, 40, 16, 3, 1, 59), (3, 128, 64, 16, 3, 1, 59),
    (2, 128, 128, 32, 3, 1, 58), (1, 128, 256, 32, 3, 1, 58), (1, 256, 128, 16, 3, 1, 58), (3, 64, 100, 32, 3, 1, 58),
    # 4 waves, one window buffer re-filled per chunk, two blocks per CU
    (1, 256, 64, 16, 3, 1, 59),
    # 1-D window kernels (any image size; 256 consecutive pixels span image rows and images): the reference's 40x40 crop
    # pyramid (40 / 20 / 10 / 5), the 224x224 pyramid's 56 / 28 / 14 / 7, tiny images (many per tile), ragged M and Cout,
    # the widest image the LDS takes (86)
    (3, 64, 64, 40, 3, 1, 53), (2, 64, 128, 20, 3, 1, 53), (5, 128, 40, 10, 3, 1, 53), (7, 64, 64, 5, 3, 1, 53),
    (1, 32, 64, 56, 3, 1, 53), (50, 64, 64, 2, 3, 1, 53), (3, 64, 64, 3, 3, 1, 53),
    (2, 128, 128, 56, 3, 1, 56), (3, 256, 256, 28, 3, 1, 56), (5, 128, 200, 14, 3, 1, 56), (9, 512, 512, 7, 3, 1, 56),
    (2, 64, 128, 40, 3, 1, 56), (1, 64, 128, 86, 3, 1, 56), (1, 32, 128, 17, 3, 1, 56),
    # 1-D window, 4 waves, one window buffer re-filled per chunk (two blocks per CU)
    (1, 64, 128, 86, 3, 1, 53),
    (9, 256, 64, 7, 3, 1, 53)])
def test_conv_b3_matches_fp32(n, cin, cout, hw, k, stride, tile):
    from feature_vs_text_compound_emotion_amd import ops
    x, w = _setup(n, cin, cout, hw, k, n * 100 + cin + cout)
    ref = F.conv2d(x.double(), w.double(), None, stride, k // 2)
    # per-output error bound derived from the algorithm instead of a constant: every product drops lo*lo (2^-18) and
    # rounds both lo parts to bf16 (2^-17 each): <= 2^-15 * sum_k |x_k| |w_k|, plus the fp32 accumulation (K * 2^-24)
    mag = F.conv2d(x.double().abs(), w.double().abs(), None, stride, k // 2)
    bound = mag * (2.0 ** -15 + cin * k * k * 2.0 ** -24) + 1e-7
    xs = ops.split_bf16(x.permute(0, 2, 3, 1).contiguous().cuda())
    ws = ops.split_bf16(ops.pack_conv_weight(w.cuda()))
    r = ops.conv2d_b3(xs, ws, k, k, stride=stride, pad=(k // 2, k // 2), tile=tile, out_f32=True, out_split=True)
    got = r["y"].cpu().permute(0, 3, 1, 2).double()
    assert ((got - ref).abs() <= bound).all(), ((got - ref).abs() / bound).max().item()
    assert (got - ref).abs().max().item() < 5e-5      # and in absolute terms on these O(1) outputs
    assert (r["split"].float().cpu().permute(0, 3, 1, 2).double() - ref).abs().max().item() < 8e-5


@pytest.mark.parametrize("tile", [0, 41, 42, 44, 45, 48])
def test_conv_b3_fused_epilogue_outputs(tile):
    from feature_vs_text_compound_emotion_amd import ops
    g = torch.Generator().manual_seed(5)
    n, cin, cout, hw = 2, 64, 128, 10
    x, w = _setup(n, cin, cout, hw, 3, 77)
    bias, alpha = torch.randn(cout, generator=g), torch.rand(cout, generator=g) * 0.3 + 0.1
    res = torch.randn(n, cout, hw, hw, generator=g)
    s2, t2 = torch.rand(cout, generator=g) + 0.5, torch.randn(cout, generator=g) * 0.2
    z = F.conv2d(x, w, None, 2, 1) + bias.view(1, -1, 1, 1)
    z = torch.where(z >= 0, z, z * alpha.view(1, -1, 1, 1)) + res[:, :, ::2, ::2]
    xs = ops.split_bf16(x.permute(0, 2, 3, 1).contiguous().cuda())
    ws = ops.split_bf16(ops.pack_conv_weight(w.cuda()))
    rs = ops.split_bf16(res.permute(0, 2, 3, 1).contiguous().cuda())
    r = ops.conv2d_b3(xs, ws, 3, 3, stride=2, pad=(1, 1), bias=bias.cuda(), alpha=alpha.cuda(), act1=ops.ACT_PRELU,
                      residual=rs, res_stride=2, out_f32=True, next_affine=(s2.cuda(), t2.cuda()), want_stats=True, tile=tile)
    assert (r["y"].cpu().permute(0, 3, 1, 2) - z).abs().max().item() < 1e-4
    nxt = z * s2.view(1, -1, 1, 1) + t2.view(1, -1, 1, 1)
    assert (r["next"].float().cpu().permute(0, 3, 1, 2) - nxt).abs().max().item() < 2e-4
    raw = F.conv2d(x, w, None, 2, 1)
    st = r["stats"].cpu().sum(0)
    assert (st[0] - raw.sum((0, 2, 3))).abs().max().item() < 1e-2
    assert (st[1] - (raw * raw).sum((0, 2, 3))).abs().max().item() < 1e-2


@pytest.mark.parametrize("cin,cout,tile", [(64, 64, 59), (64, 128, 58), (128, 200, 58), (64, 64, 0), (128, 128, 0),
                                           (64, 64, 53), (64, 128, 56), (128, 200, 56)])
def test_conv_b3_patch_kernel_epilogue_on_non_square_images(cin, cout, tile):
    """bf16x3 patch kernels: H != W, bias9 (folded input BatchNorm) + PReLU + split residual + statistics; tile 0 on a shape
    the picker routes to them."""
    from feature_vs_text_compound_emotion_amd import ops
    g = torch.Generator().manual_seed(cin + cout)
    n, h, w = (176, 32, 48) if tile == 0 else ((3, 13, 21) if tile in (53, 56) else (2, 32, 48))
    x = torch.randn(n, cin, h, w, generator=g)
    wt = torch.randn(cout, cin, 3, 3, generator=g) / (cin * 9) ** 0.5
    s1, t1 = torch.rand(cin, generator=g) + 0.5, torch.randn(cin, generator=g) * 0.5
    alpha = torch.rand(cout, generator=g) * 0.3 + 0.1
    res = torch.randn(n, cout, h, w, generator=g)
    raw = F.conv2d((x * s1.view(1, -1, 1, 1)).double(), wt.double(), None, 1, 1)
    z = F.conv2d((x * s1.view(1, -1, 1, 1) + t1.view(1, -1, 1, 1)).double(), wt.double(), None, 1, 1)
    ref = torch.where(z >= 0, z, z * alpha.double().view(1, -1, 1, 1)) + res.double()
    wp, b9 = ops.fold_bn_3x3_packed(ops.pack_conv_weight(wt.cuda()), s1.cuda(), t1.cuda(), "split")
    xs = ops.split_bf16(x.permute(0, 2, 3, 1).contiguous().cuda())
    rs = ops.split_bf16(res.permute(0, 2, 3, 1).contiguous().cuda())
    r = ops.conv2d_b3(xs, wp, 3, 3, pad=(1, 1), alpha=alpha.cuda(), act1=ops.ACT_PRELU, bias9=b9, residual=rs, tile=tile,
                      out_f32=True, want_stats=True)
    assert (r["y"].cpu().permute(0, 3, 1, 2).double() - ref).abs().max().item() < 1e-4
    st = r["stats"].cpu().double().sum(0)
    npix = float(n * h * w)   # fp32 partial sums over npix values of O(1): relative to the count
    assert (st[0] - raw.sum((0, 2, 3))).abs().max().item() < 2e-6 * npix + 1e-2
    assert (st[1] - (raw * raw).sum((0, 2, 3))).abs().max().item() < 2e-6 * npix + 1e-2
    if tile != 0:
        with pytest.raises(RuntimeError, match="window" if tile in (53, 56) else "patch"):
            ops.conv2d_b3(xs, wp, 3, 3, stride=2, pad=(1, 1), tile=tile)


def test_linear_b3_split_k():
    from feature_vs_text_compound_emotion_amd import ops
    g = torch.Generator().manual_seed(6)
    m, k, cout = 70, 1280, 512
    x, w, b = torch.randn(m, k, generator=g), torch.randn(cout, k, generator=g) / k ** 0.5, torch.randn(cout, generator=g)
    ref = F.linear(x, w, b)
    xs = ops.split_bf16(x.cuda().view(m, 1, 1, k))
    ws = ops.split_bf16(w.cuda().contiguous())
    for tile in (0, 41, 42, 44, 45):
        r = ops.conv2d_b3(xs, ws, 1, 1, bias=b.cuda(), split_k=5, out_f32=True, out_split=False, tile=tile)
        assert (r["y"].view(m, cout).cpu() - ref).abs().max().item() < 5e-5


@pytest.mark.parametrize("n,cin,cout,hw,tile", [(2, 64, 64, 9, 0), (3, 64, 128, 6, 41), (2, 128, 256, 5, 44), (1, 64, 100, 12, 42)])
def test_input_batchnorm_folded_into_the_conv(n, cin, cout, hw, tile):
    """conv3x3(pad0(s*x + t)) == conv3x3'(pad0(x)) + bias9[border case]: the pre-conv BatchNorm of an IR unit folded
    into the conv (ops.fold_input_bn_3x3) instead of a pass over the activations."""
    from feature_vs_text_compound_emotion_amd import ops
    g = torch.Generator().manual_seed(hw * 7 + cout)
    x, w = _setup(n, cin, cout, hw, 3, 11 * n + cout)
    s1, t1 = torch.rand(cin, generator=g) + 0.5, torch.randn(cin, generator=g) * 0.5
    alpha = torch.rand(cout, generator=g) * 0.3 + 0.1
    z = F.conv2d(x * s1.view(1, -1, 1, 1) + t1.view(1, -1, 1, 1), w, None, 1, 1)
    ref = torch.where(z >= 0, z, z * alpha.view(1, -1, 1, 1))
    wp, b9 = ops.fold_input_bn_3x3(w.cuda(), s1.cuda(), t1.cuda())
    xs = ops.split_bf16(x.permute(0, 2, 3, 1).contiguous().cuda())
    r = ops.conv2d_b3(xs, ops.split_bf16(wp), 3, 3, pad=(1, 1), alpha=alpha.cuda(), act1=ops.ACT_PRELU, bias9=b9, tile=tile,
                      out_f32=True, out_split=False)
    assert (r["y"].cpu().permute(0, 3, 1, 2) - ref).abs().max().item() < 6e-5
    with pytest.raises(RuntimeError, match="bias9"):
        ops.conv2d_b3(xs, ops.split_bf16(wp), 3, 3, stride=2, pad=(1, 1), bias9=b9)


def test_bn_apply_split_io():
    from feature_vs_text_compound_emotion_amd import ops
    g = torch.Generator().manual_seed(9)
    n, h, c = 3, 10, 64
    z, res = torch.randn(n, 5, 5, c, generator=g), torch.randn(n, h, h, c, generator=g)
    s, t = torch.rand(c, generator=g) + 0.5, torch.randn(c, generator=g)
    rs = ops.split_bf16(res.cuda())
    o = ops.bn_apply_nhwc_b3(z.cuda(), s.cuda(), t.cuda(), res=rs, res_stride=2, want_stats=True, out_f32=True)
    ref = z * s + t + rs.float().cpu()[:, ::2, ::2]
    assert torch.equal(o["y"].cpu(), ops.bn_apply_nhwc(z.cuda(), s.cuda(), t.cuda(), res=rs.float(), res_stride=2).cpu())
    assert (o["y"].cpu() - ref).abs().max().item() < 1e-5
    assert torch.equal(o["split"].hi.cpu(), o["y"].cpu().bfloat16())
    assert (o["split"].float().cpu() - o["y"].cpu()).abs().max().item() < 2e-4
    st = o["stats"].cpu().sum(0)
    assert (st[0] - ref.sum((0, 1, 2))).abs().max().item() < 1e-2


@pytest.mark.parametrize("tile,hw", [(56, 14), (58, 16), (41, 9), (53, 10), (59, 16)])
@pytest.mark.parametrize("kind", ["raw_f32", "b9_prelu_split", "bias_res_split"])
def test_specialised_row_epilogues_on_every_kernel_family(kind, tile, hw):
    """The straight-line row epilogues (conv_common.h: epi_mode / epi_row) are selected by the EXACT combination of outputs a
    launch asks for; here each of the three bf16x3 combinations alone, on the window, patch and flat kernels, against float64:
    raw conv result -> fp32 (+ statistics); border bias (folded input BatchNorm) + PReLU -> split; bias + same-geometry split
    residual -> split."""
    from feature_vs_text_compound_emotion_amd import ops
    cin, cout = 64, (64 if tile in (53, 59) else 128)
    n = 5
    g = torch.Generator().manual_seed(tile * 10 + len(kind))
    x = torch.randn(n, cin, hw, hw, generator=g)
    wt = torch.randn(cout, cin, 3, 3, generator=g) / (cin * 9) ** 0.5
    xs = ops.split_bf16(x.permute(0, 2, 3, 1).contiguous().cuda())
    mag = F.conv2d(x.double().abs(), wt.double().abs(), None, 1, 1)
    tol = mag * (2.0 ** -15 + cin * 9 * 2.0 ** -24) + 2e-6
    if kind == "raw_f32":
        ref = F.conv2d(x.double(), wt.double(), None, 1, 1)
        r = ops.conv2d_b3(xs, ops.split_bf16(ops.pack_conv_weight(wt.cuda())), 3, 3, pad=(1, 1), tile=tile, out_f32=True,
                          out_split=False, want_stats=True)
        got = r["y"].cpu().permute(0, 3, 1, 2).double()
        st = r["stats"].cpu().double().sum(0)
        assert (st[0] - ref.sum((0, 2, 3))).abs().max().item() < 1e-2 and (st[1] - (ref * ref).sum((0, 2, 3))).abs().max().item() < 1e-2
    elif kind == "b9_prelu_split":
        s1, t1 = torch.rand(cin, generator=g) + 0.5, torch.randn(cin, generator=g) * 0.5
        alpha = torch.rand(cout, generator=g) * 0.3 + 0.1
        z = F.conv2d((x * s1.view(1, -1, 1, 1) + t1.view(1, -1, 1, 1)).double(), wt.double(), None, 1, 1)
        ref = torch.where(z >= 0, z, z * alpha.double().view(1, -1, 1, 1))
        wp, b9 = ops.fold_bn_3x3_packed(ops.pack_conv_weight(wt.cuda()), s1.cuda(), t1.cuda(), "split")
        r = ops.conv2d_b3(xs, wp, 3, 3, pad=(1, 1), alpha=alpha.cuda(), act1=ops.ACT_PRELU, bias9=b9, tile=tile)
        got = r["split"].float().cpu().permute(0, 3, 1, 2).double()
        tol = tol * 2.5 + 2.0 ** -15 * ref.abs()      # |s1| <= 1.5 scales the products; the split output rounds once more
    else:
        bias = torch.randn(cout, generator=g)
        res = torch.randn(n, cout, hw, hw, generator=g)
        ref = F.conv2d(x.double(), wt.double(), bias.double(), 1, 1) + res.double()
        rs = ops.split_bf16(res.permute(0, 2, 3, 1).contiguous().cuda())
        r = ops.conv2d_b3(xs, ops.split_bf16(ops.pack_conv_weight(wt.cuda())), 3, 3, pad=(1, 1), bias=bias.cuda(), residual=rs,
                          tile=tile)
        got = r["split"].float().cpu().permute(0, 3, 1, 2).double()
        tol = tol + 2.0 ** -14 * (ref.abs() + res.double().abs())   # split residual in, split result out
    assert ((got - ref).abs() <= tol).all(), ((got - ref).abs() / tol).max().item()


# ---- space-to-depth hand-over of the stride-2 units: producer layout, then the window-resident stride-2 kernel ----
@pytest.mark.parametrize("n,cin,cout,h,w,tile", [(2, 64, 64, 32, 48, 59), (1, 64, 128, 16, 32, 58), (3, 64, 128, 20, 12, 56),
                                                 (5, 128, 40, 10, 6, 53), (12, 64, 64, 40, 40, 0)])
def test_producer_writes_the_space_to_depth_layout(n, cin, cout, h, w, tile):
    from feature_vs_text_compound_emotion_amd import ops
    g = torch.Generator().manual_seed(n + cin + h)
    x = torch.randn(n, h, w, cin, generator=g).cuda()
    wt = torch.randn(cout, cin, 3, 3, generator=g) / (9 * cin) ** 0.5
    b9, alpha = torch.randn(9, cout, generator=g).cuda(), (torch.rand(cout, generator=g) * 0.3 + 0.1).cuda()
    xs, ws = ops.split_bf16(x), ops.split_bf16(ops.pack_conv_weight(wt.cuda()))
    kw = dict(pad=(1, 1), bias9=b9, alpha=alpha, act1=ops.ACT_PRELU, tile=tile)
    plain = ops.conv2d_b3(xs, ws, 3, 3, **kw)["split"]
    s2d = ops.conv2d_b3(xs, ws, 3, 3, y_s2d=True, **kw)["split"]
    assert tuple(s2d.shape) == (n, h // 2, w // 2, 4 * cout)
    want = ops.space_to_depth(plain)
    assert torch.equal(s2d.hi, want.hi) and torch.equal(s2d.lo, want.lo)     # the same values, the stores permuted
    # and without a bias9 (the generic epilogue)
    plain = ops.conv2d_b3(xs, ws, 3, 3, pad=(1, 1), tile=tile)["split"]
    s2d = ops.conv2d_b3(xs, ws, 3, 3, pad=(1, 1), tile=tile, y_s2d=True)["split"]
    assert torch.equal(s2d.hi, ops.space_to_depth(plain.hi)) and torch.equal(s2d.lo, ops.space_to_depth(plain.lo))


@pytest.mark.parametrize("n,cin,cout,h,w", [
    (3, 64, 64, 16, 16), (2, 64, 128, 40, 40), (2, 128, 128, 20, 20), (1, 256, 200, 12, 10), (7, 512, 512, 10, 10),
    (1, 64, 64, 224, 224), (2, 128, 128, 112, 112), (1, 64, 40, 4, 252), (5, 128, 256, 6, 2), (9, 64, 64, 2, 2), (1, 64, 128, 34, 30)])
def test_stride2_conv_on_a_space_to_depth_input(n, cin, cout, h, w):
    from feature_vs_text_compound_emotion_amd import ops
    g = torch.Generator().manual_seed(n * 7 + cin + cout + h)
    x = torch.randn(n, cin, h, w, generator=g)
    wt = torch.randn(cout, cin, 3, 3, generator=g) / (9 * cin) ** 0.5
    ref = F.conv2d(x.double(), wt.double(), None, 2, 1)
    mag = F.conv2d(x.double().abs(), wt.double().abs(), None, 2, 1)
    bound = mag * (2.0 ** -15 + cin * 9 * 2.0 ** -24) + 1e-7
    xs = ops.split_bf16(x.permute(0, 2, 3, 1).contiguous().cuda())
    ws = ops.split_bf16(ops.pack_conv_weight(wt.cuda()))
    r = ops.conv2d_b3(ops.space_to_depth(xs), ops.pack_s2d_weight(ws, cin), 3, 3, stride=2, pad=(1, 1), x_s2d=True,
                      out_f32=True, out_split=False, want_stats=True)
    got = r["y"].cpu().permute(0, 3, 1, 2).double()
    assert tuple(got.shape) == tuple(ref.shape)
    assert ((got - ref).abs() <= bound).all(), ((got - ref).abs() / bound).max().item()
    flat = ops.conv2d_b3(xs, ws, 3, 3, stride=2, pad=(1, 1), out_f32=True, out_split=False)["y"]
    assert (flat - r["y"]).abs().max().item() < 2e-5            # the flat kernel: same products, another summation order
    st = r["stats"].double().cpu().sum(0)
    assert (st[0] - ref.sum((0, 2, 3))).abs().max().item() < 1e-2 * max(1.0, (n * h * w / 400) ** 0.5)
    assert (st[1] - (ref * ref).sum((0, 2, 3))).abs().max().item() < 1e-2 * max(1.0, n * h * w / 400)


def test_stride2_space_to_depth_conv_with_the_eval_epilogue_and_its_errors():
    from feature_vs_text_compound_emotion_amd import ops
    g = torch.Generator().manual_seed(11)
    n, cin, cout, h, w = 2, 64, 128, 20, 12
    x, wt = torch.randn(n, cin, h, w, generator=g), torch.randn(cout, cin, 3, 3, generator=g) / 24.0
    bias, res = torch.randn(cout, generator=g), torch.randn(n, cout, h, w, generator=g)
    z = F.conv2d(x, wt, None, 2, 1) + bias.view(1, -1, 1, 1) + res[:, :, ::2, ::2]
    xs = ops.space_to_depth(ops.split_bf16(x.permute(0, 2, 3, 1).contiguous().cuda()))
    ws = ops.pack_s2d_weight(ops.split_bf16(ops.pack_conv_weight(wt.cuda())), cin)
    rs = ops.split_bf16(res.permute(0, 2, 3, 1).contiguous().cuda())
    r = ops.conv2d_b3(xs, ws, 3, 3, stride=2, pad=(1, 1), x_s2d=True, bias=bias.cuda(), residual=rs, res_stride=2)["split"]
    assert (r.float().cpu().permute(0, 3, 1, 2) - z).abs().max().item() < 1e-4
    with pytest.raises(RuntimeError, match="space-to-depth"):      # a flat tile cannot read the layout
        ops.conv2d_b3(xs, ws, 3, 3, stride=2, pad=(1, 1), x_s2d=True, tile=41)
    x96 = ops.Split(xs.hi[..., :4 * 32].contiguous(), xs.lo[..., :4 * 32].contiguous())
    w96 = ops.Split(ws.hi[:, :9 * 32].contiguous(), ws.lo[:, :9 * 32].contiguous())
    with pytest.raises(RuntimeError, match="Cin % 64"):
        ops.conv2d_b3(x96, w96, 3, 3, stride=2, pad=(1, 1), x_s2d=True)
    plain = ops.split_bf16(x.permute(0, 2, 3, 1).contiguous().cuda())
    w1 = ops.split_bf16(ops.pack_conv_weight(torch.randn(cout, cin, 3, 3, generator=g).cuda()))
    with pytest.raises(RuntimeError, match="y_s2d"):               # flat tiles do not write it either
        ops.conv2d_b3(plain, w1, 3, 3, pad=(1, 1), y_s2d=True, tile=41)
