"""Parity of the implicit-GEMM conv kernel (through the C-ABI) against torch CPU fp32."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

TOL = 2e-4  # fp32 MFMA is an exact-fp32 fma chain; only summation order differs from the CPU


def _ops():
    from feature_vs_text_compound_emotion_amd import ops
    return ops


def _ref_conv(x_nchw, w, stride, pad, dil=1):
    return F.conv2d(x_nchw, w, None, stride, pad, dil)


@pytest.mark.parametrize("n,cin,cout,hw,k,stride,tile", [
    (3, 64, 64, 12, 3, 1, 0),
    (3, 64, 128, 12, 3, 2, 0),
    (2, 128, 256, 10, 3, 1, 1),
    (2, 128, 256, 10, 3, 1, 2),
    (2, 128, 256, 10, 3, 1, 4),
    (2, 128, 256, 10, 3, 1, 5),
    (5, 256, 512, 5, 3, 2, 0),
    (4, 64, 128, 9, 1, 2, 0),
    (2, 32, 7, 6, 1, 1, 0),     # Cout not a multiple of 4 (regressor-like)
    (7, 96, 100, 7, 3, 1, 0),   # ragged M and Cout
])
def test_conv_plain(n, cin, cout, hw, k, stride, tile):
    ops = _ops()
    g = torch.Generator().manual_seed(n * 1000 + cin + cout + hw + k)
    x = torch.randn(n, cin, hw, hw, generator=g)
    w = torch.randn(cout, cin, k, k, generator=g) / (cin * k * k) ** 0.5
    pad = k // 2
    ref = _ref_conv(x, w, stride, pad)
    xd = x.permute(0, 2, 3, 1).contiguous().cuda()
    wp = ops.pack_conv_weight(w.cuda())
    y = ops.conv2d(xd, wp, k, k, stride=stride, pad=(pad, pad), tile=tile)
    torch.cuda.synchronize()
    got = y.cpu().permute(0, 3, 1, 2)
    assert got.shape == ref.shape
    assert (got - ref).abs().max().item() < TOL


def test_conv_small_cin_nchw_stem():
    ops = _ops()
    g = torch.Generator().manual_seed(5)
    x = torch.randn(4, 3, 20, 20, generator=g)
    w = torch.randn(64, 3, 3, 3, generator=g) / 27 ** 0.5
    ref = _ref_conv(x, w, 1, 1)
    wp = ops.pack_conv_weight(w.cuda())
    y = ops.conv2d(x.cuda(), wp, 3, 3, pad=(1, 1), x_nchw=True)
    assert (y.cpu().permute(0, 3, 1, 2) - ref).abs().max().item() < TOL
    y2 = ops.conv2d(x.permute(0, 2, 3, 1).contiguous().cuda(), wp, 3, 3, pad=(1, 1))
    assert (y2.cpu().permute(0, 3, 1, 2) - ref).abs().max().item() < TOL


def test_conv_fused_prologue_epilogue():
    """in-affine on in-bounds pixels only, bias, PReLU, strided residual, second activation."""
    ops = _ops()
    g = torch.Generator().manual_seed(11)
    n, cin, cout, hw = 3, 64, 128, 10
    x = torch.randn(n, cin, hw, hw, generator=g)
    w = torch.randn(cout, cin, 3, 3, generator=g) / (cin * 9) ** 0.5
    sc = torch.rand(cin, generator=g) + 0.5
    sh = torch.randn(cin, generator=g) * 0.3
    bias = torch.randn(cout, generator=g)
    alpha = torch.rand(cout, generator=g) * 0.3 + 0.1
    res = torch.randn(n, cout, hw, hw, generator=g)
    xin = x * sc.view(1, -1, 1, 1) + sh.view(1, -1, 1, 1)
    r = _ref_conv(xin, w, 2, 1) + bias.view(1, -1, 1, 1)
    r = torch.where(r >= 0, r, r * alpha.view(1, -1, 1, 1))
    r = r + res[:, :, ::2, ::2]
    ref = F.leaky_relu(r, 0.01)
    y = ops.conv2d(x.permute(0, 2, 3, 1).contiguous().cuda(), ops.pack_conv_weight(w.cuda()), 3, 3, stride=2,
                   pad=(1, 1), in_scale=sc.cuda(), in_shift=sh.cuda(), bias=bias.cuda(), alpha=alpha.cuda(),
                   residual=res.permute(0, 2, 3, 1).contiguous().cuda(), res_stride=2, act1=ops.ACT_PRELU,
                   act2=ops.ACT_LEAKY)
    assert (y.cpu().permute(0, 3, 1, 2) - ref).abs().max().item() < TOL


def test_conv_bn_fold_scale():
    ops = _ops()
    g = torch.Generator().manual_seed(12)
    x = torch.randn(2, 64, 8, 8, generator=g)
    w = torch.randn(64, 64, 3, 3, generator=g) / 24.0
    s = torch.rand(64, generator=g) + 0.5
    ref = _ref_conv(x, w, 1, 1) * s.view(1, -1, 1, 1)
    wp = ops.pack_conv_weight(w.cuda(), out_scale=s.cuda())
    y = ops.conv2d(x.permute(0, 2, 3, 1).contiguous().cuda(), wp, 3, 3, pad=(1, 1))
    assert (y.cpu().permute(0, 3, 1, 2) - ref).abs().max().item() < TOL


@pytest.mark.parametrize("split_k", [2, 5, 16])
def test_linear_split_k(split_k):
    ops = _ops()
    g = torch.Generator().manual_seed(13)
    m, k, cout = 70, 1280, 512
    x = torch.randn(m, k, generator=g)
    w = torch.randn(cout, k, generator=g) / k ** 0.5
    b = torch.randn(cout, generator=g)
    ref = F.relu(F.linear(x, w, b))
    y = ops.linear(x.cuda(), w.cuda().contiguous(), bias=b.cuda(), act=ops.ACT_RELU, split_k=split_k)
    assert (y.cpu() - ref).abs().max().item() < TOL


def test_causal_dilated_conv1d_as_conv2d():
    """TCN conv: [B,L,C] viewed as an [B,L,1,C] image, KHx1 kernel, dilation d, left pad (k-1)d."""
    ops = _ops()
    g = torch.Generator().manual_seed(14)
    b, l, cin, cout, k, d = 3, 32, 128, 64, 5, 4
    x = torch.randn(b, cin, l, generator=g)
    w = torch.randn(cout, cin, k, generator=g) / (cin * k) ** 0.5
    bias = torch.randn(cout, generator=g)
    ref = F.conv1d(F.pad(x, ((k - 1) * d, 0)), w, bias, dilation=d)
    xd = x.transpose(1, 2).contiguous().cuda().view(b, l, 1, cin)
    wp = ops.pack_conv_weight(w.cuda().view(cout, cin, k, 1))
    y = ops.conv2d(xd, wp, k, 1, dil=(d, 1), pad=((k - 1) * d, 0), out_hw=(l, 1), bias=bias.cuda())
    got = y.view(b, l, cout).cpu().transpose(1, 2)
    assert (got - ref).abs().max().item() < TOL


def test_l2norm_and_maxpool():
    ops = _ops()
    g = torch.Generator().manual_seed(15)
    x = torch.randn(37, 512, generator=g)
    assert (ops.l2norm_rows(x.cuda()).cpu() - x / x.norm(dim=1, keepdim=True)).abs().max().item() < 1e-6
    im = torch.randn(3, 64, 12, 8, generator=g)
    ref = F.max_pool2d(im, 2, 2)
    got = ops.maxpool2x2_nhwc(im.permute(0, 2, 3, 1).contiguous().cuda()).cpu().permute(0, 3, 1, 2)
    assert torch.equal(got, ref)


def test_bad_arguments_raise():
    ops = _ops()
    x = torch.zeros(1, 4, 4, 64, device="cuda")
    w = torch.zeros(64, 576, device="cuda")
    with pytest.raises(RuntimeError):
        ops.conv2d(x, w, 3, 3, pad=(1, 1), act1=ops.ACT_PRELU)  # PReLU without alpha
    with pytest.raises(ValueError):
        ops.conv2d(x.cpu(), w, 3, 3)


def test_output_grid_that_outruns_the_input_is_rejected():
    """The staged kernels address rows by unsigned distances from the tile's first row; a forced out_hw beyond the natural
    output size would make them wrap (found as a GPU fault while building the conv data gradient): loud error instead."""
    from feature_vs_text_compound_emotion_amd import ops
    x = torch.randn(2, 5, 5, 64).cuda()
    w = ops.pack_conv_weight(torch.randn(64, 64, 3, 3).cuda())
    with pytest.raises(RuntimeError, match="outruns"):
        ops.conv2d(x, w, 3, 3, pad=(1, 1), out_hw=(6, 6))


# ---- the Cin = 3 input layer as a direct convolution (csrc/stem_conv.hip): statistics pass + apply pass ----
@pytest.mark.parametrize("n,h,w", [(3, 40, 40), (2, 224, 224), (5, 7, 13), (1, 2, 2), (2, 33, 50)])
def test_stem_conv_statistics_and_apply_passes(n, h, w):
    from feature_vs_text_compound_emotion_amd import ops
    g = torch.Generator().manual_seed(n + h)
    x = torch.randn(n, 3, h, w, generator=g)
    wt = torch.randn(64, 3, 3, 3, generator=g) / 27 ** 0.5
    scale, shift = torch.rand(64, generator=g) + 0.5, torch.randn(64, generator=g) * 0.3
    alpha = torch.rand(64, generator=g) * 0.3 + 0.1
    raw = F.conv2d(x.double(), wt.double(), None, 1, 1)
    xd, wd = x.cuda(), ops.pack_conv_weight(wt.cuda())
    st = ops.stem_conv(xd, wd)
    assert tuple(st.shape[1:]) == (2, 64)
    tot = st.double().cpu().sum(0)
    cnt = n * h * w
    assert (tot[0] - raw.sum((0, 2, 3))).abs().max().item() < 1e-5 * cnt ** 0.5 + 1e-4
    assert (tot[1] - (raw * raw).sum((0, 2, 3))).abs().max().item() < 1e-5 * cnt + 1e-4
    z = raw * scale.double().view(1, -1, 1, 1) + shift.double().view(1, -1, 1, 1)
    ref = torch.where(z >= 0, z, z * alpha.double().view(1, -1, 1, 1)).permute(0, 2, 3, 1)
    args = (xd, wd, scale.cuda(), shift.cuda(), alpha.cuda())
    r = ops.stem_conv(*args, out="f32", want_stats=True)
    assert (r["y"].double().cpu() - ref).abs().max().item() < 5e-6      # 27 fp32 multiply-adds per value
    so = r["stats"].double().cpu().sum(0)
    assert (so[0] - ref.sum((0, 1, 2))).abs().max().item() < 1e-5 * cnt ** 0.5 + 1e-4
    assert (so[1] - (ref * ref).sum((0, 1, 2))).abs().max().item() < 1e-5 * cnt + 1e-4
    # the other storages hold the same values, rounded once
    sp = ops.stem_conv(*args, out="split")["split"]
    want = ops.split_bf16(r["y"])
    assert torch.equal(sp.hi, want.hi) and torch.equal(sp.lo, want.lo)
    for dt in (torch.bfloat16, torch.float16):
        assert torch.equal(ops.stem_conv(*args, out=dt)["n16"], r["y"].to(dt))
    # eval form: no scale, shift = bias; and the implicit-GEMM kernel computes the same layer
    e = ops.stem_conv(xd, wd, None, shift.cuda(), alpha.cuda(), out="f32")["y"]
    ig = ops.conv2d(xd, wd, 3, 3, pad=(1, 1), bias=shift.cuda(), alpha=alpha.cuda(), act1=ops.ACT_PRELU, x_nchw=True)
    assert (e - ig).abs().max().item() < 5e-6


def test_stem_conv_argument_errors():
    from feature_vs_text_compound_emotion_amd import ops
    x, w = torch.randn(1, 3, 8, 8).cuda(), torch.randn(64, 32).cuda()
    with pytest.raises(RuntimeError, match="statistics pass"):
        ops.stem_conv(x, w, scale=torch.ones(64).cuda())           # scale without an output tensor
    with pytest.raises(ValueError, match="3 -> 64"):
        ops.stem_conv(torch.randn(1, 4, 8, 8).cuda(), w)
