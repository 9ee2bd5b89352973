"""Narrow (single bf16 / half plane) convolution, one MFMA per product with fp32 accumulation (csrc/conv_n16.hip),
vs torch-CPU float64 on the SAME rounded operands: the only differences left are the fp32 accumulation order and the
final rounding of the narrow output, so the bounds are tight and derived, not tuned."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

DTYPES = [torch.bfloat16, torch.float16]
EPS = {torch.bfloat16: 2.0 ** -8, torch.float16: 2.0 ** -11}   # half an ulp, relative


def _setup(n, cin, cout, hw, k, seed, dtype):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(n, cin, hw, hw, generator=g).to(dtype)
    w = (torch.randn(cout, cin, k, k, generator=g) / (cin * k * k) ** 0.5).to(dtype)
    return x, w


def _dev(x_nchw, w_oihw):
    from feature_vs_text_compound_emotion_amd import ops
    xd = x_nchw.permute(0, 2, 3, 1).contiguous().cuda()
    wd = ops.to_n16(ops.pack_conv_weight(w_oihw.float().cuda()), w_oihw.dtype)   # exact: the values are already narrow
    return xd, wd


def test_to_n16_is_round_to_nearest_even_and_from_n16_is_exact():
    from feature_vs_text_compound_emotion_amd import ops
    x = torch.randn(4096, generator=torch.Generator().manual_seed(0)) * 37.0
    s, t = torch.rand(64) + 0.5, torch.randn(64)
    for dt in DTYPES:
        y = ops.to_n16(x.cuda(), dt)
        assert torch.equal(y.cpu(), x.to(dt))
        assert torch.equal(ops.from_n16(y).cpu(), x.to(dt).float())
        ya = ops.to_n16(x.view(64, 64).cuda(), dt, scale=s.cuda(), shift=t.cuda())
        ref = torch.addcmul(t, x.view(64, 64), s)   # fused multiply-add like the kernel's v*s+t contraction, or ...
        alt = (x.view(64, 64) * s + t)              # ... two roundings
        got = ya.cpu()
        assert (torch.eq(got, ref.to(dt)) | torch.eq(got, alt.to(dt))).all()


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("n,cin,cout,hw,k,stride,tile", [
    (3, 64, 64, 12, 3, 1, 0), (4, 64, 128, 9, 1, 2, 0), (2, 128, 256, 10, 3, 1, 0), (7, 512, 512, 5, 3, 1, 0),
    # 256x256, 8 waves (two epilogue passes): several small images per tile, ragged M and Cout, stride 2, 1x1
    (4, 128, 256, 10, 3, 1, 91), (7, 256, 512, 5, 3, 2, 91), (3, 64, 300, 7, 3, 1, 91), (4, 64, 256, 9, 1, 2, 91),
    (1, 64, 256, 23, 3, 1, 91),
    # 256x64, 4 waves (two blocks per CU)
    (3, 64, 64, 12, 3, 1, 63), (3, 64, 40, 7, 3, 1, 63), (5, 128, 64, 9, 3, 2, 63), (4, 64, 64, 9, 1, 2, 63),
    # 128x128 / 128x64 / 64x64 / 64x128
    (2, 128, 256, 10, 3, 1, 64), (5, 64, 128, 9, 3, 2, 64), (3, 64, 100, 7, 3, 1, 64), (2, 128, 256, 10, 3, 1, 94),
    (4, 128, 128, 10, 3, 1, 94), (2, 256, 128, 13, 1, 1, 94),
    (3, 64, 64, 12, 3, 1, 65), (3, 64, 96, 7, 3, 1, 65),
    (3, 64, 96, 7, 3, 1, 66), (1, 64, 40, 5, 3, 1, 66),
    (2, 256, 256, 10, 3, 1, 67), (2, 128, 256, 10, 3, 2, 67),
    # patch kernels (input window of a 16x16 output patch resident in LDS across the nine taps): image borders on every
    # side of a patch, several patches per image, ragged Cout, 1 / 2 / 4 channel chunks, two cout tiles per patch; 78 = the
    # 128-cout tile with ping-pong phases
    (2, 64, 64, 32, 3, 1, 71), (1, 64, 128, 48, 3, 1, 71), (2, 64, 40, 16, 3, 1, 71), (3, 64, 64, 16, 3, 1, 71),
    (2, 128, 128, 32, 3, 1, 72), (1, 128, 256, 32, 3, 1, 72), (1, 256, 128, 16, 3, 1, 72), (3, 64, 100, 32, 3, 1, 72),
    (1, 192, 128, 48, 3, 1, 72), (2, 128, 128, 32, 3, 1, 78), (1, 256, 128, 16, 3, 1, 78), (3, 64, 100, 32, 3, 1, 78),
    # 1-D window kernels (any image size; 256 consecutive pixels span image rows and images): the 40x40 pyramid (40 / 20 / 10
    # / 5), the 224x224 pyramid's 56 / 28 / 14 / 7, tiny images (many per tile), ragged M and Cout, the widest image (86);
    # 76 = 128 couts with ping-pong phases
    (3, 64, 64, 40, 3, 1, 73), (2, 64, 128, 20, 3, 1, 73), (5, 128, 40, 10, 3, 1, 73), (7, 64, 64, 5, 3, 1, 73),
    (1, 64, 64, 56, 3, 1, 73), (50, 64, 64, 2, 3, 1, 73), (3, 64, 64, 3, 3, 1, 73),
    (2, 128, 128, 56, 3, 1, 76), (3, 256, 256, 28, 3, 1, 76), (5, 128, 200, 14, 3, 1, 76), (9, 512, 512, 7, 3, 1, 76),
    (2, 64, 128, 40, 3, 1, 76), (1, 64, 128, 86, 3, 1, 76), (1, 64, 128, 17, 3, 1, 76),
    # single-window tile (Cin == 64, 4 waves, two blocks per CU)
    (3, 64, 64, 40, 3, 1, 77), (7, 64, 64, 5, 3, 1, 77), (1, 64, 40, 56, 3, 1, 77), (50, 64, 64, 2, 3, 1, 77), (1, 64, 128, 86, 3, 1, 77)])
def test_conv_n16_matches_float64_on_the_same_operands(n, cin, cout, hw, k, stride, tile, dtype):
    from feature_vs_text_compound_emotion_amd import ops
    x, w = _setup(n, cin, cout, hw, k, n * 100 + cin + cout, dtype)
    ref = F.conv2d(x.double(), w.double(), None, stride, k // 2)
    mag = F.conv2d(x.double().abs(), w.double().abs(), None, stride, k // 2)
    xd, wd = _dev(x, w)
    r = ops.conv2d_n16(xd, wd, k, k, stride=stride, pad=(k // 2, k // 2), tile=tile, out_f32=True, out_n16=True, want_stats=True)
    got = r["y"].cpu().permute(0, 3, 1, 2).double()
    # products of two narrow values are exact in fp32; what is left is the fp32 accumulation: K * 2^-24 * sum |x||w|
    bound = mag * (cin * k * k * 2.0 ** -24) + 1e-9
    assert ((got - ref).abs() <= bound).all(), ((got - ref).abs() / bound).max().item()
    # the narrow output is the fp32 result rounded once
    assert torch.equal(r["n16"].cpu(), r["y"].cpu().to(dtype))
    st = r["stats"].cpu().double().sum(0)
    assert (st[0] - ref.sum((0, 2, 3))).abs().max().item() < 1e-2
    assert (st[1] - (ref * ref).sum((0, 2, 3))).abs().max().item() < 1e-2


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("cin,cout,tile", [(64, 64, 71), (64, 128, 71), (128, 128, 72), (64, 200, 72), (64, 64, 0), (128, 128, 0),
                                           (64, 64, 73), (64, 128, 76), (128, 200, 76), (128, 128, 78)])
def test_conv_n16_patch_kernel_epilogue_on_non_square_images(cin, cout, tile, dtype):
    """Patch kernels: H != W, bias9 (folded input BatchNorm) + PReLU + narrow residual + statistics; and the automatic
    choice on a shape the picker routes to them."""
    from feature_vs_text_compound_emotion_amd import ops
    g = torch.Generator().manual_seed(cin + cout)
    # tile 0: enough patches for the picker to choose a patch kernel; window kernels: odd sizes, several images per tile
    n, h, w = (176, 32, 48) if tile == 0 else ((3, 13, 21) if tile in (73, 76) else (2, 32, 48))
    x = torch.randn(n, cin, h, w, generator=g).to(dtype)
    wt = (torch.randn(cout, cin, 3, 3, generator=g) / (cin * 9) ** 0.5).to(dtype)
    s1, t1 = torch.rand(cin, generator=g) + 0.5, torch.randn(cin, generator=g) * 0.5
    alpha = torch.rand(cout, generator=g) * 0.3 + 0.1
    res = torch.randn(n, cout, h, w, generator=g).to(dtype)
    wp, b9 = ops.fold_input_bn_3x3(wt.float().cuda(), s1.cuda(), t1.cuda())
    wn = ops.to_n16(wp, dtype)
    w_eff = (wt.float() * s1.view(1, -1, 1, 1)).to(dtype).double()
    shift_img = torch.ones(1, cin, h, w, dtype=torch.float64) * t1.double().view(1, -1, 1, 1)
    raw = F.conv2d(x.double(), w_eff, None, 1, 1)
    z = raw + F.conv2d(shift_img, wt.double(), None, 1, 1)
    ref = torch.where(z >= 0, z, z * alpha.double().view(1, -1, 1, 1)) + res.double()
    xd = x.permute(0, 2, 3, 1).contiguous().cuda()
    rd = res.permute(0, 2, 3, 1).contiguous().cuda()
    r = ops.conv2d_n16(xd, wn, 3, 3, pad=(1, 1), alpha=alpha.cuda(), act1=ops.ACT_PRELU, bias9=b9, residual=rd, tile=tile,
                       out_f32=True, want_stats=True)
    assert (r["y"].cpu().permute(0, 3, 1, 2).double() - ref).abs().max().item() < 3e-5
    assert torch.equal(r["n16"].cpu(), r["y"].cpu().to(dtype))
    st = r["stats"].cpu().double().sum(0)
    npix = float(n * h * w)   # fp32 partial sums over npix values of O(1): relative to the count
    assert (st[0] - raw.sum((0, 2, 3))).abs().max().item() < 2e-6 * npix + 1e-2
    assert (st[1] - (raw * raw).sum((0, 2, 3))).abs().max().item() < 2e-6 * npix + 1e-2


def test_conv_n16_patch_kernel_rejects_what_it_cannot_take():
    from feature_vs_text_compound_emotion_amd import ops
    x = torch.zeros(1, 20, 20, 64, dtype=torch.bfloat16).cuda()
    w = torch.zeros(64, ops.conv_kpad(3, 3, 64), dtype=torch.bfloat16).cuda()
    with pytest.raises(RuntimeError, match="patch"):
        ops.conv2d_n16(x, w, 3, 3, pad=(1, 1), tile=71)          # 20 is not a multiple of 16
    x = torch.zeros(1, 32, 32, 128, dtype=torch.bfloat16).cuda()
    w = torch.zeros(64, ops.conv_kpad(3, 3, 128), dtype=torch.bfloat16).cuda()
    with pytest.raises(RuntimeError, match="patch"):
        ops.conv2d_n16(x, w, 3, 3, pad=(1, 1), tile=71)          # tile 71 keeps ONE window: Cin == 64 only
    with pytest.raises(RuntimeError, match="patch"):
        ops.conv2d_n16(x, w, 3, 3, stride=2, pad=(1, 1), tile=72)
    with pytest.raises(RuntimeError, match="window"):
        ops.conv2d_n16(x, w, 3, 3, stride=2, pad=(1, 1), tile=76)
    x = torch.zeros(1, 4, 100, 128, dtype=torch.bfloat16).cuda()
    with pytest.raises(RuntimeError, match="window"):
        ops.conv2d_n16(x, w, 3, 3, pad=(1, 1), tile=73)          # W = 100: two windows do not fit in the LDS


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("tile", [0, 91, 94, 64, 67])
def test_conv_n16_fused_epilogue(tile, dtype):
    """bias + PReLU + strided narrow residual, fp32 and narrow outputs, statistics of the raw conv."""
    from feature_vs_text_compound_emotion_amd import ops
    g = torch.Generator().manual_seed(5)
    n, cin, cout, hw = 2, 64, 128, 10
    x, w = _setup(n, cin, cout, hw, 3, 77, dtype)
    bias, alpha = torch.randn(cout, generator=g), torch.rand(cout, generator=g) * 0.3 + 0.1
    res = torch.randn(n, cout, hw, hw, generator=g).to(dtype)
    raw = F.conv2d(x.double(), w.double(), None, 2, 1)
    z = raw + bias.double().view(1, -1, 1, 1)
    z = torch.where(z >= 0, z, z * alpha.double().view(1, -1, 1, 1)) + res.double()[:, :, ::2, ::2]
    xd, wd = _dev(x, w)
    rd = res.permute(0, 2, 3, 1).contiguous().cuda()
    r = ops.conv2d_n16(xd, wd, 3, 3, stride=2, pad=(1, 1), bias=bias.cuda(), alpha=alpha.cuda(), act1=ops.ACT_PRELU,
                       residual=rd, res_stride=2, out_f32=True, want_stats=True, tile=tile)
    assert (r["y"].cpu().permute(0, 3, 1, 2).double() - z).abs().max().item() < 2e-5
    assert torch.equal(r["n16"].cpu(), r["y"].cpu().to(dtype))
    st = r["stats"].cpu().double().sum(0)
    assert (st[0] - raw.sum((0, 2, 3))).abs().max().item() < 1e-2
    assert (st[1] - (raw * raw).sum((0, 2, 3))).abs().max().item() < 1e-2
    # fp32 residual
    r2 = ops.conv2d_n16(xd, wd, 3, 3, stride=2, pad=(1, 1), bias=bias.cuda(), alpha=alpha.cuda(), act1=ops.ACT_PRELU,
                        residual=rd.float(), res_stride=2, out_f32=True, out_n16=False, tile=tile)
    assert torch.equal(r2["y"], r["y"])


@pytest.mark.parametrize("dtype", DTYPES)
def test_linear_n16_split_k(dtype):
    from feature_vs_text_compound_emotion_amd import ops
    g = torch.Generator().manual_seed(6)
    m, k, cout = 70, 1280, 512
    x = torch.randn(m, k, generator=g).to(dtype)
    w = (torch.randn(cout, k, generator=g) / k ** 0.5).to(dtype)
    b = torch.randn(cout, generator=g)
    ref = F.linear(x.double(), w.double(), b.double())
    for tile in (0, 91, 64, 66, 67):
        r = ops.conv2d_n16(x.cuda().view(m, 1, 1, k), w.cuda().contiguous(), 1, 1, bias=b.cuda(), split_k=5, out_f32=True,
                           out_n16=False, tile=tile)
        assert (r["y"].view(m, cout).cpu().double() - ref).abs().max().item() < 2e-5


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("n,cin,cout,hw,tile", [(2, 64, 64, 9, 0), (3, 64, 128, 6, 94), (2, 128, 256, 5, 91), (1, 64, 100, 12, 65),
                                                 (1, 64, 64, 37, 63)])
def test_input_batchnorm_folded_into_the_narrow_conv(n, cin, cout, hw, tile, dtype):
    """conv3x3(pad0(s*x + t)) == conv3x3'(pad0(x)) + bias9[border case] with w' = w*s rounded to the narrow type."""
    from feature_vs_text_compound_emotion_amd import ops
    g = torch.Generator().manual_seed(hw * 7 + cout)
    x, w = _setup(n, cin, cout, hw, 3, 11 * n + cout, dtype)
    s1, t1 = torch.rand(cin, generator=g) + 0.5, torch.randn(cin, generator=g) * 0.5
    alpha = torch.rand(cout, generator=g) * 0.3 + 0.1
    wp, b9 = ops.fold_input_bn_3x3(w.float().cuda(), s1.cuda(), t1.cuda())
    wn = ops.to_n16(wp, dtype)
    # reference on the operands the kernel really sees: narrow x, narrow(w*s), fp32 shift term
    w_eff = (w.float() * s1.view(1, -1, 1, 1)).to(dtype).double()
    shift_img = torch.ones(1, cin, hw, hw, dtype=torch.float64) * t1.double().view(1, -1, 1, 1)
    z = F.conv2d(x.double(), w_eff, None, 1, 1) + F.conv2d(shift_img, w.double(), None, 1, 1)
    ref = torch.where(z >= 0, z, z * alpha.double().view(1, -1, 1, 1))
    xd = x.permute(0, 2, 3, 1).contiguous().cuda()
    r = ops.conv2d_n16(xd, wn, 3, 3, pad=(1, 1), alpha=alpha.cuda(), act1=ops.ACT_PRELU, bias9=b9, tile=tile, out_f32=True,
                       out_n16=False)
    assert (r["y"].cpu().permute(0, 3, 1, 2).double() - ref).abs().max().item() < 3e-5
    with pytest.raises(RuntimeError, match="bias9"):
        ops.conv2d_n16(xd, wn, 3, 3, stride=2, pad=(1, 1), bias9=b9)


def test_conv_n16_rejects_what_it_cannot_take():
    from feature_vs_text_compound_emotion_amd import ops
    x = torch.zeros(1, 4, 4, 32, dtype=torch.bfloat16).cuda()
    w = torch.zeros(8, ops.conv_kpad(1, 1, 32), dtype=torch.bfloat16).cuda()
    with pytest.raises(RuntimeError, match="multiple of 64"):
        ops.conv2d_n16(x, w, 1, 1)
    with pytest.raises(ValueError):
        ops.conv2d_n16(x, w.half(), 1, 1)   # mixed storage types


@pytest.mark.parametrize("dtype", DTYPES)
def test_stem_fp32_kernel_with_narrow_output(dtype):
    """The Cin = 3 stem stays on the fp32 kernel (reads the caller's NCHW frames) and writes the narrow plane directly."""
    from feature_vs_text_compound_emotion_amd import ops
    g = torch.Generator().manual_seed(3)
    x, w = torch.randn(3, 3, 12, 12, generator=g), torch.randn(64, 3, 3, 3, generator=g) * 0.2
    b, a = torch.randn(64, generator=g), torch.rand(64, generator=g) * 0.3 + 0.1
    z = F.conv2d(x, w, b, 1, 1)
    ref = torch.where(z >= 0, z, z * a.view(1, -1, 1, 1)).permute(0, 2, 3, 1)
    r = ops.conv2d(x.cuda(), ops.pack_conv_weight(w.cuda()), 3, 3, pad=(1, 1), bias=b.cuda(), alpha=a.cuda(), act1=ops.ACT_PRELU,
                   x_nchw=True, out_n16=dtype)
    assert (r["y"].cpu() - ref).abs().max().item() < 2e-5
    assert torch.equal(r["n16"].cpu(), r["y"].cpu().to(dtype))


@pytest.mark.parametrize("dtype", DTYPES)
def test_bn_apply_narrow_io(dtype):
    from feature_vs_text_compound_emotion_amd import ops
    g = torch.Generator().manual_seed(9)
    n, h, c = 3, 10, 64
    z, res = torch.randn(n, 5, 5, c, generator=g), torch.randn(n, h, h, c, generator=g).to(dtype)
    s, t = torch.rand(c, generator=g) + 0.5, torch.randn(c, generator=g)
    alpha = torch.rand(c, generator=g) * 0.3 + 0.1
    rs, rt = torch.rand(c, generator=g) + 0.5, torch.randn(c, generator=g)
    for zin in (z, z.to(dtype)):   # conv result as fp32 or as a narrow plane
        zz = zin.double()
        o = ops.bn_apply_nhwc_n16(zin.cuda(), s.cuda(), t.cuda(), dtype=dtype, res=res.cuda(), res_stride=2, want_stats=True,
                                  out_f32=True)
        ref = zz * s.double() + t.double() + res.double()[:, ::2, ::2]
        assert (o["y"].cpu().double() - ref).abs().max().item() < 1e-5
        assert torch.equal(o["n16"].cpu(), o["y"].cpu().to(dtype))
        st = o["stats"].cpu().double().sum(0)
        assert (st[0] - ref.sum((0, 1, 2))).abs().max().item() < 1e-2
        assert (st[1] - (ref * ref).sum((0, 1, 2))).abs().max().item() < 1e-2
    # PReLU + mask + normalised projection shortcut (fp32 residual with its own affine)
    mask = (torch.rand(n, 5, 5, c, generator=g) > 0.4).float() / 0.6
    r32 = torch.randn(n, 5, 5, c, generator=g)
    o = ops.bn_apply_nhwc_n16(z.cuda(), s.cuda(), t.cuda(), dtype=dtype, alpha=alpha.cuda(), res=r32.cuda(), res_scale=rs.cuda(),
                              res_shift=rt.cuda(), mask=mask.cuda(), out_f32=True, out_n16=False)
    v = z * s + t
    ref = torch.where(v >= 0, v, v * alpha) * mask + (r32 * rs + rt)
    assert (o["y"].cpu() - ref).abs().max().item() < 1e-5


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("tile,hw", [(76, 14), (78, 16), (72, 16), (91, 9), (73, 10), (71, 16), (79, 48)])
@pytest.mark.parametrize("kind", ["raw_n16", "b9_prelu_n16", "bias_res_n16"])
def test_specialised_row_epilogues_on_every_narrow_kernel_family(kind, tile, hw, dtype):
    """The three narrow launch kinds of the encoder, each alone, on the window, patch and flat kernels: the narrow output must
    be the float64 result of the same (already rounded) operands, rounded once -- up to the fp32 accumulation."""
    from feature_vs_text_compound_emotion_amd import ops
    cin, cout = 64, (64 if tile in (73, 71, 79) else 128)
    n = 5
    g = torch.Generator().manual_seed(tile * 10 + len(kind))
    x = torch.randn(n, cin, hw, hw, generator=g).to(dtype)
    wt = (torch.randn(cout, cin, 3, 3, generator=g) / (cin * 9) ** 0.5).to(dtype)
    xd = x.permute(0, 2, 3, 1).contiguous().cuda()
    ulp = 2.0 ** (-8 if dtype == torch.bfloat16 else -11)
    if kind == "raw_n16":
        ref = F.conv2d(x.double(), wt.double(), None, 1, 1)
        wn = ops.to_n16(ops.pack_conv_weight(wt.float().cuda()), dtype)
        r = ops.conv2d_n16(xd, wn, 3, 3, pad=(1, 1), tile=tile, want_stats=True)
        st = r["stats"].cpu().double().sum(0)
        assert (st[0] - ref.sum((0, 2, 3))).abs().max().item() < 1e-2
    elif kind == "b9_prelu_n16":
        s1, t1 = torch.rand(cin, generator=g) + 0.5, torch.randn(cin, generator=g) * 0.5
        alpha = torch.rand(cout, generator=g) * 0.3 + 0.1
        wp, b9 = ops.fold_input_bn_3x3(wt.float().cuda(), s1.cuda(), t1.cuda())
        wn = ops.to_n16(wp, dtype)
        w_eff = (wt.float() * s1.view(1, -1, 1, 1)).to(dtype).double()
        shift_img = torch.ones(1, cin, hw, hw, dtype=torch.float64) * t1.double().view(1, -1, 1, 1)
        z = F.conv2d(x.double(), w_eff, None, 1, 1) + F.conv2d(shift_img, wt.double(), None, 1, 1)
        ref = torch.where(z >= 0, z, z * alpha.double().view(1, -1, 1, 1))
        r = ops.conv2d_n16(xd, wn, 3, 3, pad=(1, 1), alpha=alpha.cuda(), act1=ops.ACT_PRELU, bias9=b9, tile=tile)
    else:
        bias = torch.randn(cout, generator=g)
        res = torch.randn(n, cout, hw, hw, generator=g).to(dtype)
        ref = F.conv2d(x.double(), wt.double(), bias.double(), 1, 1) + res.double()
        wn = ops.to_n16(ops.pack_conv_weight(wt.float().cuda()), dtype)
        r = ops.conv2d_n16(xd, wn, 3, 3, pad=(1, 1), bias=bias.cuda(), residual=res.permute(0, 2, 3, 1).contiguous().cuda(),
                           tile=tile)
    got = r["n16"].float().cpu().permute(0, 3, 1, 2).double()
    assert ((got - ref).abs() <= ulp * ref.abs() + 3e-5).all(), ((got - ref).abs() - ulp * ref.abs()).max().item()


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("n,h,w,cout", [(3, 256, 256, 64), (3, 256, 256, 128), (1, 64, 80, 256), (7, 16, 16, 64)])
@pytest.mark.parametrize("kind", ["raw_n16", "raw_f32", "b9_prelu_n16", "bias_res_n16", "b9_prelu_s2d"])
def test_persistent_patch_kernel_equals_the_one_patch_per_block_kernel(kind, n, h, w, cout, dtype):
    """Tile 79 (conv_n16_p64.hip: one block per CU walks its patches, the next window prefetched, the previous patch's
    epilogue under the current patch's MFMAs) against tile 71 (validated against float64 above): the same MFMA order per
    accumulator, so the outputs are bit-identical; the per-patch batch statistics are summed in another order.  Sizes: 768
    patches (3 .. 6 per block, every body variant of the walk, two cout tiles), fewer patches than blocks, four cout tiles."""
    from feature_vs_text_compound_emotion_amd import ops
    cin = 64
    g = torch.Generator().manual_seed(n * 3 + h + cout + len(kind))
    x = torch.randn(n, h, w, cin, generator=g).to(dtype).cuda()
    wt = (torch.randn(cout, cin, 3, 3, generator=g) / (9 * cin) ** 0.5).to(dtype)
    wd = ops.to_n16(ops.pack_conv_weight(wt.float().cuda()), dtype)
    if kind == "raw_n16":
        kw = dict(want_stats=True)
    elif kind == "raw_f32":
        kw = dict(want_stats=True, out_f32=True, out_n16=False)
    elif kind.startswith("b9_prelu"):
        kw = dict(bias9=torch.randn(9, cout, generator=g).cuda(), alpha=(torch.rand(cout, generator=g) * 0.3 + 0.1).cuda(),
                  act1=ops.ACT_PRELU, y_s2d=kind.endswith("s2d"))
    else:
        kw = dict(bias=torch.randn(cout, generator=g).cuda(), residual=torch.randn(n, h, w, cout, generator=g).to(dtype).cuda())
    a = ops.conv2d_n16(x, wd, 3, 3, pad=(1, 1), tile=71, **kw)
    b = ops.conv2d_n16(x, wd, 3, 3, pad=(1, 1), tile=79, **kw)
    key = "y" if kind == "raw_f32" else "n16"
    assert torch.equal(a[key], b[key])
    if "stats" in a and a["stats"] is not None:
        # (the consumers sum the per-patch rows; the persistent kernel keeps one running sum per block and writes zero rows besides)
        assert tuple(a["stats"].shape) == tuple(b["stats"].shape)
        sa, sb = a["stats"].double().sum(0), b["stats"].double().sum(0)
        assert (sa - sb).abs().max().item() <= 2e-6 * sa.abs().max().item() * max(1.0, (n * h * w / 4096) ** 0.5)
    with pytest.raises(RuntimeError, match="specialised epilogues"):     # both outputs at once is the generic epilogue
        ops.conv2d_n16(x, wd, 3, 3, pad=(1, 1), tile=79, out_f32=True, out_n16=True)


# ---- space-to-depth hand-over of the stride-2 units (narrow twins of the bf16x3 tests) ----
@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("n,cin,cout,h,w,tile", [(2, 64, 64, 32, 48, 71), (1, 128, 128, 16, 32, 72), (1, 128, 128, 16, 32, 78),
                                                 (3, 64, 128, 20, 12, 76), (5, 128, 40, 10, 6, 73), (2, 64, 64, 14, 6, 77)])
def test_narrow_producer_writes_the_space_to_depth_layout(n, cin, cout, h, w, tile, dtype):
    from feature_vs_text_compound_emotion_amd import ops
    g = torch.Generator().manual_seed(n + cin + h)
    x = torch.randn(n, h, w, cin, generator=g).to(dtype).cuda()
    wt = (torch.randn(cout, cin, 3, 3, generator=g) / (9 * cin) ** 0.5).to(dtype)
    b9, alpha = torch.randn(9, cout, generator=g).cuda(), (torch.rand(cout, generator=g) * 0.3 + 0.1).cuda()
    wd = ops.to_n16(ops.pack_conv_weight(wt.float().cuda()), dtype)
    kw = dict(pad=(1, 1), bias9=b9, alpha=alpha, act1=ops.ACT_PRELU, tile=tile)
    plain = ops.conv2d_n16(x, wd, 3, 3, **kw)["n16"]
    s2d = ops.conv2d_n16(x, wd, 3, 3, y_s2d=True, **kw)["n16"]
    assert tuple(s2d.shape) == (n, h // 2, w // 2, 4 * cout)
    assert torch.equal(s2d, ops.space_to_depth(plain))                   # the same values, the stores permuted


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("n,cin,cout,h,w", [
    (3, 128, 128, 16, 16), (2, 128, 128, 40, 40), (1, 256, 200, 12, 10), (7, 512, 512, 10, 10), (1, 128, 128, 224, 224),
    (1, 128, 40, 4, 252), (5, 128, 256, 6, 2), (9, 128, 64, 2, 2), (1, 256, 128, 34, 30)])
def test_narrow_stride2_conv_on_a_space_to_depth_input(n, cin, cout, h, w, dtype):
    from feature_vs_text_compound_emotion_amd import ops
    g = torch.Generator().manual_seed(n * 7 + cin + cout + h)
    x = torch.randn(n, cin, h, w, generator=g).to(dtype)
    wt = (torch.randn(cout, cin, 3, 3, generator=g) / (9 * cin) ** 0.5).to(dtype)
    ref = F.conv2d(x.double(), wt.double(), None, 2, 1)
    mag = F.conv2d(x.double().abs(), wt.double().abs(), None, 2, 1)
    xd, wd = _dev(x, wt)
    r = ops.conv2d_n16(ops.space_to_depth(xd), ops.pack_s2d_weight(wd, cin), 3, 3, stride=2, pad=(1, 1), x_s2d=True,
                       out_f32=True, out_n16=True, want_stats=True)
    got = r["y"].cpu().permute(0, 3, 1, 2).double()
    assert tuple(got.shape) == tuple(ref.shape)
    bound = mag * (cin * 9 * 2.0 ** -24) + 1e-9                          # exact products, fp32 accumulation
    assert ((got - ref).abs() <= bound).all(), ((got - ref).abs() / bound).max().item()
    assert torch.equal(r["n16"].cpu(), r["y"].cpu().to(dtype))
    st = r["stats"].cpu().double().sum(0)
    assert (st[0] - ref.sum((0, 2, 3))).abs().max().item() < 1e-2 * max(1.0, (n * h * w / 400) ** 0.5)
    assert (st[1] - (ref * ref).sum((0, 2, 3))).abs().max().item() < 1e-2 * max(1.0, n * h * w / 400)


def test_narrow_space_to_depth_errors_are_loud():
    from feature_vs_text_compound_emotion_amd import ops
    dt = torch.float16
    x = torch.randn(2, 10, 10, 4 * 64).to(dt).cuda()                     # Cin = 64: one chunk per phase is not supported
    w = torch.randn(64, 9 * 64).to(dt).cuda()
    with pytest.raises(RuntimeError, match="Cin % 128"):
        ops.conv2d_n16(x, w, 3, 3, stride=2, pad=(1, 1), x_s2d=True)
    x = torch.randn(2, 10, 10, 4 * 128).to(dt).cuda()
    w = torch.randn(64, 9 * 128).to(dt).cuda()
    with pytest.raises(RuntimeError, match="space-to-depth"):            # a flat tile cannot read the layout
        ops.conv2d_n16(x, w, 3, 3, stride=2, pad=(1, 1), x_s2d=True, tile=94)
    with pytest.raises(RuntimeError, match="y_s2d"):                     # ... or write it
        ops.conv2d_n16(torch.randn(2, 10, 10, 128).to(dt).cuda(), w, 3, 3, pad=(1, 1), y_s2d=True, tile=94)
