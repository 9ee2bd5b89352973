"""IR-ResNet50 vision encoder on the HIP implicit-GEMM kernels.

Host-side mirror of the reference's ``VisualBackbone`` / ``Backbone``
(models/backbone.py:69-130, models/arcface_model.py:44-60,95-151): same
constructor arguments, same ``state_dict`` keys (351 entries) and the same
``forward(x[N,3,H,W]) -> [N,512]`` unit-norm embedding, but the arithmetic runs
as NHWC fp32-MFMA convolutions with fused prologue/epilogue:

    unit:  t   = PReLU(conv3x3(BN1(x)))            BN1 = in-affine prologue, PReLU epilogue
           out = BN2(conv3x3_s(t)) + shortcut(x)   BN2 folded into the weights, residual epilogue
    head:  BN2d -> flatten -> Linear -> BN1d folded into one split-K GEMM, then x/||x||.

The torch.nn layers below only HOLD parameters (so keys, ``.to()``,
``deepcopy`` and ``requires_grad`` behave like the reference); their own
``forward`` is never called.
"""
import copy

import torch
from torch import nn

from . import ops
from .synth import ir50_units

BN_EPS = 1e-5


class _Unit(nn.Module):
    """Parameter holder for one bottleneck_IR unit (arcface_model.py:44-60)."""

    def __init__(self, cin, depth, stride):
        super().__init__()
        self.cin, self.depth, self.stride = cin, depth, stride
        if cin == depth:
            self.shortcut_layer = nn.MaxPool2d(1, stride)  # no parameters; == spatial subsample
        else:
            self.shortcut_layer = nn.Sequential(nn.Conv2d(cin, depth, 1, stride, bias=False), nn.BatchNorm2d(depth))
        self.res_layer = nn.Sequential(
            nn.BatchNorm2d(cin), nn.Conv2d(cin, depth, 3, 1, 1, bias=False), nn.PReLU(depth),
            nn.Conv2d(depth, depth, 3, stride, 1, bias=False), nn.BatchNorm2d(depth))


class _Flatten(nn.Module):
    pass


def _conv_prec(x, wp, kh, kw, stride, pad, prec, out_hw=None):
    """Dense NHWC fp32 in / out; the arithmetic of the encoder's precision mode: "bf16x3" (split operands, three MFMAs per
    product, <= 2^-15 relative per product), narrow storage ("bf16" / "fp16": operands rounded once, one MFMA), or the exact
    fp32 kernels.  Used by the released units' forward and data-gradient convs (the reference runs them in the same
    arithmetic as the rest of the encoder: fp32, or fp16 under --amp autocast)."""
    if prec == "bf16x3":   # x may be a Split already (the released units keep their conv inputs split for the weight gradient)
        xs = x if isinstance(x, ops.Split) else ops.split_bf16(x)
        return ops.conv2d_b3(xs, ops.split_bf16(wp), kh, kw, stride=stride, pad=pad, out_hw=out_hw, out_f32=True,
                             out_split=False)["y"]
    if isinstance(x, ops.Split):
        x = x.float()
    if prec in ("bf16", "fp16"):
        dt = torch.bfloat16 if prec == "bf16" else torch.float16
        return ops.conv2d_n16(ops.to_n16(x, dt), ops.to_n16(wp, dt), kh, kw, stride=stride, pad=pad, out_hw=out_hw, out_f32=True,
                              out_n16=False)["y"]
    return ops.conv2d(x, wp, kh, kw, stride=stride, pad=pad, out_hw=out_hw)


# stride-2 data gradient, per dimension (k = 3, pad 1: y[o] = sum_k x[2o + k - 1] w[k]; k = 1, pad 0: y[o] = x[2o] w[0]):
#   dx[2j]     = dy[j] w[1]                       -> input parity 0: one tap, at dy offset 0
#   dx[2j + 1] = dy[j] w[2] + dy[j + 1] w[0]      -> input parity 1: two taps, at dy offsets (0, 1)
_S2_TAPS = {3: ([1], [2, 0]), 1: ([0], [])}


def _conv_dgrad(dy, w_oihw, stride, pad, in_hw, prec="fp32"):
    """Data gradient of a conv in the encoder's arithmetic.  Stride 1: the forward kernel on the transposed + flipped filter
    with padding k-1-pad.  Stride 2 (the 3x3 / pad 1 second conv and the 1x1 shortcut of the first unit of a stage,
    arcface_model.py:38-41,46-52): the four input-pixel parities are four stride-1 convs of dy with the filter taps that can
    reach them -- 1x1, 1x2, 2x1 and 2x2 taps for a 3x3 filter (the decomposition conv_b3_s2d.hip runs forward), 9 tap-MACs
    per output pixel of dy like the forward conv, instead of a stride-1 conv over dy spread onto a 4x larger zero image (36)."""
    kh, kw = w_oihw.shape[2], w_oihw.shape[3]
    n, ho, wo, c = dy.shape
    if stride == 1:
        if (in_hw[0] - kh + 1 + 2 * pad, in_hw[1] - kw + 1 + 2 * pad) != (ho, wo):
            raise ValueError("stride-1 data gradient: dy does not have the conv's output geometry")
        wt = ops.pack_conv_weight(w_oihw.contiguous(), flip=True, transpose=True)
        return _conv_prec(dy, wt, kh, kw, 1, (kh - 1 - pad, kw - 1 - pad), prec)
    if stride != 2 or kh != kw or kh not in _S2_TAPS or pad != kh // 2:
        raise NotImplementedError("data gradient: stride 1, or stride 2 with a 3x3 / pad 1 or 1x1 / pad 0 filter (the IR-50 set)")
    hin, win = in_hw
    if ((hin + 2 * pad - kh) // 2 + 1, (win + 2 * pad - kw) // 2 + 1) != (ho, wo):
        raise ValueError("stride-2 data gradient: dy does not have the conv's output geometry")
    cin = w_oihw.shape[1]
    dev = dy.hi.device if isinstance(dy, ops.Split) else dy.device
    taps = _S2_TAPS[kh]
    dx = (torch.empty if kh == 3 else torch.zeros)((n, hin, win, cin), device=dev, dtype=torch.float32)
    for py in (0, 1):
        for px in (0, 1):
            th, tw = taps[py], taps[px]
            hp, wq = (hin - py + 1) // 2, (win - px + 1) // 2
            if not th or not tw or hp == 0 or wq == 0:
                continue
            sub = w_oihw[:, :, th][:, :, :, tw].contiguous()                      # [Cout, Cin, len(th), len(tw)]
            wt = ops.pack_conv_weight(sub, flip=False, transpose=True)
            dx[:, py::2, px::2] = _conv_prec(dy, wt, len(th), len(tw), 1, (0, 0), prec, out_hw=(hp, wq))
    return dx


def _prelu_bwd_chunked(dy, x, alpha, split_out=False, max_bytes=4 << 30):
    """``ops.prelu_bwd`` over frame chunks: the kernel writes a per-element slope-gradient term tensor before reducing it, which at
    1024 frames of 224x224 x 128 channels is a 26 GB transient next to five live tensors of that size; in chunks of
    ``max_bytes`` it is bounded, and the slope gradient is the sum of the chunk results (fixed order).  The chunks write
    straight into slices of the result (fp32, or the Split tensor the weight / data gradient kernels read)."""
    n = dy.shape[0]
    per = dy[0].numel() * 4
    step = max(1, min(n, max_bytes // max(per, 1)))
    if step >= n:
        return ops.prelu_bwd(dy, x, alpha, split_out=split_out)
    dx = ops.Split.empty(dy.shape, dy.device) if split_out else torch.empty_like(dy)
    da = None
    for i in range(0, n, step):
        o = ops.Split(dx.hi[i:i + step], dx.lo[i:i + step]) if split_out else dx[i:i + step]
        _, a = ops.prelu_bwd(dy[i:i + step], x[i:i + step], alpha, out=o, split_out=split_out)
        da = a if da is None else da + a
    return dx, da


_ZERO_B9 = {}


def _zero_bias9(cout, device):
    key = (cout, str(device))
    if key not in _ZERO_B9:
        _ZERO_B9[key] = torch.zeros((9, cout), device=device, dtype=torch.float32)
    return _ZERO_B9[key]


def _bn_affine_from_saved(save_mean, save_invstd, gamma, beta):
    scale = save_invstd * gamma
    return scale.contiguous(), (beta - save_mean * scale).contiguous()


class _ReleasedUnit(torch.autograd.Function):
    """One bottleneck_IR unit (arcface_model.py:44-60) in train mode WITH its backward, for the body groups of the
    reference's gradual release (base/parameter_control.py:55-103: stage 4, then half of stage 3) and, as an extension, the
    whole body (BASELINE configs[1]).  Row BatchNorm fwd/bwd, convs (forward and data gradient) in the encoder's precision
    mode, the matrix-core weight-gradient kernel, PReLU fwd/bwd.

    Activation memory: only the RAW tensors are kept for the backward -- the unit input ``x`` (the previous unit's output,
    kept anyway), the two raw conv results ``z1`` / ``z2`` (and the raw shortcut conv ``zs``), 12-16 bytes per unit element
    -- and the two conv INPUTS are recomputed there: ``BN1(x)`` from the saved batch statistics (one fused affine + split
    pass) and ``PReLU(z1)``.  Round 2 also kept both conv inputs as split tensors (+8 bytes per element: 218 -> 131 MB per
    224x224 frame over the whole encoder).

    ``memory == "recompute"`` (``IR50.activation_memory``, what BASELINE configs[1] at B = 32 x 32 frames of 224x224 needs:
    stage 1 of this IR-50 runs at the full 224x224, so even the raw tensors are 410 MB per frame = 420 GB per step): a unit
    keeps only its INPUT (125 MB per frame, 128 GB per step) and the backward runs the unit's two convs again from it with
    the saved batch statistics -- one extra forward of the encoder per step (+1/3 of the conv work) for 3.3x less
    activation memory.  The kernels are deterministic, so the rebuilt tensors and hence all gradients are BIT-IDENTICAL to
    the "raw" plan (tested).  ``"recompute16"`` halves that again by keeping the input normalised by its first BatchNorm as
    one fp16 plane (62 MB per frame); the 2^-11 rounding of that plane reaches the gradients amplified like every other
    per-operation error of this 24-unit batch-statistics backward (~20x: 1e-2 on the whole-encoder gradient against 2.6e-3)
    -- an explicit trade, held to its own bar in tests/test_head_release_gpu.py."""

    @staticmethod
    def forward(ctx, x, u, prec, memory, g1, b1, w1, a1, w2, g2, b2, ws, gs, bs):
        n, h, w, cin = x.shape
        bn1, bn2, s = u.res_layer[0], u.res_layer[4], u.stride
        b3 = prec == "bf16x3"
        # BatchNorm 1: statistics pass only; the normalisation is ONE affine (+ split) pass from them -- the same arithmetic the
        # backward uses to rebuild it, so the "recompute" memory plan reproduces z1 / z2 bit for bit
        sm1, si1 = ops.bn_rows_stats(x.view(-1, cin), bn1.running_mean, bn1.running_var, bn1.eps, bn1.momentum)
        sc1, sh1 = _bn_affine_from_saved(sm1, si1, g1.detach(), b1.detach())
        xb_k = ops.split_bf16(x, sc1, sh1) if b3 else torch.addcmul(sh1, x, sc1)
        if b3 and memory != "raw":   # z1 is not kept: PReLU and the split happen in the conv's epilogue (no z1 / t1 round trips)
            z1 = None
            # (a zero border-bias table selects the straight-line "bias9 + PReLU -> split" row epilogue of the window kernels)
            t1_k = ops.conv2d_b3(xb_k, ops.split_bf16(ops.pack_conv_weight(w1.detach().contiguous())), 3, 3, pad=(1, 1),
                                 bias9=_zero_bias9(w1.shape[0], x.device), alpha=a1.detach().contiguous(),
                                 act1=ops.ACT_PRELU)["split"]
            del xb_k
        else:
            z1 = _conv_prec(xb_k, ops.pack_conv_weight(w1.detach().contiguous()), 3, 3, 1, (1, 1), prec)
            del xb_k
            t1 = ops.prelu_fwd(z1, a1.detach().contiguous())
            t1_k = ops.split_bf16(t1) if b3 else t1
            del t1
        z2 = _conv_prec(t1_k, ops.pack_conv_weight(w2.detach().contiguous()), 3, 3, s, (1, 1), prec)
        del t1_k
        _, ho, wo, depth = z2.shape
        # BatchNorm 2 (+ the shortcut's BatchNorm) and the residual add in ONE pass over z2: statistics, then scale / shift
        sm2, si2 = ops.bn_rows_stats(z2.view(-1, depth), bn2.running_mean, bn2.running_var, bn2.eps, bn2.momentum)
        sc2, sh2 = _bn_affine_from_saved(sm2, si2, g2.detach(), b2.detach())
        zs = sms = sis = None
        if ws is not None:
            bns = u.shortcut_layer[1]
            zs = _conv_prec(x, ops.pack_conv_weight(ws.detach().contiguous()), 1, 1, s, (0, 0), prec)
            sms, sis = ops.bn_rows_stats(zs.view(-1, depth), bns.running_mean, bns.running_var, bns.eps, bns.momentum)
            scs, shs = _bn_affine_from_saved(sms, sis, gs.detach(), bs.detach())
            out = ops.bn_apply_nhwc(z2, sc2, sh2, res=zs, res_scale=scs, res_shift=shs)
        else:
            out = ops.bn_apply_nhwc(z2, sc2, sh2, res=x, res_stride=s)   # MaxPool2d(1, s) == subsample
        if memory == "recompute":      # keep the unit input only; z1 / z2 / zs are rebuilt from it in the backward (bit-identical)
            xk, z1, z2, zs = x, None, None, None
        elif memory == "recompute16":
            # keep ONE 16-bit plane of the unit input; z1 / z2 / zs are rebuilt from it in the backward.  What is kept is the
            # NORMALISED input (x - mean) * invstd of BatchNorm 1 -- zero mean, unit variance per channel, so fp16's 2^-11 is an
            # absolute 5e-4 on O(1) values -- not x itself: the residual stream's mean is many standard deviations wide after a
            # few units, and rounding x would cost the BatchNorm backward's x_hat that factor in accuracy (measured: 1.1e-2
            # on the whole-encoder gradient against 2.6e-3 with fp32 tensors)
            xk, z1, z2, zs = ops.to_n16(x, torch.float16, si1, (-sm1 * si1).contiguous()), None, None, None
        else:
            xk = x
        ctx.save_for_backward(xk, z1, z2, sm1, si1, sm2, si2, zs, sms, sis, g1.detach(), b1.detach(), w1.detach(),
                              a1.detach(), w2.detach(), g2.detach(), ws.detach() if ws is not None else None,
                              gs.detach() if gs is not None else None)
        ctx.stride = s
        ctx.prec = prec
        ctx.out_shape = tuple(out.shape)
        return out

    @staticmethod
    def backward(ctx, dout):
        x, z1, z2, sm1, si1, sm2, si2, zs, sms, sis, g1, b1, w1, a1, w2, g2, ws, gs = ctx.saved_tensors
        s, prec = ctx.stride, ctx.prec
        # fp16 storage has no range for UN-scaled gradients: a mean cross-entropy gradient divided over B*L rows and H*W
        # pixels sits at or below fp16's smallest subnormal (6e-8) and would flush to zero when the output gradient is
        # rounded to the storage type.  The reference always pairs fp16 with GradScaler (trainer.py:341,389); this package's
        # own Trainer has no scaler, so the data-gradient convs of an fp16 encoder run on the bf16x3 kernels (fp32 range,
        # 2^-15 per product) -- scaled or not, nothing underflows.
        dprec = "bf16x3" if prec == "fp16" else prec
        n, h, w, cin = x.shape              # (the saved tensor: x, or its normalised fp16 plane -- same shape)
        _, ho, wo, depth = ctx.out_shape
        dout = dout.contiguous()
        b3 = prec != "fp32"   # the weight gradients follow the convs onto the bf16x3 matrix-core kernel
        split = prec == "bf16x3"   # operands split once per tensor, shared by the weight and the data gradient
        recompute = z2 is None
        xh = None
        if recompute and x.dtype == torch.float16:   # "recompute16": the saved plane is the normalised input; BN1(x) = xh * gamma + beta
            xh = ops.from_n16(x)
            sc1, sh1 = g1.contiguous(), b1.contiguous()
            xb = ops.split_bf16(xh, sc1, sh1) if split else torch.addcmul(sh1, xh, sc1)
            x = torch.addcmul(sm1, xh, 1.0 / si1) if ws is not None else None       # only the projection shortcut reads x itself
        else:
            sc1, sh1 = _bn_affine_from_saved(sm1, si1, g1, b1)
            xb = ops.split_bf16(x, sc1, sh1) if split else torch.addcmul(sh1, x, sc1)   # BN1(x) (and split) in one pass
        if recompute:
            z1 = _conv_prec(xb, ops.pack_conv_weight(w1.contiguous()), 3, 3, 1, (1, 1), prec)
        # the second conv's input, rebuilt from the raw conv result (see the class docstring); split in the same pass
        t1 = ops.prelu_split(z1, a1.contiguous()) if split else ops.prelu_fwd(z1, a1.contiguous())
        if recompute:
            z2 = _conv_prec(t1, ops.pack_conv_weight(w2.contiguous()), 3, 3, s, (1, 1), prec)
            if ws is not None:
                zs = _conv_prec(x, ops.pack_conv_weight(ws.contiguous()), 1, 1, s, (0, 0), prec)
        dz2, dg2, db2 = ops.bn_rows_bwd(dout.view(-1, depth), z2.view(-1, depth), sm2, si2, g2, split_out=split)
        del z2
        dz2 = dz2.view(n, ho, wo, depth)
        dw2 = ops.conv2d_wgrad(dz2, t1, 3, 3, stride=s, pad=(1, 1), b3=b3)
        del t1
        dt1 = _conv_dgrad(dz2, w2, s, 1, (h, w), dprec)
        del dz2
        dz1, da1 = _prelu_bwd_chunked(dt1, z1, a1.contiguous(), split_out=split)
        del dt1, z1
        dw1 = ops.conv2d_wgrad(dz1, xb, 3, 3, stride=1, pad=(1, 1), b3=b3)
        del xb
        need_dx = ctx.needs_input_grad[0]
        dxb = _conv_dgrad(dz1, w1, 1, 1, (h, w), dprec)
        del dz1
        # the shortcut branch's share of the input gradient: dout itself (identity), or the projection's data gradient; it is
        # added inside the BatchNorm-1 backward pass (``add``) instead of a read-modify-write pass over dx afterwards
        dws = dgs = dbs = None
        addend = even = None
        if ws is not None:
            dzs, dgs, dbs = ops.bn_rows_bwd(dout.view(-1, depth), zs.view(-1, depth), sms, sis, gs)
            dzs = dzs.view(n, ho, wo, depth)
            if split and s == 1:
                dzs = ops.split_bf16(dzs)
            dws = ops.conv2d_wgrad(dzs, x, 1, 1, stride=s, pad=(0, 0), b3=b3)
            even = None
            if need_dx:
                if s == 1:
                    addend = _conv_dgrad(dzs, ws, 1, 0, (h, w), dprec)
                else:   # a 1x1 / stride-2 conv only reaches the even input pixels: a quarter-size result, added strided below
                    wt = ops.pack_conv_weight(ws.contiguous(), flip=False, transpose=True)
                    even = _conv_prec(dzs, wt, 1, 1, 1, (0, 0), dprec)
            del dzs
        elif need_dx and s == 1:
            addend = dout
        rows_ok = need_dx and addend is not None and cin % 4 == 0
        if xh is not None:   # x_hat is what was saved: mean 0, invstd 1, and the outer gamma * invstd factor as the "weight"
            dx, dg1, db1 = ops.bn_rows_bwd(dxb.view(-1, cin), xh.view(-1, cin), torch.zeros_like(sm1), torch.ones_like(si1),
                                           (g1 * si1).contiguous(), add=addend.view(-1, cin) if rows_ok else None)
            del xh
        else:
            dx, dg1, db1 = ops.bn_rows_bwd(dxb.view(-1, cin), x.view(-1, cin), sm1, si1, g1,
                                           add=addend.view(-1, cin) if rows_ok else None)
        del dxb
        dx = dx.view(n, h, w, cin)
        if need_dx and not rows_ok:
            if addend is not None:
                ops.add_inplace(dx, addend)
            elif ws is not None and s > 1:
                dx[:, ::2, ::2] += even
            elif ws is None and s > 1:
                dx[:, ::s, ::s] += dout
        return (dx if need_dx else None), None, None, None, dg1, db1, dw1, da1, dw2, dg2, db2, dws, dgs, dbs


class _ReleasedStem(torch.autograd.Function):
    """``input_layer`` (Conv2d(3, 64, 3, 1, 1, bias=False) -> BatchNorm2d -> PReLU, arcface_model.py:130-132) in train mode with
    its backward: the last piece of a backward through the WHOLE encoder (BASELINE configs[1]; the reference's own release
    schedule stops at half of stage 3).  The conv is the NCHW-reading fp32 kernel; its weight gradient runs on the bf16x3
    matrix-core kernel over a 4-channel NHWC copy of the frames (no data gradient: the frames need none).  Only the raw conv
    result is kept for the backward; the BatchNorm output the PReLU saw is recomputed from the saved statistics."""

    @staticmethod
    def forward(ctx, x, bn, w, g, b, a):
        n, _, h, wd = x.shape
        z = ops.conv2d(x.contiguous(), ops.pack_conv_weight(w.detach().contiguous()), 3, 3, pad=(1, 1), x_nchw=True)
        zb, sm, si = ops.bn_rows_fwd(z.view(-1, 64), g.detach(), b.detach(), bn.running_mean, bn.running_var, True, bn.eps,
                                     bn.momentum)
        y = ops.prelu_fwd(zb.view(n, h, wd, 64), a.detach().contiguous())
        ctx.save_for_backward(x, z, sm, si, g.detach(), b.detach(), a.detach())
        return y

    @staticmethod
    def backward(ctx, dy):
        x, z, sm, si, g, b, a = ctx.saved_tensors
        n, _, h, wd = x.shape
        sc, sh = _bn_affine_from_saved(sm, si, g, b)
        zb = torch.addcmul(sh, z, sc)
        dzb, da = ops.prelu_bwd(dy.contiguous(), zb, a.contiguous())
        del zb
        dz, dg, db = ops.bn_rows_bwd(dzb.view(-1, 64), z.view(-1, 64), sm, si, g)
        del dzb
        x4 = torch.zeros((n, h, wd, 4), device=x.device, dtype=torch.float32)
        x4[..., :3] = x.permute(0, 2, 3, 1)
        dw = ops.conv2d_wgrad(dz.view(n, h, wd, 64), x4, 3, 3, stride=1, pad=(1, 1), b3=True)[:, :3].contiguous()
        return None, None, dw, dg, db, da


class _ReleasedHead(torch.autograd.Function):
    """Forward + backward of the encoder's output layer in train mode, for the FIRST group of the reference's gradual
    release (base/parameter_control.py:55-103: parameters 4..9 of the visual encoder = ``output_layer``:
    BatchNorm2d(512) -> Dropout(0.4) -> flatten (c,h,w) -> Linear -> BatchNorm1d(512), then ``l2_norm``;
    arcface_model.py:133-137,147-151).  The body below it stays frozen, so no data gradient leaves this function.
    Everything runs on the exact-fp32 kernels (row BatchNorm fwd/bwd, igemm GEMMs, TN weight-gradient GEMM)."""

    @staticmethod
    def forward(ctx, y, mask, bn2, fc, bn1, w2, b2, wfc, bfc, w1, b1):
        n, h, w, c = y.shape
        rows = y.view(n * h * w, c)
        o2, sm2, si2 = ops.bn_rows_fwd(rows, w2.detach(), b2.detach(), bn2.running_mean, bn2.running_var, True, bn2.eps,
                                       bn2.momentum)
        hfeat = ops.act_mask_bwd(o2, o2, mask.view(n * h * w, c), slope=1.0) if mask is not None else o2  # o2 * mask
        k = h * w * c
        whwc = wfc.detach().view(wfc.shape[0], c, h * w).permute(0, 2, 1).contiguous().view(wfc.shape[0], k)  # (c,h,w)->(h,w,c)
        e = ops.linear(hfeat.view(n, k), whwc, bias=bfc.detach(), split_k=max(1, min(k // 32, 192 // ((n + 127) // 128))))
        e2, sm1, si1 = ops.bn_rows_fwd(e, w1.detach(), b1.detach(), bn1.running_mean, bn1.running_var, True, bn1.eps,
                                       bn1.momentum)
        ctx.save_for_backward(rows, mask, hfeat, e, e2, sm2, si2, sm1, si1, whwc, w2.detach(), w1.detach())
        ctx.dims = (n, h, w, c)
        return ops.l2norm_rows(e2)

    @staticmethod
    def backward(ctx, demb):
        rows, mask, hfeat, e, e2, sm2, si2, sm1, si1, whwc, w2, w1 = ctx.saved_tensors
        n, h, w, c = ctx.dims
        k = h * w * c
        de2 = ops.l2norm_rows_bwd(demb.contiguous(), e2)
        de, dw1, db1 = ops.bn_rows_bwd(de2, e, sm1, si1, w1)
        dwhwc = ops.conv1d_wgrad(de, hfeat.view(n, k), n, 1, 1).view(-1, h * w, c)      # dW[o][(h,w,c)] = de^T hfeat
        dwfc = dwhwc.permute(0, 2, 1).reshape(-1, k)                                     # back to the (c,h,w) flatten order
        dbfc = ops.col_sum(de)
        dh = ops.linear(de, whwc.t().contiguous())                                       # [n, k] = de @ W
        do2 = ops.act_mask_bwd(dh.view(n * h * w, c), dh.view(n * h * w, c), mask.view(n * h * w, c), slope=1.0) \
            if mask is not None else dh.view(n * h * w, c)
        dy, dw2, db2 = ops.bn_rows_bwd(do2, rows, sm2, si2, w2)
        dy = dy.view(n, h, w, c) if ctx.needs_input_grad[0] else None  # only when body units below are released too
        return dy, None, None, None, None, dw2, db2, dwfc, dbfc, dw1, db1


class IR50(nn.Module):
    """``Backbone(num_layers=50, mode='ir')`` with an h x w output head."""

    def __init__(self, input_channels=3, drop_ratio=0.4, head_hw=5, embedding_dim=512):
        super().__init__()
        self.head_hw = head_hw
        if input_channels != 3:
            raise NotImplementedError("the input layer kernel (csrc/stem_conv.hip) is the reference's 3-channel one")
        self.input_layer = nn.Sequential(nn.Conv2d(input_channels, 64, 3, 1, 1, bias=False), nn.BatchNorm2d(64),
                                         nn.PReLU(64))
        self.output_layer = nn.Sequential(nn.BatchNorm2d(embedding_dim), nn.Dropout(drop_ratio), _Flatten(),
                                          nn.Linear(embedding_dim * head_hw * head_hw, embedding_dim),
                                          nn.BatchNorm1d(embedding_dim))
        self.body = nn.Sequential(*[_Unit(*u) for u in ir50_units()])
        self._packed = None
        self._packed_key = None
        self._packed_train = None
        self._packed_train_key = None
        self.bn_mode = "reference"  # or "frozen": encoder BatchNorm/Dropout stay in eval behaviour under train()
        # "bf16x3": split hi/lo bf16 operands, 3 bf16 MFMAs per product (<= 2^-15 relative per product, logit error
        # ~1e-6, 2-2.5x faster than fp32); "fp32": the exact-fp32 MFMA kernels
        self.precision = "bf16x3"
        self._packed_b3 = None
        self._packed_b3_key = None
        self._packed_train_b3 = None
        self._packed_train_b3_key = None
        # "bf16" / "fp16": narrow storage -- ONE 16-bit plane per tensor, one MFMA per product, fp32 accumulate and fp32
        # epilogue arithmetic (csrc/conv_n16.hip): what the reference's --amp recipe computes (fp16 autocast,
        # trainer.py:341,367) and BASELINE cfg5's "bf16 storage / fp32 accumulate"
        # the reference trains under torch.cuda.amp.autocast when --amp is set (trainer.py:341,367): inside an autocast region
        # the encoder follows it onto the narrow kernels of the autocast dtype (float16 -> "fp16", bfloat16 -> "bf16")
        self.follow_autocast = True
        self._packed_n16 = None
        self._packed_n16_key = None
        self._packed_train_n16 = None
        self._packed_train_n16_key = None
        self.dropout_seed = 0
        self._dropout_calls = 0
        # released units (_ReleasedUnit): "raw" keeps the raw conv results for the backward, "recompute" only the unit inputs
        # (the unit's convs run again in the backward; gradients bit-identical to "raw"), "recompute16" the normalised
        # inputs as one fp16 plane (lossy); "auto" = "raw" when its ~410 MB per 224x224 frame fit the free device memory
        # with room for the backward's transients, else "recompute"
        self.activation_memory = "auto"

    def __deepcopy__(self, memo):
        """trainer.py:656,705 deep-copies the model: copy parameters/buffers, not the packed caches."""
        names = ("_packed", "_packed_train", "_packed_b3", "_packed_train_b3", "_packed_n16", "_packed_train_n16")
        caches = [getattr(self, n) for n in names]
        for n in names:
            setattr(self, n, None)
        try:
            new = self.__class__.__new__(self.__class__)
            memo[id(self)] = new
            new.__dict__ = copy.deepcopy(self.__dict__, memo)
        finally:
            for n, c in zip(names, caches):
                setattr(self, n, c)
        return new

    # ------------------------------------------------------------------ packing
    @staticmethod
    def _bn_affine(bn):
        s = bn.weight.detach() * torch.rsqrt(bn.running_var + BN_EPS)
        return s.contiguous(), (bn.bias.detach() - bn.running_mean * s).contiguous()

    def _state_key(self):
        return tuple((p.data_ptr(), p._version) for p in list(self.parameters()) + list(self.buffers()))

    def pack(self):
        """Fold eval-mode BatchNorms and lay weights out for the kernels (cached until a
        parameter or buffer changes)."""
        key = self._state_key()
        if self._packed is not None and key == self._packed_key:
            return self._packed
        dev = self.input_layer[0].weight.device
        if dev.type != "cuda":
            raise RuntimeError("IR50 runs on the HIP kernels only: move the module to a GPU (no CPU fallback)")
        P = {}
        s, b = self._bn_affine(self.input_layer[1])
        P["stem_w"] = ops.pack_conv_weight(self.input_layer[0].weight.detach().contiguous(), s)
        P["stem_b"], P["stem_a"] = b, self.input_layer[2].weight.detach().contiguous()
        units = []
        for u in self.body:
            d = {"stride": u.stride, "proj": u.cin != u.depth}
            d["in_s"], d["in_b"] = self._bn_affine(u.res_layer[0])
            d["w1"] = ops.pack_conv_weight(u.res_layer[1].weight.detach().contiguous())
            d["a1"] = u.res_layer[2].weight.detach().contiguous()
            s2, b2 = self._bn_affine(u.res_layer[4])
            d["w2"] = ops.pack_conv_weight(u.res_layer[3].weight.detach().contiguous(), s2)
            d["b2"] = b2
            if d["proj"]:
                ss, sb = self._bn_affine(u.shortcut_layer[1])
                d["ws"] = ops.pack_conv_weight(u.shortcut_layer[0].weight.detach().contiguous(), ss)
                d["bs"] = sb
            units.append(d)
        P["units"] = units
        # head: y = BN1d(W . flatten_chw(BN2d(x)) + b); fold both BNs, permute K to (h, w, c)
        hw = self.head_hw
        s0, t0 = self._bn_affine(self.output_layer[0])
        s4, t4 = self._bn_affine(self.output_layer[4])
        fc = self.output_layer[3]
        w = fc.weight.detach().view(fc.out_features, -1, hw * hw)  # [512, c, hw]
        bias = torch.mv(w.sum(dim=2), t0) + fc.bias.detach()
        w = (w * s0.view(1, -1, 1)).permute(0, 2, 1).contiguous().view(fc.out_features, -1)  # [512, (hw, c)]
        P["head_w"] = (w * s4.view(-1, 1)).contiguous()
        P["head_b"] = (bias * s4 + t4).contiguous()
        self._packed, self._packed_key = P, key
        return P

    def pack_train(self):
        """Raw (un-folded) kernel layouts for the batch-statistics path; weights are frozen, so this
        is cached on the parameter versions only."""
        key = tuple((p.data_ptr(), p._version) for p in self.parameters())
        if self._packed_train is not None and key == self._packed_train_key:
            return self._packed_train
        dev = self.input_layer[0].weight.device
        if dev.type != "cuda":
            raise RuntimeError("IR50 runs on the HIP kernels only: move the module to a GPU (no CPU fallback)")
        P = {"stem_w": ops.pack_conv_weight(self.input_layer[0].weight.detach().contiguous()), "units": []}
        for u in self.body:
            d = {"w1": ops.pack_conv_weight(u.res_layer[1].weight.detach().contiguous()),
                 "w2": ops.pack_conv_weight(u.res_layer[3].weight.detach().contiguous())}
            if u.cin != u.depth:
                d["ws"] = ops.pack_conv_weight(u.shortcut_layer[0].weight.detach().contiguous())
            P["units"].append(d)
        fc, hw = self.output_layer[3], self.head_hw
        P["head_w"] = fc.weight.detach().view(fc.out_features, -1, hw * hw).permute(0, 2, 1).contiguous().view(
            fc.out_features, -1)  # K order (c,h,w) -> (h,w,c)
        self._packed_train, self._packed_train_key = P, key
        return P

    # ------------------------------------------------------------------ bf16x3 packing
    def pack_b3(self):
        """Eval-mode layouts for the bf16x3 kernels: weights as split (hi/lo bf16) planes with EVERY BatchNorm folded in:
        post-conv BatchNorms as output scale + bias, pre-conv BatchNorms through ``ops.fold_input_bn_3x3`` (input scale
        into the weights, input shift into the border-dependent ``bias9``), the head's BatchNorm2d/BatchNorm1d into the FC.
        Each unit then writes ONE split tensor (its raw output, which is both the next conv's input and the shortcut)."""
        key = self._state_key()
        if self._packed_b3 is not None and key == self._packed_b3_key:
            return self._packed_b3
        if self.input_layer[0].weight.device.type != "cuda":
            raise RuntimeError("IR50 runs on the HIP kernels only: move the module to a GPU (no CPU fallback)")
        P = {}
        s, b = self._bn_affine(self.input_layer[1])
        P["stem_w"] = ops.pack_conv_weight(self.input_layer[0].weight.detach().contiguous(), s)
        P["stem_b"], P["stem_a"] = b, self.input_layer[2].weight.detach().contiguous()
        units = []
        for u in self.body:
            d = {"stride": u.stride, "proj": u.cin != u.depth}
            in_s, in_b = self._bn_affine(u.res_layer[0])
            w1, d["b9"] = ops.fold_input_bn_3x3(u.res_layer[1].weight.detach(), in_s, in_b)
            d["w1"] = ops.split_bf16(w1)
            d["a1"] = u.res_layer[2].weight.detach().contiguous()
            s2, b2 = self._bn_affine(u.res_layer[4])
            d["w2"] = ops.split_bf16(ops.pack_conv_weight(u.res_layer[3].weight.detach().contiguous(), s2))
            if u.stride == 2 and u.depth % 64 == 0:   # the same columns in the space-to-depth kernel's step order
                d["w2_s2d"] = ops.pack_s2d_weight(d["w2"], u.depth)
            d["b2"] = b2
            if d["proj"]:
                ss, sb = self._bn_affine(u.shortcut_layer[1])
                d["ws"] = ops.split_bf16(ops.pack_conv_weight(u.shortcut_layer[0].weight.detach().contiguous(), ss))
                d["bs"] = sb
            units.append(d)
        P["units"] = units
        hw = self.head_hw
        s0, t0 = self._bn_affine(self.output_layer[0])       # BatchNorm2d(512) in front of the flatten
        s4, t4 = self._bn_affine(self.output_layer[4])       # BatchNorm1d(512) behind the FC
        fc = self.output_layer[3]
        w = fc.weight.detach().view(fc.out_features, -1, hw * hw).permute(0, 2, 1)  # [o][(h,w)][c]: K order (c,h,w) -> (h,w,c)
        bias = fc.bias.detach() + (w * t0.view(1, 1, -1)).sum((1, 2))                 # W . t0
        w = (w * s0.view(1, 1, -1)).contiguous().view(fc.out_features, -1)            # W . diag(s0)
        P["head_w"] = ops.split_bf16((w * s4.view(-1, 1)).contiguous())
        P["head_b"] = (bias * s4 + t4).contiguous()
        self._packed_b3, self._packed_b3_key = P, key
        return P

    def pack_train_b3(self):
        key = tuple((p.data_ptr(), p._version) for p in self.parameters())
        if self._packed_train_b3 is not None and key == self._packed_train_b3_key:
            return self._packed_train_b3
        if self.input_layer[0].weight.device.type != "cuda":
            raise RuntimeError("IR50 runs on the HIP kernels only: move the module to a GPU (no CPU fallback)")
        P = {"stem_w": ops.pack_conv_weight(self.input_layer[0].weight.detach().contiguous()), "units": []}
        for u in self.body:
            d = {"w1_f32": ops.pack_conv_weight(u.res_layer[1].weight.detach().contiguous()),  # folded per step (batch statistics)
                 "w2": ops.split_bf16(ops.pack_conv_weight(u.res_layer[3].weight.detach().contiguous()))}
            if u.stride == 2 and u.depth % 64 == 0:
                d["w2_s2d"] = ops.pack_s2d_weight(d["w2"], u.depth)
            if u.cin != u.depth:
                d["ws"] = ops.split_bf16(ops.pack_conv_weight(u.shortcut_layer[0].weight.detach().contiguous()))
            P["units"].append(d)
        fc, hw = self.output_layer[3], self.head_hw
        P["head_w"] = ops.split_bf16(fc.weight.detach().view(fc.out_features, -1, hw * hw).permute(0, 2, 1).contiguous().view(
            fc.out_features, -1))
        self._packed_train_b3, self._packed_train_b3_key = P, key
        return P

    # ------------------------------------------------------------------ narrow (bf16 / fp16 storage) packing and forward
    NARROW = {"bf16": torch.bfloat16, "fp16": torch.float16}

    def pack_n16(self, dtype):
        """Eval-mode layouts for the narrow kernels: the same folds as ``pack_b3`` (post-conv BatchNorms as output scale +
        bias, pre-conv BatchNorms through ``fold_input_bn_3x3``, the head's two BatchNorms into the FC), computed in fp32
        and rounded ONCE to the storage type."""
        key = (dtype, self._state_key())
        if self._packed_n16 is not None and key == self._packed_n16_key:
            return self._packed_n16
        if self.input_layer[0].weight.device.type != "cuda":
            raise RuntimeError("IR50 runs on the HIP kernels only: move the module to a GPU (no CPU fallback)")
        P = {}
        s, b = self._bn_affine(self.input_layer[1])
        P["stem_w"] = ops.pack_conv_weight(self.input_layer[0].weight.detach().contiguous(), s)
        P["stem_b"], P["stem_a"] = b, self.input_layer[2].weight.detach().contiguous()
        units = []
        for u in self.body:
            d = {"stride": u.stride, "proj": u.cin != u.depth}
            in_s, in_b = self._bn_affine(u.res_layer[0])
            w1, d["b9"] = ops.fold_input_bn_3x3(u.res_layer[1].weight.detach(), in_s, in_b)
            d["w1"] = ops.to_n16(w1, dtype)
            d["a1"] = u.res_layer[2].weight.detach().contiguous()
            s2, b2 = self._bn_affine(u.res_layer[4])
            d["w2"] = ops.to_n16(ops.pack_conv_weight(u.res_layer[3].weight.detach().contiguous(), s2), dtype)
            if u.stride == 2 and u.depth % 128 == 0:   # the same columns in the space-to-depth kernel's step order
                d["w2_s2d"] = ops.pack_s2d_weight(d["w2"], u.depth)
            d["b2"] = b2
            if d["proj"]:
                ss, sb = self._bn_affine(u.shortcut_layer[1])
                d["ws"] = ops.to_n16(ops.pack_conv_weight(u.shortcut_layer[0].weight.detach().contiguous(), ss), dtype)
                d["bs"] = sb
            units.append(d)
        P["units"] = units
        hw = self.head_hw
        s0, t0 = self._bn_affine(self.output_layer[0])
        s4, t4 = self._bn_affine(self.output_layer[4])
        fc = self.output_layer[3]
        w = fc.weight.detach().view(fc.out_features, -1, hw * hw).permute(0, 2, 1)  # K order (c,h,w) -> (h,w,c)
        bias = fc.bias.detach() + (w * t0.view(1, 1, -1)).sum((1, 2))
        w = (w * s0.view(1, 1, -1)).contiguous().view(fc.out_features, -1)
        P["head_w"] = ops.to_n16((w * s4.view(-1, 1)).contiguous(), dtype)
        P["head_b"] = (bias * s4 + t4).contiguous()
        self._packed_n16, self._packed_n16_key = P, key
        return P

    def pack_train_n16(self, dtype):
        key = (dtype, tuple((p.data_ptr(), p._version) for p in self.parameters()))
        if self._packed_train_n16 is not None and key == self._packed_train_n16_key:
            return self._packed_train_n16
        if self.input_layer[0].weight.device.type != "cuda":
            raise RuntimeError("IR50 runs on the HIP kernels only: move the module to a GPU (no CPU fallback)")
        P = {"stem_w": ops.pack_conv_weight(self.input_layer[0].weight.detach().contiguous()), "units": []}
        for u in self.body:
            d = {"w1_f32": ops.pack_conv_weight(u.res_layer[1].weight.detach().contiguous()),
                 "w2": ops.to_n16(ops.pack_conv_weight(u.res_layer[3].weight.detach().contiguous()), dtype)}
            if u.stride == 2 and u.depth % 128 == 0:
                d["w2_s2d"] = ops.pack_s2d_weight(d["w2"], u.depth)
            if u.cin != u.depth:
                d["ws"] = ops.to_n16(ops.pack_conv_weight(u.shortcut_layer[0].weight.detach().contiguous()), dtype)
            P["units"].append(d)
        fc, hw = self.output_layer[3], self.head_hw
        P["head_w"] = ops.to_n16(fc.weight.detach().view(fc.out_features, -1, hw * hw).permute(0, 2, 1).contiguous().view(
            fc.out_features, -1), dtype)
        self._packed_train_n16, self._packed_train_n16_key = P, key
        return P

    def _forward_n16(self, x, dtype):
        """Eval / frozen forward on the narrow kernels (Cin = 3 stem on the fp32 small-Cin kernel, narrow output)."""
        P = self.pack_n16(dtype)
        xs = ops.stem_conv(x.contiguous(), P["stem_w"], None, P["stem_b"], P["stem_a"], out=dtype)["n16"]
        for d in P["units"]:
            s = d["stride"]
            s2d = "w2_s2d" in d and self._s2d_pair_ok(tuple(xs.shape), d["w2"].shape[0], narrow=True)
            t = ops.conv2d_n16(xs, d["w1"], 3, 3, pad=(1, 1), bias9=d["b9"], alpha=d["a1"], act1=ops.ACT_PRELU, y_s2d=s2d)["n16"]
            w2 = d["w2_s2d"] if s2d else d["w2"]
            if d["proj"]:
                sc = ops.conv2d_n16(xs, d["ws"], 1, 1, stride=s, bias=d["bs"])["n16"]
                xs = ops.conv2d_n16(t, w2, 3, 3, stride=s, pad=(1, 1), bias=d["b2"], residual=sc, res_stride=1, x_s2d=s2d)["n16"]
            else:
                xs = ops.conv2d_n16(t, w2, 3, 3, stride=s, pad=(1, 1), bias=d["b2"], residual=xs, res_stride=s, x_s2d=s2d)["n16"]
        n, h, w, c = xs.shape
        if h != self.head_hw or w != self.head_hw:
            raise RuntimeError(f"IR50 head was built for {self.head_hw}x{self.head_hw} feature maps "
                               f"({8 * self.head_hw}x{8 * self.head_hw} frames) but got {h}x{w}")
        k = h * w * c
        e = ops.conv2d_n16(xs.view(n, 1, 1, k), P["head_w"], 1, 1, bias=P["head_b"], split_k=self._head_split_k(n, k),
                           out_f32=True, out_n16=False)["y"]
        return ops.l2norm_rows(e.view(n, -1))

    def _forward_batch_stats_n16(self, x, dtype, head_mask=None):
        """Reference train() semantics (batch-statistics BatchNorm everywhere) on the narrow kernels.  Every activation
        tensor -- including the raw conv results the BatchNorms normalise -- is stored as one 16-bit plane, exactly what
        the reference's autocast does with its conv / batch_norm outputs; sums, statistics (taken from the fp32
        accumulators, before the rounding), normalisation and the residual adds are fp32.  Structure as the bf16x3 path:
        ONE bandwidth-bound ``bn_apply`` pass per unit (6 bytes per element instead of 12), the next unit's pre-conv
        BatchNorm folded into its 3x3 conv."""
        P = self.pack_train_n16(dtype)
        self._packed = self._packed_b3 = self._packed_n16 = None  # running statistics are about to change
        n = x.shape[0]
        plan = self._release_plan()
        first_released = len(P["units"]) if plan is None else plan
        if plan is not None:
            self._resolve_activation_memory(x.shape[0], x.shape[2])
        y = ys = xst = None
        if plan is not None and self._stem_released():
            y = self._released_stem(x)
        else:
            # the input layer is write-bound: statistics pass, then the conv again with BatchNorm + PReLU applied (stem_conv.hip)
            xc = x.contiguous()
            s, t = self._finalize(ops.stem_conv(xc, P["stem_w"]), n * x.shape[2] * x.shape[3], self.input_layer[1])
            r = ops.stem_conv(xc, P["stem_w"], s, t, self.input_layer[2].weight.detach(),
                              out="f32" if first_released == 0 else dtype, want_stats=True)
            ys, xst, y = r.get("n16"), r["stats"], r.get("y")
            del r
        for i, (u, d) in enumerate(zip(self.body, P["units"])):
            if i >= first_released:  # released for training: fp32 tensors, convs in this mode's arithmetic, with a backward
                y = self._released_unit(u, y, "fp16" if dtype == torch.float16 else "bf16")
                continue
            last = i + 1 == first_released  # the next consumer (released unit or head) wants fp32
            s1, t1 = self._finalize(xst, ys.numel() // u.cin, u.res_layer[0])
            w1, b9 = ops.fold_bn_3x3_packed(d["w1_f32"], s1, t1, dtype)
            s2d = "w2_s2d" in d and self._s2d_pair_ok(tuple(ys.shape), u.depth, narrow=True)
            tt = ops.conv2d_n16(ys, w1, 3, 3, pad=(1, 1), alpha=u.res_layer[2].weight.detach(),
                                act1=ops.ACT_PRELU, bias9=b9, y_s2d=s2d)["n16"]
            r = ops.conv2d_n16(tt, d["w2_s2d"] if s2d else d["w2"], 3, 3, stride=u.stride, pad=(1, 1), want_stats=True, x_s2d=s2d)
            del tt
            z = r["n16"]
            cnt = z.numel() // u.depth
            s2, t2 = self._finalize(r["stats"], cnt, u.res_layer[4])
            if u.cin != u.depth:
                rs = ops.conv2d_n16(ys, d["ws"], 1, 1, stride=u.stride, want_stats=True)
                ss, stt = self._finalize(rs["stats"], cnt, u.shortcut_layer[1])
                o = ops.bn_apply_nhwc_n16(z, s2, t2, res=rs["n16"], res_scale=ss, res_shift=stt, want_stats=True,
                                          out_f32=last, out_n16=not last)
            else:
                o = ops.bn_apply_nhwc_n16(z, s2, t2, res=ys, res_stride=u.stride, want_stats=True, out_f32=last,
                                          out_n16=not last)
            ys, xst, y = o.get("n16"), o["stats"], o.get("y")
            del z, r, o
        nn_, h, w, c = y.shape
        if h != self.head_hw or w != self.head_hw:
            raise RuntimeError(f"IR50 head was built for {self.head_hw}x{self.head_hw} feature maps but got {h}x{w}")
        p_drop = self.output_layer[1].p
        if head_mask is None and p_drop > 0:
            self._dropout_calls += 1
            head_mask = ops.dropout_mask(tuple(y.shape), p_drop, 0x1f50 + self.dropout_seed, self._dropout_calls * y.numel(),
                                         y.device)
        if plan is not None:
            return self._released_head(y, head_mask)
        s0, t0 = self._finalize(xst, y.numel() // c, self.output_layer[0])
        hfeat = ops.bn_apply_nhwc_n16(y, s0, t0, dtype=dtype, mask=head_mask)["n16"]
        k = h * w * c
        fc, bn1 = self.output_layer[3], self.output_layer[4]
        e = ops.conv2d_n16(hfeat.view(n, 1, 1, k), P["head_w"], 1, 1, bias=fc.bias.detach(),
                           split_k=self._head_split_k(n, k), out_f32=True, out_n16=False)["y"].view(n, -1)
        e, _, _ = ops.bn_rows_fwd(e, bn1.weight.detach(), bn1.bias.detach(), bn1.running_mean, bn1.running_var, True,
                                  bn1.eps, bn1.momentum)
        torch._foreach_add_([m.num_batches_tracked for m in self.modules()
                             if isinstance(m, (nn.BatchNorm2d, nn.BatchNorm1d))], 1)
        return ops.l2norm_rows(e)

    _S2D_OK = {}

    @classmethod
    def _s2d_pair_ok(cls, xshape, depth, narrow=False):
        """Can the two 3x3 convs of a stride-2 unit hand their intermediate over space-to-depth?  The first conv (xshape ->
        depth, stride 1) has to run on a window / patch kernel (their epilogues can permute the stores) and the second one
        needs even H and W, depth % 64 == 0 (narrow storage: % 128) and Wo <= 126 (csrc/conv_b3_s2d.hip, conv_n16_s2d.hip);
        otherwise the flat stride-2 kernel runs."""
        key = (xshape, depth, narrow)
        if key not in cls._S2D_OK:
            n, h, w, cin = xshape
            ok = h % 2 == 0 and w % 2 == 0 and depth % (128 if narrow else 64) == 0 and w // 2 <= 126 and cin % (64 if narrow else 32) == 0
            if ok and narrow:
                # measured (tools/bench_s2d.py --n16): the flat 256x256 tile already runs the >= 256-cout stride-2 layers at
                # 0.40-0.45 of the ceiling and the space-to-depth twin is within +-5 % of it; only the 128-cout layer (flat:
                # the 128x128 tile) gains (x1.10 @224x224, x1.26 @80x80)
                ok = depth == 128 and ops.conv2d_n16_tile(n, h, w, cin, depth, 3, 3, 1, (1, 1)) in ops.S2D_PRODUCER_TILES_N16
            elif ok:
                ok = ops.conv2d_b3_tile(n, h, w, cin, depth, 3, 3, 1, (1, 1)) in ops.S2D_PRODUCER_TILES
            cls._S2D_OK[key] = ok
        return cls._S2D_OK[key]

    def _forward_b3(self, x):
        """Eval / frozen forward on the bf16x3 kernels (Cin = 3 stem on the fp32 small-Cin kernel)."""
        P = self.pack_b3()
        U = P["units"]
        xs = ops.stem_conv(x.contiguous(), P["stem_w"], None, P["stem_b"], P["stem_a"], out="split")["split"]
        for d in U:
            s = d["stride"]
            s2d = "w2_s2d" in d and self._s2d_pair_ok(tuple(xs.shape), d["w2"].shape[0])
            t = ops.conv2d_b3(xs, d["w1"], 3, 3, pad=(1, 1), bias9=d["b9"], alpha=d["a1"], act1=ops.ACT_PRELU, y_s2d=s2d)["split"]
            w2 = d["w2_s2d"] if s2d else d["w2"]
            if d["proj"]:
                sc = ops.conv2d_b3(xs, d["ws"], 1, 1, stride=s, bias=d["bs"])["split"]
                xs = ops.conv2d_b3(t, w2, 3, 3, stride=s, pad=(1, 1), bias=d["b2"], residual=sc, res_stride=1, x_s2d=s2d)["split"]
            else:
                xs = ops.conv2d_b3(t, w2, 3, 3, stride=s, pad=(1, 1), bias=d["b2"], residual=xs, res_stride=s, x_s2d=s2d)["split"]
        n, h, w, c = xs.shape
        if h != self.head_hw or w != self.head_hw:
            raise RuntimeError(f"IR50 head was built for {self.head_hw}x{self.head_hw} feature maps "
                               f"({8 * self.head_hw}x{8 * self.head_hw} frames) but got {h}x{w}")
        k = h * w * c
        e = ops.conv2d_b3(xs.view(n, 1, 1, k), P["head_w"], 1, 1, bias=P["head_b"], split_k=self._head_split_k(n, k),
                          out_f32=True, out_split=False)["y"]
        return ops.l2norm_rows(e.view(n, -1))

    def _forward_batch_stats_b3(self, x, head_mask=None):
        """Reference train() semantics (batch-statistics BatchNorm everywhere) with the convolutions on the
        bf16x3 kernels.  Statistics and normalisation stay fp32.  Per unit there is ONE bandwidth-bound pass
        (``bn_apply``: post-conv BatchNorm + shortcut + the statistics of the sum), which stores the unit output as
        a split tensor; the NEXT unit's pre-conv BatchNorm -- whose scale/shift only exist once that pass has reduced
        the whole batch -- is folded into its 3x3 conv (input scale into the weights, input shift into a
        border-dependent bias, ``ops.fold_input_bn_3x3``), so no re-split pass over the activations is needed."""
        P = self.pack_train_b3()
        self._packed = self._packed_b3 = self._packed_n16 = None  # running statistics are about to change
        n = x.shape[0]
        plan = self._release_plan()
        first_released = len(P["units"]) if plan is None else plan
        if plan is not None:
            self._resolve_activation_memory(x.shape[0], x.shape[2])
        y = ys = xst = None
        if plan is not None and self._stem_released():
            y = self._released_stem(x)
        else:
            # the input layer is write-bound: statistics pass, then the conv again with BatchNorm + PReLU applied (stem_conv.hip)
            xc = x.contiguous()
            s, t = self._finalize(ops.stem_conv(xc, P["stem_w"]), n * x.shape[2] * x.shape[3], self.input_layer[1])
            r = ops.stem_conv(xc, P["stem_w"], s, t, self.input_layer[2].weight.detach(),
                              out="f32" if first_released == 0 else "split", want_stats=True)
            ys, xst, y = r.get("split"), r["stats"], r.get("y")
            del r
        for i, (u, d) in enumerate(zip(self.body, P["units"])):
            if i >= first_released:  # released for training: fp32 tensors, bf16x3 forward / data-gradient convs, with a backward
                y = self._released_unit(u, y, "bf16x3")
                continue
            last = i + 1 == first_released  # the next consumer (released unit or head) wants fp32
            s1, t1 = self._finalize(xst, ys.hi.numel() // u.cin, u.res_layer[0])
            w1, b9 = ops.fold_bn_3x3_packed(d["w1_f32"], s1, t1, "split")
            s2d = "w2_s2d" in d and self._s2d_pair_ok(tuple(ys.shape), u.depth)
            tt = ops.conv2d_b3(ys, w1, 3, 3, pad=(1, 1), alpha=u.res_layer[2].weight.detach(),
                               act1=ops.ACT_PRELU, bias9=b9, y_s2d=s2d)["split"]
            r = ops.conv2d_b3(tt, d["w2_s2d"] if s2d else d["w2"], 3, 3, stride=u.stride, pad=(1, 1), out_f32=True,
                              out_split=False, want_stats=True, x_s2d=s2d)
            del tt
            z = r["y"]
            cnt = z.numel() // u.depth
            s2, t2 = self._finalize(r["stats"], cnt, u.res_layer[4])
            if u.cin != u.depth:
                rs = ops.conv2d_b3(ys, d["ws"], 1, 1, stride=u.stride, out_f32=True, out_split=False, want_stats=True)
                ss, stt = self._finalize(rs["stats"], cnt, u.shortcut_layer[1])
                o = ops.bn_apply_nhwc_b3(z, s2, t2, res=rs["y"], res_scale=ss, res_shift=stt, want_stats=True,
                                         out_f32=last, out_split=not last)
            else:
                o = ops.bn_apply_nhwc_b3(z, s2, t2, res=ys, res_stride=u.stride, want_stats=True, out_f32=last,
                                         out_split=not last)
            ys, xst, y = o.get("split"), o["stats"], o.get("y")
            del z, r, o
        nn_, h, w, c = y.shape
        if h != self.head_hw or w != self.head_hw:
            raise RuntimeError(f"IR50 head was built for {self.head_hw}x{self.head_hw} feature maps but got {h}x{w}")
        p_drop = self.output_layer[1].p
        if head_mask is None and p_drop > 0:
            self._dropout_calls += 1
            head_mask = ops.dropout_mask(tuple(y.shape), p_drop, 0x1f50 + self.dropout_seed, self._dropout_calls * y.numel(),
                                         y.device)
        if plan is not None:
            return self._released_head(y, head_mask)
        s0, t0 = self._finalize(xst, y.numel() // c, self.output_layer[0])
        hfeat = ops.bn_apply_nhwc_b3(y, s0, t0, mask=head_mask)["split"]
        k = h * w * c
        fc, bn1 = self.output_layer[3], self.output_layer[4]
        e = ops.conv2d_b3(hfeat.view(n, 1, 1, k), P["head_w"], 1, 1, bias=fc.bias.detach(),
                          split_k=self._head_split_k(n, k), out_f32=True, out_split=False)["y"].view(n, -1)
        e, _, _ = ops.bn_rows_fwd(e, bn1.weight.detach(), bn1.bias.detach(), bn1.running_mean, bn1.running_var, True,
                                  bn1.eps, bn1.momentum)
        torch._foreach_add_([m.num_batches_tracked for m in self.modules()
                             if isinstance(m, (nn.BatchNorm2d, nn.BatchNorm1d))], 1)
        return ops.l2norm_rows(e)

    # ------------------------------------------------------------------ gradual release (first group: the head)
    def _release_plan(self):
        """base/parameter_control.py:85-96 flips ``requires_grad`` of parameter groups of the visual encoder: group 1
        (indices 4..9) is exactly ``output_layer``; groups 2 and 3 are stage 4 and the second half of stage 3, i.e. always
        a SUFFIX of the body.  Returns None (nothing released / no autograd) or the index of the first released body
        unit (len(body) when only the head is released; 0 = the whole body, the extension BASELINE configs[1] asks for, with
        or without the input layer).  Anything else fails loudly."""
        if not torch.is_grad_enabled():
            return None
        head = [p.requires_grad for p in self.output_layer.parameters()]
        stem = [p.requires_grad for p in self.input_layer.parameters()]
        if any(stem) and not all(stem):
            raise NotImplementedError("release the whole input layer (conv, BatchNorm, PReLU) or nothing")
        flags = []
        for u in self.body:
            f = [p.requires_grad for p in u.parameters()]
            if any(f) and not all(f):
                raise NotImplementedError("release whole units (all parameters of a bottleneck_IR unit) or nothing")
            flags.append(all(f))
        if not any(head) and not any(flags) and not any(stem):
            return None
        if not all(head):
            raise NotImplementedError("release the whole output layer (parameters 4..9) first, as the reference does")
        first = len(flags)
        while first > 0 and flags[first - 1]:
            first -= 1
        if any(flags[:first]):
            raise NotImplementedError("released body units must form a suffix of the body (the reference releases from the top)")
        if all(stem) and first != 0:
            raise NotImplementedError("the input layer can only be released together with the whole body (gradients flow top-down)")
        return first

    def _stem_released(self):
        return torch.is_grad_enabled() and all(p.requires_grad for p in self.input_layer.parameters())

    def _released_stem(self, x):
        il = self.input_layer
        return _ReleasedStem.apply(x, il[1], il[0].weight, il[1].weight, il[1].bias, il[2].weight)

    def _resolve_activation_memory(self, frames, hw):
        """What the released units of THIS forward keep (see ``activation_memory``); called once per forward."""
        mode = self.activation_memory
        if mode not in ("auto", "raw", "recompute", "recompute16"):
            raise ValueError(f"unknown activation_memory {mode!r}")
        if mode == "auto":
            released = sum(1 for u in self.body if all(p.requires_grad for p in u.parameters()))
            raw = 410e6 * (hw / 224.0) ** 2 * frames * released / len(self.body)      # fp32 raw tensors of the released units
            free, _ = torch.cuda.mem_get_info()
            free += torch.cuda.memory_reserved() - torch.cuda.memory_allocated()      # the allocator's cached blocks are reusable
            mode = "raw" if 1.6 * raw < free else "recompute"
        self._act_mem = mode
        return mode

    def _released_unit(self, u, y, prec="fp32"):
        pr = u.res_layer
        sc = u.shortcut_layer if u.cin != u.depth else None
        return _ReleasedUnit.apply(y, u, prec, self._act_mem, pr[0].weight, pr[0].bias, pr[1].weight, pr[2].weight, pr[3].weight, pr[4].weight,
                                   pr[4].bias, sc[0].weight if sc is not None else None,
                                   sc[1].weight if sc is not None else None, sc[1].bias if sc is not None else None)

    def _released_head(self, y, head_mask):
        bn2, fc, bn1 = self.output_layer[0], self.output_layer[3], self.output_layer[4]
        out = _ReleasedHead.apply(y, head_mask, bn2, fc, bn1, bn2.weight, bn2.bias, fc.weight, fc.bias, bn1.weight, bn1.bias)
        torch._foreach_add_([m.num_batches_tracked for m in self.modules()
                             if isinstance(m, (nn.BatchNorm2d, nn.BatchNorm1d))], 1)
        return out

    # ------------------------------------------------------------------ forward
    def _head_split_k(self, n, k):
        tiles = ((n + 127) // 128) * 4
        return max(1, min(k // 32, (768 + tiles - 1) // tiles))

    def _finalize(self, stats, count, bn):
        return ops.bn_finalize(stats, count, bn.weight.detach(), bn.bias.detach(), bn.running_mean, bn.running_var,
                               momentum=bn.momentum, eps=bn.eps)

    def _forward_batch_stats(self, x, head_mask=None):
        """Reference train() semantics: every BatchNorm uses batch statistics over the N frames and
        updates its running buffers; Dropout(0.4) before the head FC.  ``head_mask`` ([N,h,w,512],
        pre-scaled) overrides the generated dropout mask (parity tests)."""
        P = self.pack_train()
        self._packed = self._packed_b3 = self._packed_n16 = None  # running statistics are about to change: folded eval weights go stale
        n = x.shape[0]
        plan = self._release_plan()
        first_released = len(P["units"]) if plan is None else plan
        if plan is not None:
            self._resolve_activation_memory(x.shape[0], x.shape[2])
        xst = None
        if plan is not None and self._stem_released():
            y = self._released_stem(x)
        else:
            xc = x.contiguous()
            s, t = self._finalize(ops.stem_conv(xc, P["stem_w"]), x.shape[0] * x.shape[2] * x.shape[3], self.input_layer[1])
            r = ops.stem_conv(xc, P["stem_w"], s, t, self.input_layer[2].weight.detach(), out="f32", want_stats=True)
            y, xst = r["y"], r["stats"]
        for i, (u, d) in enumerate(zip(self.body, P["units"])):
            if i >= first_released:
                y = self._released_unit(u, y)
                continue
            s1, t1 = self._finalize(xst, y.numel() // u.cin, u.res_layer[0])
            tt = ops.conv2d(y, d["w1"], 3, 3, pad=(1, 1), in_scale=s1, in_shift=t1,
                            alpha=u.res_layer[2].weight.detach(), act1=ops.ACT_PRELU)
            z, zst = ops.conv2d(tt, d["w2"], 3, 3, stride=u.stride, pad=(1, 1), want_stats=True)
            del tt
            cnt = z.numel() // u.depth
            s2, t2 = self._finalize(zst, cnt, u.res_layer[4])
            if u.cin != u.depth:
                sz, sst = ops.conv2d(y, d["ws"], 1, 1, stride=u.stride, want_stats=True)
                ss, stt = self._finalize(sst, cnt, u.shortcut_layer[1])
                y, xst = ops.bn_apply_nhwc(z, s2, t2, res=sz, res_scale=ss, res_shift=stt, want_stats=True)
            else:
                y, xst = ops.bn_apply_nhwc(z, s2, t2, res=y, res_stride=u.stride, want_stats=True)
            del z
        nn_, h, w, c = y.shape
        if h != self.head_hw or w != self.head_hw:
            raise RuntimeError(f"IR50 head was built for {self.head_hw}x{self.head_hw} feature maps but got {h}x{w}")
        p_drop = self.output_layer[1].p
        if head_mask is None and p_drop > 0:
            self._dropout_calls += 1
            head_mask = ops.dropout_mask(tuple(y.shape), p_drop, 0x1f50 + self.dropout_seed, self._dropout_calls * y.numel(),
                                         y.device)
        if plan is not None:
            return self._released_head(y, head_mask)
        s0, t0 = self._finalize(xst, y.numel() // c, self.output_layer[0])
        hfeat = ops.bn_apply_nhwc(y, s0, t0, mask=head_mask)
        k = h * w * c
        fc, bn1 = self.output_layer[3], self.output_layer[4]
        e = ops.linear(hfeat.view(n, k), P["head_w"], bias=fc.bias.detach(), split_k=self._head_split_k(n, k))
        e, _, _ = ops.bn_rows_fwd(e, bn1.weight.detach(), bn1.bias.detach(), bn1.running_mean, bn1.running_var, True,
                                  bn1.eps, bn1.momentum)
        torch._foreach_add_([m.num_batches_tracked for m in self.modules()
                             if isinstance(m, (nn.BatchNorm2d, nn.BatchNorm1d))], 1)
        return ops.l2norm_rows(e)

    def forward(self, x, head_mask=None):
        """x: [N,3,H,W] float32 on the GPU -> [N,512], rows of unit L2 norm.

        eval(): running statistics, every BatchNorm folded into the conv kernels (fast path).
        train(): ``bn_mode == "reference"`` reproduces the reference, whose model.train() also puts
        this frozen encoder's BatchNorm/Dropout layers in train mode;
        ``bn_mode == "frozen"`` keeps the encoder in eval behaviour (common practice, faster)."""
        if self.precision not in ("bf16x3", "fp32", "bf16", "fp16"):
            raise ValueError(f"unknown precision {self.precision!r}")
        narrow = self.NARROW.get(self.precision)
        if self.follow_autocast and torch.is_autocast_enabled():
            narrow = {torch.float16: torch.float16, torch.bfloat16: torch.bfloat16}.get(torch.get_autocast_dtype("cuda"), narrow)
        with torch.autocast("cuda", enabled=False):   # the kernels take fp32 / explicit narrow tensors: no implicit casts inside
            return self._forward_impl(x, head_mask, narrow)

    def _forward_impl(self, x, head_mask, narrow):
        if self.training and self.bn_mode == "reference":
            if narrow is not None:
                return self._forward_batch_stats_n16(x, narrow, head_mask)
            if self.precision == "bf16x3":
                return self._forward_batch_stats_b3(x, head_mask)
            return self._forward_batch_stats(x, head_mask)
        if self._release_plan() is not None:
            raise NotImplementedError("released encoder parameters need model.train() with bn_mode = 'reference' (what the "
                                      "reference's gradual release runs in); use torch.no_grad() for evaluation")
        if narrow is not None:
            return self._forward_n16(x, narrow)
        if self.precision == "bf16x3":
            return self._forward_b3(x)
        P = self.pack()
        x = x.contiguous()
        y = ops.stem_conv(x, P["stem_w"], None, P["stem_b"], P["stem_a"], out="f32")["y"]
        for d in P["units"]:
            s = d["stride"]
            t = ops.conv2d(y, d["w1"], 3, 3, pad=(1, 1), in_scale=d["in_s"], in_shift=d["in_b"], alpha=d["a1"],
                           act1=ops.ACT_PRELU)
            if d["proj"]:
                sc = ops.conv2d(y, d["ws"], 1, 1, stride=s, bias=d["bs"])
                y = ops.conv2d(t, d["w2"], 3, 3, stride=s, pad=(1, 1), bias=d["b2"], residual=sc, res_stride=1)
            else:
                y = ops.conv2d(t, d["w2"], 3, 3, stride=s, pad=(1, 1), bias=d["b2"], residual=y, res_stride=s)
        n, h, w, c = y.shape
        if h != self.head_hw or w != self.head_hw:
            raise RuntimeError(f"IR50 head was built for {self.head_hw}x{self.head_hw} feature maps "
                               f"({8 * self.head_hw}x{8 * self.head_hw} frames) but got {h}x{w}")
        k = h * w * c
        e = ops.linear(y.view(n, k), P["head_w"], bias=P["head_b"], split_k=self._head_split_k(n, k))
        return ops.l2norm_rows(e)


class VisualBackbone(nn.Module):
    """Same constructor and keys as the reference (models/backbone.py:69-130).  ``head_hw``
    is the one extension: 5 for the reference's 40x40 crops, 28 for 224x224 frames."""

    def __init__(self, input_channels=3, num_classes=8, use_pretrained=True, state_dict_path="", mode="ir",
                 embedding_dim=512, head_hw=5):
        super().__init__()
        if mode != "ir":
            raise NotImplementedError("only the 'ir' units are on the hot path (the reference never builds ir_se)")
        self.backbone = IR50(input_channels=input_channels, drop_ratio=0.4, head_hw=head_hw,
                             embedding_dim=embedding_dim)
        self.logits = nn.Linear(embedding_dim, num_classes)  # unused by forward, kept for the state dict
        if use_pretrained:
            state_dict = torch.load(state_dict_path, map_location="cpu", weights_only=True)
            if "backbone" in list(state_dict.keys())[0]:
                state_dict = {k[9:]: v for k, v in state_dict.items() if "logits" not in k}
            self.backbone.load_state_dict(state_dict)
            for p in self.backbone.parameters():
                p.requires_grad = False
        # head re-initialisation the reference always performs (backbone.py:99-122)
        for m in self.backbone.output_layer.modules():
            if isinstance(m, nn.Linear):
                nn.init.xavier_uniform_(m.weight)
                nn.init.constant_(m.bias, 0)
            elif isinstance(m, (nn.BatchNorm2d, nn.BatchNorm1d)):
                m.weight.data.fill_(1)
                m.bias.data.zero_()
        nn.init.xavier_uniform_(self.logits.weight)
        nn.init.constant_(self.logits.bias, 0)

    def forward(self, x, head_mask=None):
        return self.backbone(x, head_mask)

    def extract(self, x):
        return self.backbone(x)
