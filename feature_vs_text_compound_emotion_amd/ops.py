"""Tensor-level wrappers over the C-ABI (device memory is borrowed from torch).

Everything here is a thin marshalling layer: shape checks, output allocation,
one C call.  No arithmetic happens in Python.
"""
import ctypes

import torch

from . import _lib
from ._lib import ACT_GELU, ACT_LEAKY, ACT_NONE, ACT_PRELU, ACT_RELU, ConvDesc, check, current_stream, ptr  # noqa: F401


def _dev_f32(t, name):
    if t is None:
        return
    if not (t.is_cuda and t.dtype == torch.float32 and t.is_contiguous()):
        raise ValueError(f"{name}: expected a contiguous float32 tensor on the GPU, got "
                         f"{t.dtype} {t.device} contiguous={t.is_contiguous()}")


def conv_kpad(kh, kw, cin):
    return _lib.load().cer_conv_kpad(kh, kw, cin)


def pack_conv_weight(w_oihw, out_scale=None, flip=False):
    """[Cout,Cin,KH,KW] -> [Cout,Kpad] with k = (kh*KW+kw)*Cin + c, BN scale folded."""
    _dev_f32(w_oihw, "w_oihw")
    _dev_f32(out_scale, "out_scale")
    cout, cin, kh, kw = w_oihw.shape
    out = torch.empty((cout, conv_kpad(kh, kw, cin)), device=w_oihw.device, dtype=torch.float32)
    check(_lib.load().cer_pack_conv_weight(ptr(w_oihw), ptr(out_scale), ptr(out), cout, cin, kh, kw,
                                           1 if flip else 0, current_stream()), "cer_pack_conv_weight")
    return out


def conv2d(x, w_packed, kh, kw, *, stride=1, dil=(1, 1), pad=(0, 0), out_hw=None, in_scale=None,
           in_shift=None, bias=None, alpha=None, residual=None, res_stride=1, mask=None, act1=ACT_NONE,
           act2=ACT_NONE, slope=0.01, split_k=1, x_nchw=False, tile=0, out=None):
    """y[N,Ho,Wo,Cout] = act2(mask*act1(conv(affine(x), w)+bias) + residual).  x is NHWC
    (or NCHW with ``x_nchw`` on the small-Cin path)."""
    lib = _lib.load()
    for t, n in ((x, "x"), (w_packed, "w"), (in_scale, "in_scale"), (in_shift, "in_shift"), (bias, "bias"),
                 (alpha, "alpha"), (residual, "residual"), (mask, "mask")):
        _dev_f32(t, n)
    if x_nchw:
        n, cin, h, w = x.shape
    else:
        n, h, w, cin = x.shape
    cout = w_packed.shape[0]
    if w_packed.shape[1] != conv_kpad(kh, kw, cin):
        raise ValueError(f"packed weight has K={w_packed.shape[1]}, expected {conv_kpad(kh, kw, cin)}")
    if out_hw is None:
        ho = (h + 2 * pad[0] - dil[0] * (kh - 1) - 1) // stride + 1
        wo = (w + 2 * pad[1] - dil[1] * (kw - 1) - 1) // stride + 1
    else:
        ho, wo = out_hw
    d = ConvDesc()
    d.N, d.H, d.W, d.Cin, d.Ho, d.Wo, d.Cout = n, h, w, cin, ho, wo, cout
    d.KH, d.KW, d.stride, d.dil_h, d.dil_w, d.pad_t, d.pad_l = kh, kw, stride, dil[0], dil[1], pad[0], pad[1]
    d.x_nchw = 1 if x_nchw else 0
    d.res_stride, d.Hr, d.Wr = res_stride, 0, 0
    if residual is not None:
        if residual.shape[0] != n or residual.shape[3] != cout:
            raise ValueError(f"residual shape {tuple(residual.shape)} does not match N={n}, Cout={cout}")
        d.Hr, d.Wr = residual.shape[1], residual.shape[2]
    for t, nme, length in ((in_scale, "in_scale", cin), (in_shift, "in_shift", cin), (bias, "bias", cout),
                           (alpha, "alpha", cout)):
        if t is not None and t.numel() != length:
            raise ValueError(f"{nme} has {t.numel()} elements, expected {length}")
    if mask is not None and mask.numel() != n * ho * wo * cout:
        raise ValueError("mask must have the output's shape")
    d.act1, d.act2, d.slope, d.split_k, d.tile = act1, act2, slope, split_k, tile
    if out is None:
        out = torch.empty((n, ho, wo, cout), device=x.device, dtype=torch.float32)
    else:
        _dev_f32(out, "out")
        if out.numel() != n * ho * wo * cout:
            raise ValueError("out has the wrong size")
    ws_bytes = lib.cer_conv2d_workspace_bytes(ctypes.byref(d))
    ws = torch.empty((ws_bytes // 4,), device=x.device, dtype=torch.float32) if ws_bytes else None
    check(lib.cer_conv2d_fwd(ctypes.byref(d), ptr(x), ptr(w_packed), ptr(in_scale), ptr(in_shift), ptr(bias),
                             ptr(alpha), ptr(residual), ptr(mask), ptr(out), ptr(ws), ws_bytes,
                             current_stream()), "cer_conv2d_fwd")
    return out


def linear(x2d, w_packed, bias=None, act=ACT_NONE, split_k=1, residual=None, out=None):
    """[M,K] @ W[Cout,K]^T as a 1x1 conv on an [M,1,1,K] image."""
    m, k = x2d.shape
    res = residual.view(m, 1, 1, -1) if residual is not None else None
    y = conv2d(x2d.view(m, 1, 1, k), w_packed, 1, 1, bias=bias, act1=act, split_k=split_k, residual=res,
               out=out)
    return y.view(m, -1)


def l2norm_rows(x):
    _dev_f32(x, "x")
    y = torch.empty_like(x)
    check(_lib.load().cer_l2norm_rows(ptr(x), ptr(y), x.shape[0], x.shape[1], current_stream()),
          "cer_l2norm_rows")
    return y


def maxpool2x2_nhwc(x):
    _dev_f32(x, "x")
    n, h, w, c = x.shape
    y = torch.empty((n, h // 2, w // 2, c), device=x.device, dtype=torch.float32)
    check(_lib.load().cer_maxpool2x2_nhwc(ptr(x), ptr(y), n, h, w, c, current_stream()), "cer_maxpool2x2_nhwc")
    return y
