"""Tensor-level wrappers over the C-ABI (device memory is borrowed from torch).

Everything here is a thin marshalling layer: shape checks, output allocation,
one C call.  No arithmetic happens in Python.
"""
import ctypes
import os

import torch

from . import _lib
from ._lib import (ACT_GELU, ACT_LEAKY, ACT_NONE, ACT_PRELU, ACT_RELU, STORE_BF16, STORE_F16, STORE_NONE, ConvDesc,  # noqa: F401
                   ConvIO, check, current_stream, ptr)

LEAKY_SLOPE = 0.01


def _dev_f32(t, name, contiguous=True):
    if t is None:
        return
    if not (t.is_cuda and t.dtype == torch.float32 and (t.is_contiguous() or not contiguous)):
        raise ValueError(f"{name}: expected a contiguous float32 tensor on the GPU, got "
                         f"{t.dtype} {t.device} contiguous={t.is_contiguous()}")


def _rows(t, name):
    """Accept a [R,C] tensor that is dense or a column slice of a wider row-major buffer.
    Returns (rows, cols, pitch)."""
    if not (t.is_cuda and t.dtype == torch.float32 and t.dim() == 2 and t.stride(1) == 1):
        raise ValueError(f"{name}: expected a 2-D float32 GPU tensor with unit column stride")
    return t.shape[0], t.shape[1], (t.stride(0) if t.shape[0] > 1 else max(t.stride(0), t.shape[1]))


def _empty(shape, like):
    return torch.empty(shape, device=like.device, dtype=torch.float32)


def conv_kpad(kh, kw, cin):
    return _lib.load().cer_conv_kpad(kh, kw, cin)


def pack_conv_weight(w_oihw, out_scale=None, flip=False, transpose=False):
    """[Cout,Cin,KH,KW] -> [Cout,Kpad] with k = (kh*KW+kw)*Cin + c, BN scale folded;
    ``transpose`` gives the data-gradient filter [Cin, Kpad(KH,KW,Cout)]."""
    _dev_f32(w_oihw, "w_oihw")
    _dev_f32(out_scale, "out_scale")
    cout, cin, kh, kw = w_oihw.shape
    rows, inner = (cin, cout) if transpose else (cout, cin)
    out = _empty((rows, conv_kpad(kh, kw, inner)), w_oihw)
    check(_lib.load().cer_pack_conv_weight(ptr(w_oihw), ptr(out_scale), ptr(out), cout, cin, kh, kw,
                                           1 if flip else 0, 1 if transpose else 0, current_stream()),
          "cer_pack_conv_weight")
    return out


def conv2d(x, w_packed, kh, kw, *, stride=1, dil=(1, 1), pad=(0, 0), out_hw=None, in_scale=None,
           in_shift=None, bias=None, alpha=None, residual=None, res_stride=1, mask=None, act1=ACT_NONE,
           act2=ACT_NONE, slope=LEAKY_SLOPE, split_k=1, x_nchw=False, tile=0, out=None, aux=None, x_ld=0, y_ld=0,
           x_shape=None, want_stats=False, out_split=False, next_affine=None, want_f32=True, out_n16=None):
    """y[N,Ho,Wo,Cout] = act2(mask*act1(conv(affine(x), w)+bias) + residual).  x is NHWC
    (or NCHW with ``x_nchw`` on the small-Cin path).  ``x_shape`` = (N,H,W,Cin) overrides
    x.shape when x is a column slice (then ``x_ld`` is its row pitch)."""
    lib = _lib.load()
    for t, n in ((w_packed, "w"), (in_scale, "in_scale"), (in_shift, "in_shift"), (bias, "bias"),
                 (alpha, "alpha"), (residual, "residual"), (mask, "mask"), (aux, "aux")):
        _dev_f32(t, n)
    _dev_f32(x, "x", contiguous=(x_ld == 0))
    if x_shape is not None:
        n, h, w, cin = x_shape
    elif x_nchw:
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
    d.x_ld, d.y_ld = x_ld, y_ld
    if residual is not None:
        if residual.shape[0] != n or residual.shape[3] != cout:
            raise ValueError(f"residual shape {tuple(residual.shape)} does not match N={n}, Cout={cout}")
        d.Hr, d.Wr = residual.shape[1], residual.shape[2]
    for t, nme, length in ((in_scale, "in_scale", cin), (in_shift, "in_shift", cin), (bias, "bias", cout),
                           (alpha, "alpha", cout)):
        if t is not None and t.numel() != length:
            raise ValueError(f"{nme} has {t.numel()} elements, expected {length}")
    for t, nme in ((mask, "mask"), (aux, "aux")):
        if t is not None and t.numel() != n * ho * wo * cout:
            raise ValueError(f"{nme} must have the output's shape")
    d.act1, d.act2, d.slope, d.split_k, d.tile = act1, act2, slope, split_k, tile
    if out is None and want_f32:
        if y_ld:
            raise ValueError("y_ld needs an explicit out buffer")
        out = _empty((n, ho, wo, cout), x)
    elif out is not None:
        _dev_f32(out, "out", contiguous=(y_ld == 0))
    ws_bytes = lib.cer_conv2d_workspace_bytes(ctypes.byref(d))
    ws = _empty((ws_bytes // 4,), x) if ws_bytes else None
    stats = _empty((lib.cer_conv2d_stats_tiles(ctypes.byref(d), 0), 2, cout), x) if want_stats else None
    if out_n16 is not None:  # fp32 kernel, narrow (single bf16 / half plane) output: the Cin = 3 stem of the narrow encoder
        if out_split or next_affine is not None:
            raise ValueError("out_n16 excludes the split outputs")
        io = ConvIO()
        io.x, io.w = x.data_ptr(), w_packed.data_ptr()
        for name, t in (("in_scale", in_scale), ("in_shift", in_shift), ("bias", bias), ("alpha", alpha),
                        ("residual", residual), ("mask", mask), ("y", out), ("aux", aux), ("stats", stats)):
            if t is not None:
                setattr(io, name, t.data_ptr())
        d.storage = storage_of(out_n16)
        res = {"y": out, "stats": stats, "n16": torch.empty((n, ho, wo, cout), device=x.device, dtype=out_n16)}
        io.y_hi = res["n16"].data_ptr()
        check(lib.cer_conv2d_run(ctypes.byref(d), ctypes.byref(io), ptr(ws), ws_bytes, current_stream()), "cer_conv2d_run")
        return res
    if not (out_split or next_affine is not None):
        check(lib.cer_conv2d_fwd(ctypes.byref(d), ptr(x), ptr(w_packed), ptr(in_scale), ptr(in_shift), ptr(bias),
                                 ptr(alpha), ptr(residual), ptr(mask), ptr(out), ptr(aux), ptr(stats), ptr(ws), ws_bytes,
                                 current_stream()), "cer_conv2d_fwd")
        return (out, stats) if want_stats else out
    # fp32 kernel with split (bf16 hi/lo) outputs: feeds the bf16x3 layers (e.g. the Cin = 3 stem)
    io = ConvIO()
    io.x, io.w = x.data_ptr(), w_packed.data_ptr()
    for name, t in (("in_scale", in_scale), ("in_shift", in_shift), ("bias", bias), ("alpha", alpha),
                    ("residual", residual), ("mask", mask), ("y", out), ("aux", aux), ("stats", stats)):
        if t is not None:
            setattr(io, name, t.data_ptr())
    res = {"y": out, "stats": stats}
    if out_split:
        res["split"] = Split.empty((n, ho, wo, cout), x.device)
        io.y_hi, io.y_lo = res["split"].hi.data_ptr(), res["split"].lo.data_ptr()
    if next_affine is not None:
        s2, t2 = next_affine
        res["next"] = Split.empty((n, ho, wo, cout), x.device)
        io.s2, io.t2 = s2.data_ptr(), t2.data_ptr()
        io.y2_hi, io.y2_lo = res["next"].hi.data_ptr(), res["next"].lo.data_ptr()
    check(lib.cer_conv2d_run(ctypes.byref(d), ctypes.byref(io), ptr(ws), ws_bytes, current_stream()), "cer_conv2d_run")
    return res


class Split:
    """A tensor carried as two bf16 planes: value = hi + lo (hi = bf16(v), lo = bf16(v - hi))."""
    __slots__ = ("hi", "lo")

    def __init__(self, hi, lo):
        self.hi, self.lo = hi, lo

    @property
    def shape(self):
        return self.hi.shape

    def float(self):
        return self.hi.float() + self.lo.float()

    def view(self, *shape):
        return Split(self.hi.view(*shape), self.lo.view(*shape))

    @staticmethod
    def empty(shape, device):
        return Split(torch.empty(shape, device=device, dtype=torch.bfloat16),
                     torch.empty(shape, device=device, dtype=torch.bfloat16))


def split_bf16(x, scale=None, shift=None):
    """fp32 tensor -> Split (round-to-nearest-even on both parts), optionally after the per-channel
    affine x*scale[c]+shift[c] over the last (channel) axis."""
    _dev_f32(x, "x")
    _dev_f32(scale, "scale")
    _dev_f32(shift, "shift")
    out = Split.empty(x.shape, x.device)
    c = x.shape[-1] if scale is not None else 0
    check(_lib.load().cer_split_bf16(ptr(x), ptr(scale), ptr(shift), c, ptr(out.hi), ptr(out.lo), x.numel(),
                                     current_stream()), "cer_split_bf16")
    return out


def _dev_bf16(t, name):
    if t is not None and not (t.is_cuda and t.dtype == torch.bfloat16 and t.is_contiguous()):
        raise ValueError(f"{name}: expected a contiguous bfloat16 GPU tensor")


# bench.py sets this to a list to time every bf16x3 conv launch with HIP events on the launch stream:
# (kernel variant id, algorithmic FLOPs, start event, end event, algorithmic bytes)
CONV_TRACE = None


def conv2d_b3(x, w, kh, kw, *, stride=1, dil=(1, 1), pad=(0, 0), out_hw=None, bias=None, alpha=None, residual=None,
              res_stride=1, act1=ACT_NONE, act2=ACT_NONE, slope=LEAKY_SLOPE, split_k=1, tile=0, out_f32=False,
              out_split=True, next_affine=None, want_stats=False, bias9=None, x_s2d=False, y_s2d=False):
    """bf16x3 convolution.  x, w: Split tensors (x NHWC [N,H,W,Cin], w [Cout,Kpad]); residual: Split
    or fp32 tensor.  Returns a dict with the requested outputs: 'y' (fp32), 'split' (Split), 'next'
    (Split of out*s2+t2 when ``next_affine=(s2, t2)``), 'stats'.
    ``y_s2d``: the Split output is stored space-to-depth, [N, Ho/2, Wo/2, 4*Cout] (``space_to_depth`` is the torch
    statement of the layout); ``x_s2d``: x is such a tensor (of the [N, 2*x.shape[1], 2*x.shape[2], x.shape[3]/4] input of
    this 3x3 / stride 2 / pad 1 conv) and w went through ``pack_s2d_weight``."""
    lib = _lib.load()
    for t, n in ((x.hi, "x.hi"), (x.lo, "x.lo"), (w.hi, "w.hi"), (w.lo, "w.lo")):
        _dev_bf16(t, n)
    _dev_f32(bias, "bias")
    _dev_f32(alpha, "alpha")
    n, h, wd, cin = x.shape
    if x_s2d:
        if cin % 4:
            raise ValueError("a space-to-depth input has 4 * Cin channels")
        h, wd, cin = 2 * h, 2 * wd, cin // 4
    cout = w.shape[0]
    if w.shape[1] != conv_kpad(kh, kw, cin):
        raise ValueError(f"packed weight has K={w.shape[1]}, expected {conv_kpad(kh, kw, cin)}")
    if out_hw is None:
        ho = (h + 2 * pad[0] - dil[0] * (kh - 1) - 1) // stride + 1
        wo = (wd + 2 * pad[1] - dil[1] * (kw - 1) - 1) // stride + 1
    else:
        ho, wo = out_hw
    d = ConvDesc()
    d.N, d.H, d.W, d.Cin, d.Ho, d.Wo, d.Cout = n, h, wd, cin, ho, wo, cout
    d.KH, d.KW, d.stride, d.dil_h, d.dil_w, d.pad_t, d.pad_l = kh, kw, stride, dil[0], dil[1], pad[0], pad[1]
    d.res_stride, d.Hr, d.Wr = res_stride, 0, 0
    d.act1, d.act2, d.slope, d.split_k, d.tile = act1, act2, slope, split_k, tile
    d.x_s2d, d.y_s2d = int(x_s2d), int(y_s2d)
    io = ConvIO()
    io.x_hi, io.x_lo, io.w_hi, io.w_lo = x.hi.data_ptr(), x.lo.data_ptr(), w.hi.data_ptr(), w.lo.data_ptr()
    io.bias = bias.data_ptr() if bias is not None else None
    io.alpha = alpha.data_ptr() if alpha is not None else None
    if bias9 is not None:
        _dev_f32(bias9, "bias9")
        if tuple(bias9.shape) != (9, cout):
            raise ValueError("bias9 must be [9, Cout]")
        io.bias9 = bias9.data_ptr()
    if residual is not None:
        rshape = residual.shape
        if rshape[0] != n or rshape[3] != cout:
            raise ValueError("residual shape does not match the output")
        d.Hr, d.Wr = rshape[1], rshape[2]
        if isinstance(residual, Split):
            _dev_bf16(residual.hi, "residual.hi")
            io.res_hi, io.res_lo = residual.hi.data_ptr(), residual.lo.data_ptr()
        else:
            _dev_f32(residual, "residual")
            io.residual = residual.data_ptr()
    res = {}
    dev = x.hi.device
    if out_f32:
        res["y"] = torch.empty((n, ho, wo, cout), device=dev, dtype=torch.float32)
        io.y = res["y"].data_ptr()
    if out_split:
        res["split"] = Split.empty((n, ho // 2, wo // 2, 4 * cout) if y_s2d else (n, ho, wo, cout), dev)
        io.y_hi, io.y_lo = res["split"].hi.data_ptr(), res["split"].lo.data_ptr()
    if next_affine is not None:
        s2, t2 = next_affine
        _dev_f32(s2, "s2")
        _dev_f32(t2, "t2")
        res["next"] = Split.empty((n, ho, wo, cout), dev)
        io.s2, io.t2 = s2.data_ptr(), t2.data_ptr()
        io.y2_hi, io.y2_lo = res["next"].hi.data_ptr(), res["next"].lo.data_ptr()
    if want_stats:
        res["stats"] = torch.empty((lib.cer_conv2d_stats_tiles(ctypes.byref(d), 1), 2, cout), device=dev,
                                   dtype=torch.float32)
        io.stats = res["stats"].data_ptr()
    ws_bytes = lib.cer_conv2d_workspace_bytes(ctypes.byref(d))
    ws = torch.empty((ws_bytes // 4,), device=dev, dtype=torch.float32) if ws_bytes else None
    if CONV_TRACE is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    check(lib.cer_conv2d_run(ctypes.byref(d), ctypes.byref(io), ptr(ws), ws_bytes, current_stream()), "cer_conv2d_run")
    if CONV_TRACE is not None:
        e1.record()
        # algorithmic bytes: split input (4 B/elt) + every output tensor written + split weights + residual read
        nout = n * ho * wo * cout
        nbytes = 4.0 * n * h * wd * cin + 4.0 * nout * (int(out_f32) + int(out_split) + int(next_affine is not None)) + \
            4.0 * cout * cin * kh * kw + (4.0 * nout if residual is not None else 0.0)
        CONV_TRACE.append((lib.cer_conv2d_b3_tile(ctypes.byref(d)), 2.0 * n * ho * wo * cout * cin * kh * kw, e0, e1, nbytes))
    return res


def stem_conv(x, w, scale=None, shift=None, alpha=None, out=None, want_stats=False):
    """IR-50 input layer on the vector ALUs (cer_stem_conv3x3).  x [N,3,H,W] fp32 (NCHW), w the packed [64, Kpad] weight.
    ``out`` = None: the statistics pass -- returns the [rows,2,64] partial sums of the RAW conv result.  Otherwise ``out`` is
    "f32", "split", torch.bfloat16 or torch.float16 and the result is ``prelu(conv*scale+shift, alpha)`` in that storage,
    returned as a dict with 'y' / 'split' / 'n16' and (``want_stats``) 'stats' of that output."""
    _dev_f32(x, "x")
    _dev_f32(w, "w")
    for t, nme in ((scale, "scale"), (shift, "shift"), (alpha, "alpha")):
        _dev_f32(t, nme)
    n, c, h, wd = x.shape
    if c != 3 or w.shape[0] != 64:
        raise ValueError("stem_conv is the 3 -> 64 input layer")
    lib = _lib.load()
    stats = _empty((lib.cer_stem_conv3x3_stats_rows(n, h), 2, 64), x) if (want_stats or out is None) else None
    res, y, hi, lo, storage = {}, None, None, None, STORE_NONE
    if out == "f32":
        res["y"] = y = torch.empty((n, h, wd, 64), device=x.device, dtype=torch.float32)
    elif out == "split":
        res["split"] = Split.empty((n, h, wd, 64), x.device)
        hi, lo = res["split"].hi, res["split"].lo
    elif out is not None:
        storage = storage_of(out)
        res["n16"] = hi = torch.empty((n, h, wd, 64), device=x.device, dtype=out)
    check(lib.cer_stem_conv3x3(ptr(x), ptr(w), w.shape[1], ptr(scale), ptr(shift), ptr(alpha), ptr(y), ptr(hi), ptr(lo), storage,
                               ptr(stats), n, h, wd, current_stream()), "cer_stem_conv3x3")
    if out is None:
        return stats
    if want_stats:
        res["stats"] = stats
    return res


def _tile_desc(n, h, w, cin, cout, kh, kw, stride, pad, x_s2d):
    d = ConvDesc()
    d.N, d.H, d.W, d.Cin, d.Cout = n, h, w, cin, cout
    d.Ho, d.Wo = (h + 2 * pad[0] - kh) // stride + 1, (w + 2 * pad[1] - kw) // stride + 1
    d.KH, d.KW, d.stride, d.dil_h, d.dil_w, d.pad_t, d.pad_l = kh, kw, stride, 1, 1, pad[0], pad[1]
    d.split_k, d.x_s2d = 1, int(x_s2d)
    return d


def conv2d_b3_tile(n, h, w, cin, cout, kh, kw, stride, pad, x_s2d=False):
    """The bf16x3 kernel variant ``conv2d_b3`` would launch for this conv (cer_conv2d_b3_tile)."""
    return _lib.load().cer_conv2d_b3_tile(ctypes.byref(_tile_desc(n, h, w, cin, cout, kh, kw, stride, pad, x_s2d)))


def conv2d_n16_tile(n, h, w, cin, cout, kh, kw, stride, pad, x_s2d=False):
    """The narrow kernel variant ``conv2d_n16`` would launch for this conv (cer_conv2d_n16_tile)."""
    return _lib.load().cer_conv2d_n16_tile(ctypes.byref(_tile_desc(n, h, w, cin, cout, kh, kw, stride, pad, x_s2d)))


# the window / patch kernels: their epilogues can store space-to-depth
S2D_PRODUCER_TILES = (53, 56, 58, 59)
S2D_PRODUCER_TILES_N16 = (71, 72, 73, 76, 77, 78, 79)


def s2d_k_order(cin, device, chunk=32):
    """K-column order of an ``x_s2d`` conv's weights (cer_conv_s2d_k_order) as an index tensor; chunk = channels per kernel
    step: 32 (bf16x3) or 64 (narrow)."""
    order = (ctypes.c_int32 * (9 * cin))()
    check(_lib.load().cer_conv_s2d_k_order(cin, chunk, order), "cer_conv_s2d_k_order")
    return torch.tensor(list(order), dtype=torch.long, device=device)


def pack_s2d_weight(w, cin, chunk=None):
    """Packed 3x3 weights [Cout, 9*Cin] (Split, or a narrow / fp32 plane) -> the same columns in the step order of the
    space-to-depth stride-2 kernel (chunk: 32 for Split operands, 64 for a narrow plane)."""
    if isinstance(w, Split):
        idx = s2d_k_order(cin, w.hi.device, chunk or 32)
        return Split(w.hi.index_select(1, idx).contiguous(), w.lo.index_select(1, idx).contiguous())
    return w.index_select(1, s2d_k_order(cin, w.device, chunk or 64)).contiguous()


def space_to_depth(x):
    """[N, H, W, C] -> [N, H/2, W/2, 4C] with channel blocks ordered P11 | P10 | P01 | P00 (block = 3 - 2*(row & 1) - (col & 1)):
    the torch statement of the layout ``y_s2d`` stores and ``x_s2d`` reads (tests; the product path never makes this copy)."""
    if isinstance(x, Split):
        return Split(space_to_depth(x.hi), space_to_depth(x.lo))
    return torch.cat([x[:, 1::2, 1::2], x[:, 1::2, 0::2], x[:, 0::2, 1::2], x[:, 0::2, 0::2]], dim=3).contiguous()


# ------------------------------------------------------------------ narrow storage (one bf16 / half plane per tensor)
def storage_of(dtype):
    """torch dtype -> cer_storage of the narrow kernels."""
    if dtype == torch.bfloat16:
        return STORE_BF16
    if dtype == torch.float16:
        return STORE_F16
    raise ValueError(f"narrow storage is torch.bfloat16 or torch.float16, got {dtype}")


def _dev_n16(t, name, dtype=None):
    if t is None:
        return
    if not (t.is_cuda and t.dtype in (torch.bfloat16, torch.float16) and t.is_contiguous()) or (dtype is not None and t.dtype != dtype):
        raise ValueError(f"{name}: expected a contiguous {dtype or 'bfloat16/float16'} GPU tensor, got {t.dtype} {t.device}")


def to_n16(x, dtype, scale=None, shift=None):
    """fp32 tensor -> narrow tensor (round-to-nearest-even), optionally after x*scale[c]+shift[c] over the last axis."""
    _dev_f32(x, "x")
    _dev_f32(scale, "scale")
    _dev_f32(shift, "shift")
    out = torch.empty(x.shape, device=x.device, dtype=dtype)
    c = x.shape[-1] if scale is not None else 0
    check(_lib.load().cer_to_n16(ptr(x), ptr(scale), ptr(shift), c, ptr(out), x.numel(), storage_of(dtype), current_stream()),
          "cer_to_n16")
    return out


def from_n16(x):
    _dev_n16(x, "x")
    out = torch.empty(x.shape, device=x.device, dtype=torch.float32)
    check(_lib.load().cer_from_n16(ptr(x), ptr(out), x.numel(), storage_of(x.dtype), current_stream()), "cer_from_n16")
    return out


def conv2d_n16(x, w, kh, kw, *, stride=1, dil=(1, 1), pad=(0, 0), out_hw=None, bias=None, alpha=None, residual=None,
               res_stride=1, act1=ACT_NONE, act2=ACT_NONE, slope=LEAKY_SLOPE, split_k=1, tile=0, out_f32=False,
               out_n16=True, want_stats=False, bias9=None, x_s2d=False, y_s2d=False):
    """Narrow convolution: x [N,H,W,Cin] and w [Cout,Kpad] are bf16 / float16 tensors of the same dtype (one MFMA per
    product, fp32 accumulate); residual: narrow or fp32.  Returns a dict with 'n16' (narrow output), 'y' (fp32), 'stats'.
    ``x_s2d`` / ``y_s2d``: space-to-depth input / narrow output, as in ``conv2d_b3`` (weights: ``pack_s2d_weight(w, cin, 64)``)."""
    lib = _lib.load()
    _dev_n16(x, "x")
    _dev_n16(w, "w", x.dtype)
    _dev_f32(bias, "bias")
    _dev_f32(alpha, "alpha")
    n, h, wd, cin = x.shape
    if x_s2d:
        if cin % 4:
            raise ValueError("a space-to-depth input has 4 * Cin channels")
        h, wd, cin = 2 * h, 2 * wd, cin // 4
    cout = w.shape[0]
    if w.shape[1] != conv_kpad(kh, kw, cin):
        raise ValueError(f"packed weight has K={w.shape[1]}, expected {conv_kpad(kh, kw, cin)}")
    if out_hw is None:
        ho = (h + 2 * pad[0] - dil[0] * (kh - 1) - 1) // stride + 1
        wo = (wd + 2 * pad[1] - dil[1] * (kw - 1) - 1) // stride + 1
    else:
        ho, wo = out_hw
    d = ConvDesc()
    d.N, d.H, d.W, d.Cin, d.Ho, d.Wo, d.Cout = n, h, wd, cin, ho, wo, cout
    d.KH, d.KW, d.stride, d.dil_h, d.dil_w, d.pad_t, d.pad_l = kh, kw, stride, dil[0], dil[1], pad[0], pad[1]
    d.res_stride, d.Hr, d.Wr = res_stride, 0, 0
    d.act1, d.act2, d.slope, d.split_k, d.tile = act1, act2, slope, split_k, tile
    d.storage = storage_of(x.dtype)
    d.x_s2d, d.y_s2d = int(x_s2d), int(y_s2d)
    io = ConvIO()
    io.x_hi, io.w_hi = x.data_ptr(), w.data_ptr()
    io.bias = bias.data_ptr() if bias is not None else None
    io.alpha = alpha.data_ptr() if alpha is not None else None
    if bias9 is not None:
        _dev_f32(bias9, "bias9")
        if tuple(bias9.shape) != (9, cout):
            raise ValueError("bias9 must be [9, Cout]")
        io.bias9 = bias9.data_ptr()
    if residual is not None:
        rshape = residual.shape
        if rshape[0] != n or rshape[3] != cout:
            raise ValueError("residual shape does not match the output")
        d.Hr, d.Wr = rshape[1], rshape[2]
        if residual.dtype == torch.float32:
            _dev_f32(residual, "residual")
            io.residual = residual.data_ptr()
        else:
            _dev_n16(residual, "residual", x.dtype)
            io.res_hi = residual.data_ptr()
    res = {}
    dev = x.device
    if out_f32:
        res["y"] = torch.empty((n, ho, wo, cout), device=dev, dtype=torch.float32)
        io.y = res["y"].data_ptr()
    if out_n16:
        res["n16"] = torch.empty((n, ho // 2, wo // 2, 4 * cout) if y_s2d else (n, ho, wo, cout), device=dev, dtype=x.dtype)
        io.y_hi = res["n16"].data_ptr()
    if want_stats:
        res["stats"] = torch.empty((lib.cer_conv2d_stats_tiles(ctypes.byref(d), 2), 2, cout), device=dev, dtype=torch.float32)
        io.stats = res["stats"].data_ptr()
    ws_bytes = lib.cer_conv2d_workspace_bytes(ctypes.byref(d))
    ws = torch.empty((ws_bytes // 4,), device=dev, dtype=torch.float32) if ws_bytes else None
    if CONV_TRACE is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    check(lib.cer_conv2d_run(ctypes.byref(d), ctypes.byref(io), ptr(ws), ws_bytes, current_stream()), "cer_conv2d_run")
    if CONV_TRACE is not None:
        e1.record()
        # algorithmic bytes: narrow input + every output tensor written + narrow weights + residual read
        nout = n * ho * wo * cout
        nbytes = 2.0 * n * h * wd * cin + nout * (4.0 * int(out_f32) + 2.0 * int(out_n16)) + 2.0 * cout * cin * kh * kw + \
            (0.0 if residual is None else (4.0 if residual.dtype == torch.float32 else 2.0) * nout)
        CONV_TRACE.append((lib.cer_conv2d_n16_tile(ctypes.byref(d)), 2.0 * n * ho * wo * cout * cin * kh * kw, e0, e1, nbytes))
    return res


def bn_apply_nhwc_n16(y, scale, shift, dtype=None, alpha=None, res=None, res_stride=1, res_scale=None, res_shift=None, mask=None,
                      want_stats=False, out_f32=False, out_n16=True):
    """``bn_apply_nhwc`` for the narrow encoder: ``y`` (the conv result) and ``res`` are fp32 or narrow tensors; the result
    comes back narrow ('n16', what the next conv reads) and/or fp32 ('y'); 'stats' as requested."""
    for t, nme in ((scale, "scale"), (shift, "shift"), (alpha, "alpha"), (res_scale, "res_scale"), (res_shift, "res_shift"),
                   (mask, "mask")):
        _dev_f32(t, nme)
    n, ho, wo, c = y.shape
    lib = _lib.load()
    if dtype is None:
        dtype = y.dtype if y.dtype != torch.float32 else (res.dtype if res is not None else None)
    if dtype is None or dtype == torch.float32:
        raise ValueError("bn_apply_nhwc_n16: pass dtype= when neither y nor res is a narrow tensor")
    y32 = y16 = r32 = r16 = None
    if y.dtype == torch.float32:
        _dev_f32(y, "y")
        y32 = y
    else:
        _dev_n16(y, "y", dtype)
        y16 = y
    hr = wr = 0
    if res is not None:
        hr, wr = res.shape[1], res.shape[2]
        if res.dtype == torch.float32:
            _dev_f32(res, "res")
            r32 = res
        else:
            _dev_n16(res, "res", dtype)
            r16 = res
    out = {}
    if out_f32:
        out["y"] = torch.empty(tuple(y.shape), device=y.device, dtype=torch.float32)
    if out_n16:
        out["n16"] = torch.empty(tuple(y.shape), device=y.device, dtype=dtype)
    if want_stats:
        out["stats"] = _empty((lib.cer_bn_apply_stats_tiles(n * ho * wo), 2, c), scale)
    check(lib.cer_bn_apply_nhwc_n16(ptr(y32), ptr(y16), ptr(scale), ptr(shift), ptr(alpha), ptr(r32), ptr(r16), ptr(res_scale),
                                    ptr(res_shift), ptr(mask), ptr(out.get("y")), ptr(out.get("n16")), ptr(out.get("stats")),
                                    n, ho, wo, c, res_stride, hr, wr, storage_of(dtype), current_stream()),
          "cer_bn_apply_nhwc_n16")
    return out


AUTO_SPLIT_K = os.environ.get("CER_TAIL_SPLITK", "1") != "0"


def auto_split_k(m, cout, kdim):
    """Split-K factor for the small GEMMs of the trainable tail (M = B*L rows): a 1024 x 128 output is 32 tiles of 64 x 64
    on a 256-CU chip, so the K loop is cut until ~256 blocks are in flight (deterministic slabs + one reduce launch)."""
    if not AUTO_SPLIT_K:
        return 1
    tiles = ((m + 63) // 64) * ((cout + 63) // 64)
    if tiles >= 128 or kdim < 256:
        return 1
    return max(1, min(kdim // 128, 256 // tiles, 16))


def linear(x2d, w_packed, bias=None, act=ACT_NONE, split_k=1, residual=None, out=None):
    """[M,K] @ W[Cout,K]^T as a 1x1 conv on an [M,1,1,K] image.  x2d / out may be column
    slices of wider row-major buffers."""
    m, k, x_ld = _rows(x2d, "x2d")
    res = residual.view(m, 1, 1, -1) if residual is not None else None
    y_ld = 0
    if out is not None:
        _, _, y_ld = _rows(out, "out")
        if y_ld == out.shape[1]:
            y_ld = 0
    y = conv2d(x2d, w_packed, 1, 1, bias=bias, act1=act, split_k=split_k, residual=res, out=out,
               x_ld=(0 if x_ld == k else x_ld), y_ld=y_ld, x_shape=(m, 1, 1, k))
    return y if out is not None else y.view(m, -1)


def l2norm_rows(x):
    _dev_f32(x, "x")
    y = torch.empty_like(x)
    check(_lib.load().cer_l2norm_rows(ptr(x), ptr(y), x.shape[0], x.shape[1], current_stream()),
          "cer_l2norm_rows")
    return y


# bench.py sets this to a list to time every matrix-core weight-gradient launch: (Cout, Cin, taps, algorithmic FLOPs, start, end)
WGRAD_TRACE = None


def _wgrad_traced(fn, n, ho, wo, cout, cin, kh, kw):
    if WGRAD_TRACE is None:
        return fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    fn()
    e1.record()
    WGRAD_TRACE.append((cout, cin, kh * kw, 2.0 * n * ho * wo * cout * cin * kh * kw, e0, e1))


def conv2d_wgrad(dz, x, kh, kw, stride=1, pad=(0, 0), b3=False):
    """dW [Cout,Cin,KH,KW] (torch layout) from dz [N,Ho,Wo,Cout] and the conv input x [N,H,W,Cin] (both dense NHWC).
    ``b3``: the bf16x3 MFMA kernel with the pixel range split over blocks (Cout, Cin % 4 == 0; other shapes take the
    fp32-MFMA kernel either way).  ``dz`` / ``x`` may be ``Split`` tensors (then both are used split: no conversion in the
    kernel's loader)."""
    split_in = isinstance(dz, Split) or isinstance(x, Split)
    if split_in:
        dz = dz if isinstance(dz, Split) else split_bf16(dz)
        x = x if isinstance(x, Split) else split_bf16(x)
        n, ho, wo, cout = dz.hi.shape
        _, h, w, cin = x.hi.shape
        if cout % 4 or cin % 4:
            raise ValueError("conv2d_wgrad on split tensors needs channel counts that are multiples of 4")
        dw = torch.empty((cout, cin, kh, kw), device=dz.hi.device, dtype=torch.float32)
        lib = _lib.load()
        nbytes = lib.cer_conv2d_wgrad_b3_workspace_bytes(n, ho, wo, cout, cin, kh, kw)
        ws = torch.empty((nbytes // 4,), device=dw.device, dtype=torch.float32) if nbytes else None
        _wgrad_traced(lambda: check(lib.cer_conv2d_wgrad_b3s(ptr(dz.hi), ptr(dz.lo), ptr(x.hi), ptr(x.lo), ptr(dw), n, h, w, ho, wo,
                                                              cout, cin, kh, kw, stride, pad[0], pad[1], ptr(ws), nbytes,
                                                              current_stream()), "cer_conv2d_wgrad_b3s"), n, ho, wo, cout, cin, kh, kw)
        return dw
    _dev_f32(dz, "dz")
    _dev_f32(x, "x")
    n, ho, wo, cout = dz.shape
    _, h, w, cin = x.shape
    dw = _empty((cout, cin, kh, kw), dz)
    if b3 and cout % 4 == 0 and cin % 4 == 0:
        lib = _lib.load()
        nbytes = lib.cer_conv2d_wgrad_b3_workspace_bytes(n, ho, wo, cout, cin, kh, kw)
        ws = _empty((nbytes // 4,), dz) if nbytes else None
        _wgrad_traced(lambda: check(lib.cer_conv2d_wgrad_b3(ptr(dz), ptr(x), ptr(dw), n, h, w, ho, wo, cout, cin, kh, kw, stride,
                                                             pad[0], pad[1], ptr(ws), nbytes, current_stream()),
                                    "cer_conv2d_wgrad_b3"), n, ho, wo, cout, cin, kh, kw)
        return dw
    lib = _lib.load()
    nbytes = lib.cer_conv_wgrad_workspace_bytes(n * ho * wo, cout, cin, kh * kw)
    ws = _empty((nbytes // 4,), dz) if nbytes else None
    check(lib.cer_conv2d_wgrad(ptr(dz), ptr(x), ptr(dw), n, h, w, ho, wo, cout, cin, kh, kw, stride, pad[0], pad[1],
                               ptr(ws), nbytes, current_stream()), "cer_conv2d_wgrad")
    return dw


def prelu_fwd(x, alpha):
    _dev_f32(x, "x")
    _dev_f32(alpha, "alpha")
    y = torch.empty_like(x)
    check(_lib.load().cer_prelu_fwd(ptr(x), ptr(alpha), ptr(y), x.numel() // x.shape[-1], x.shape[-1], current_stream()),
          "cer_prelu_fwd")
    return y


def prelu_split(x, alpha):
    """prelu(x) as a Split tensor in one pass (the conv input a released unit rebuilds from its raw conv result)."""
    _dev_f32(x, "x")
    _dev_f32(alpha, "alpha")
    out = Split.empty(x.shape, x.device)
    c = x.shape[-1]
    check(_lib.load().cer_prelu_split(ptr(x), ptr(alpha), ptr(out.hi), ptr(out.lo), x.numel() // c, c, current_stream()),
          "cer_prelu_split")
    return out


def prelu_bwd(dy, x, alpha, out=None, split_out=False):
    """-> (dx, dalpha): torch's PReLU backward for channels-last tensors.  ``split_out``: dx comes back as a Split tensor (no fp32
    copy is written); ``out``: write dx into this fp32 tensor / Split (a slice of a larger one: chunked calls)."""
    for t, nme in ((dy, "dy"), (x, "x"), (alpha, "alpha")):
        _dev_f32(t, nme)
    terms = torch.empty_like(x)
    c = x.shape[-1]
    if split_out:
        dx = out if out is not None else Split.empty(x.shape, x.device)
        _dev_bf16(dx.hi, "out.hi")
        _dev_bf16(dx.lo, "out.lo")
        check(_lib.load().cer_prelu_bwd_split(ptr(dy), ptr(x), ptr(alpha), None, ptr(dx.hi), ptr(dx.lo), ptr(terms), x.numel() // c, c,
                                              current_stream()), "cer_prelu_bwd_split")
    else:
        dx = out if out is not None else torch.empty_like(x)
        _dev_f32(dx, "out")
        check(_lib.load().cer_prelu_bwd(ptr(dy), ptr(x), ptr(alpha), ptr(dx), ptr(terms), x.numel() // c, c, current_stream()),
              "cer_prelu_bwd")
    return dx, col_sum(terms.view(-1, c))


def l2norm_rows_bwd(dy, x):
    """Gradient of ``l2norm_rows`` w.r.t. its input ``x`` (the un-normalised rows)."""
    _dev_f32(dy, "dy")
    _dev_f32(x, "x")
    dx = torch.empty_like(x)
    check(_lib.load().cer_l2norm_rows_bwd(ptr(dy), ptr(x), ptr(dx), x.shape[0], x.shape[1], current_stream()),
          "cer_l2norm_rows_bwd")
    return dx


def maxpool2x2_nhwc(x):
    _dev_f32(x, "x")
    n, h, w, c = x.shape
    y = _empty((n, h // 2, w // 2, c), x)
    check(_lib.load().cer_maxpool2x2_nhwc(ptr(x), ptr(y), n, h, w, c, current_stream()), "cer_maxpool2x2_nhwc")
    return y


# ------------------------------------------------------------------ train-mode BatchNorm2d (encoder)
def bn_finalize(partials, count, gamma, beta, running_mean=None, running_var=None, momentum=0.1, eps=1e-5):
    """[tiles,2,C] partial sums -> (scale, shift) of the batch-statistics BatchNorm; running stats
    are updated in place like torch does in train mode."""
    _dev_f32(partials, "partials")
    tiles, _, c = partials.shape
    scale, shift = _empty((c,), partials), _empty((c,), partials)
    lib = _lib.load()
    nbytes = lib.cer_bn_finalize_workspace_bytes(tiles, c)
    ws = torch.empty((nbytes // 8,), device=partials.device, dtype=torch.float64)
    check(lib.cer_bn_finalize(ptr(partials), tiles, c, float(count), ptr(gamma), ptr(beta), ptr(running_mean),
                              ptr(running_var), momentum, eps, ptr(scale), ptr(shift), ptr(ws), nbytes,
                              current_stream()), "cer_bn_finalize")
    return scale, shift


def bn_apply_nhwc(y, scale, shift, alpha=None, res=None, res_stride=1, res_scale=None, res_shift=None, mask=None,
                  want_stats=False):
    """out = mask*prelu(y*scale+shift) + (res*res_scale+res_shift) on NHWC; optionally the partial
    statistics of ``out`` for the next BatchNorm."""
    for t, nme in ((y, "y"), (scale, "scale"), (shift, "shift"), (alpha, "alpha"), (res, "res"),
                   (res_scale, "res_scale"), (res_shift, "res_shift"), (mask, "mask")):
        _dev_f32(t, nme)
    n, ho, wo, c = y.shape
    lib = _lib.load()
    out = torch.empty_like(y)
    stats = _empty((lib.cer_bn_apply_stats_tiles(n * ho * wo), 2, c), y) if want_stats else None
    hr, wr = (res.shape[1], res.shape[2]) if res is not None else (0, 0)
    check(lib.cer_bn_apply_nhwc(ptr(y), ptr(scale), ptr(shift), ptr(alpha), ptr(res), ptr(res_scale), ptr(res_shift),
                                ptr(mask), ptr(out), ptr(stats), n, ho, wo, c, res_stride, hr, wr, current_stream()),
          "cer_bn_apply_nhwc")
    return (out, stats) if want_stats else out


def bn_apply_nhwc_b3(y, scale, shift, alpha=None, res=None, res_stride=1, res_scale=None, res_shift=None, mask=None,
                     want_stats=False, out_f32=False, out_split=True):
    """``bn_apply_nhwc`` for the bf16x3 encoder: ``res`` may be a Split tensor, the result comes back as a Split
    (what the next conv reads) and/or fp32.  Returns a dict with 'split', 'y', 'stats' as requested."""
    for t, nme in ((y, "y"), (scale, "scale"), (shift, "shift"), (alpha, "alpha"), (res_scale, "res_scale"),
                   (res_shift, "res_shift"), (mask, "mask")):
        _dev_f32(t, nme)
    n, ho, wo, c = y.shape
    lib = _lib.load()
    r_f32 = r_hi = r_lo = None
    hr = wr = 0
    if res is not None:
        hr, wr = res.shape[1], res.shape[2]
        if isinstance(res, Split):
            _dev_bf16(res.hi, "res.hi")
            r_hi, r_lo = res.hi, res.lo
        else:
            _dev_f32(res, "res")
            r_f32 = res
    out = {}
    if out_f32:
        out["y"] = torch.empty_like(y)
    if out_split:
        out["split"] = Split.empty(tuple(y.shape), y.device)
    if want_stats:
        out["stats"] = _empty((lib.cer_bn_apply_stats_tiles(n * ho * wo), 2, c), y)
    sp = out.get("split")
    check(lib.cer_bn_apply_nhwc_b3(ptr(y), ptr(scale), ptr(shift), ptr(alpha), ptr(r_f32), ptr(r_hi), ptr(r_lo),
                                   ptr(res_scale), ptr(res_shift), ptr(mask), ptr(out.get("y")),
                                   ptr(sp.hi) if sp is not None else None, ptr(sp.lo) if sp is not None else None,
                                   ptr(out.get("stats")), n, ho, wo, c, res_stride, hr, wr, current_stream()),
          "cer_bn_apply_nhwc_b3")
    return out


def fold_bn_3x3_packed(w_packed, scale, shift, out="f32"):
    """Fold a per-input-channel affine (the BatchNorm in FRONT of a 3x3 / pad 1 / stride 1 conv) into the PACKED weight
    [Cout, Kpad] in one launch: returns (w * scale[cin] in the requested form, bias9 [9, Cout]).  conv(pad0(s*x + t)) =
    conv'(pad0(x)) + sum over the taps INSIDE the image of W_tap . t, which only depends on whether the output pixel sits
    on the first / an inner / the last row and column (9 cases).  ``out``: "f32", "split" (Split) or a narrow torch dtype."""
    _dev_f32(w_packed, "w_packed")
    _dev_f32(scale, "scale")
    _dev_f32(shift, "shift")
    cout, kpad = w_packed.shape
    cin = scale.numel()
    if kpad != conv_kpad(3, 3, cin) or shift.numel() != cin:
        raise ValueError("fold_bn_3x3_packed: w_packed must be the packing of a [Cout, Cin, 3, 3] weight")
    b9 = _empty((9, cout), w_packed)
    wf = hi = lo = None
    storage = STORE_NONE
    if out == "f32":
        res = wf = torch.empty_like(w_packed)
    elif out == "split":
        res = Split.empty((cout, kpad), w_packed.device)
        hi, lo = res.hi, res.lo
    else:
        storage = storage_of(out)
        res = hi = torch.empty((cout, kpad), device=w_packed.device, dtype=out)
    check(_lib.load().cer_fold_bn_3x3(ptr(w_packed), ptr(scale), ptr(shift), cout, cin, ptr(wf), ptr(hi), ptr(lo), storage,
                                      ptr(b9), current_stream()), "cer_fold_bn_3x3")
    return res, b9


def fold_input_bn_3x3(w_oihw, scale, shift):
    """``fold_bn_3x3_packed`` from the torch-layout weight [Cout, Cin, 3, 3]: (packed fp32 weight, bias9)."""
    _dev_f32(w_oihw, "w")
    if w_oihw.shape[2] != 3 or w_oihw.shape[3] != 3:
        raise ValueError("fold_input_bn_3x3: 3x3 kernels only")
    return fold_bn_3x3_packed(pack_conv_weight(w_oihw.contiguous()), scale.contiguous(), shift.contiguous(), "f32")


def gather_rows(src, index):
    """out[i] = src[index[i]] (zeros where index < 0); src [R, C] fp32, index int64 on the same device."""
    _dev_f32(src, "src")
    if index.dtype != torch.int64 or not index.is_cuda or not index.is_contiguous():
        raise ValueError("index: expected a contiguous int64 GPU tensor")
    out = torch.empty((index.numel(), src.shape[1]), device=src.device, dtype=torch.float32)
    check(_lib.load().cer_gather_rows(ptr(src), ptr(index), ptr(out), index.numel(), src.shape[1], src.shape[0],
                                      current_stream()), "cer_gather_rows")
    return out


def sgd_nesterov_flat(param, grad, buf, lr, momentum=0.9, dampening=0.0, weight_decay=0.0, nesterov=True, first_step=False):
    for t, nme in ((param, "param"), (grad, "grad"), (buf, "buf")):
        _dev_f32(t, nme)
    check(_lib.load().cer_sgd_nesterov_flat(ptr(param), ptr(grad), ptr(buf), param.numel(), lr, momentum, dampening,
                                            weight_decay, int(nesterov), int(first_step), current_stream()),
          "cer_sgd_nesterov_flat")


# ------------------------------------------------------------------ trainable tail
def weight_norm_fwd(v, g):
    """v [Cout,Cin,k], g [Cout,1,1] -> (w [Cout,Cin,k], norm [Cout])."""
    _dev_f32(v, "v")
    _dev_f32(g, "g")
    rows, e = v.shape[0], v.numel() // v.shape[0]
    w, norm = torch.empty_like(v), _empty((rows,), v)
    check(_lib.load().cer_weight_norm_fwd(ptr(v), ptr(g), ptr(w), ptr(norm), rows, e, current_stream()),
          "cer_weight_norm_fwd")
    return w, norm


def weight_norm_fwd_packed(v, g):
    """``weight_norm_fwd`` of a [Cout,Cin,k] filter plus, from the same launch, the packed forward weight [Cout,Kpad(k,1,Cin)] and
    the flipped / transposed data-gradient weight [Cin,Kpad(k,1,Cout)] (``pack_conv_weight`` twice): (w, norm, wp, wt)."""
    _dev_f32(v, "v")
    _dev_f32(g, "g")
    cout, cin, k = v.shape
    w, norm = torch.empty_like(v), _empty((cout,), v)
    wp, wt = _empty((cout, conv_kpad(k, 1, cin)), v), _empty((cin, conv_kpad(k, 1, cout)), v)
    check(_lib.load().cer_weight_norm_fwd_packed(ptr(v), ptr(g), ptr(w), ptr(norm), ptr(wp), ptr(wt), cout, cin, k, current_stream()),
          "cer_weight_norm_fwd_packed")
    return w, norm, wp, wt


def conv1d_wgrad_weight_norm_bwd(dz, x, seq_len, k, dil, v, g, norm):
    """(dv, dg) of a weight-normed causal conv from dz [R,Cout] and the layer input x [R,Cin]: ``conv1d_wgrad`` +
    ``weight_norm_bwd`` with the split-R partial sums folded inside the second kernel (two launches)."""
    r, cout, dz_ld = _rows(dz, "dz")
    r2, cin, x_ld = _rows(x, "x")
    if r != r2:
        raise ValueError("dz and x must have the same number of rows")
    for t, n in ((v, "v"), (g, "g"), (norm, "norm")):
        _dev_f32(t, n)
    lib = _lib.load()
    nbytes = max(lib.cer_conv_wgrad_workspace_bytes(r, cout, cin, k), cout * cin * k * 4)
    ws = _empty((nbytes // 4,), dz)
    dv, dg = torch.empty_like(v), torch.empty_like(g)       # (dg in g's [Cout,1,1] shape, as weight_norm_bwd returns it)
    check(lib.cer_conv1d_wgrad_weight_norm_bwd(ptr(dz), dz_ld, ptr(x), x_ld, r, seq_len, cout, cin, k, dil, ptr(v), ptr(g), ptr(norm),
                                               ptr(dv), ptr(dg), ptr(ws), nbytes, current_stream()), "cer_conv1d_wgrad_weight_norm_bwd")
    return dv, dg


def weight_norm_bwd(dw, v, g, norm):
    for t, n in ((dw, "dw"), (v, "v"), (g, "g"), (norm, "norm")):
        _dev_f32(t, n)
    rows, e = v.shape[0], v.numel() // v.shape[0]
    dv, dg = torch.empty_like(v), torch.empty_like(g)
    check(_lib.load().cer_weight_norm_bwd(ptr(dw), ptr(v), ptr(g), ptr(norm), ptr(dv), ptr(dg), rows, e,
                                          current_stream()), "cer_weight_norm_bwd")
    return dv, dg


def conv1d_wgrad(dz, x, seq_len, k, dil):
    """dW [Cout,Cin,k] from dz [R,Cout] and the layer input x [R,Cin] (both may be column slices)."""
    r, cout, dz_ld = _rows(dz, "dz")
    r2, cin, x_ld = _rows(x, "x")
    if r != r2:
        raise ValueError("dz and x must have the same number of rows")
    dw = _empty((cout, cin, k), dz)
    lib = _lib.load()
    nbytes = lib.cer_conv_wgrad_workspace_bytes(r, cout, cin, k)
    ws = _empty((nbytes // 4,), dz) if nbytes else None
    check(lib.cer_conv1d_wgrad(ptr(dz), dz_ld, ptr(x), x_ld, ptr(dw), r, seq_len, cout, cin, k, dil, ptr(ws), nbytes,
                               current_stream()), "cer_conv1d_wgrad")
    return dw


def _col_ws(r, c, like):
    nbytes = _lib.load().cer_col_sum_workspace_bytes(r, c)
    return (_empty((nbytes // 4,), like) if nbytes else None), nbytes


def col_sum(a):
    """Column sums of a [R,C] tensor (bias gradients)."""
    r, c, ld = _rows(a, "a")
    out = _empty((c,), a)
    ws, nbytes = _col_ws(r, c, a)
    check(_lib.load().cer_col_sum(ptr(a), ld, None, 0, None, None, ptr(out), r, c, ptr(ws), nbytes,
                                  current_stream()), "cer_col_sum")
    return out


def act_mask_bwd(dy, y, mask=None, slope=LEAKY_SLOPE):
    for t, n in ((dy, "dy"), (y, "y"), (mask, "mask")):
        _dev_f32(t, n)
    dz = torch.empty_like(dy)
    check(_lib.load().cer_act_mask_bwd(ptr(dy), ptr(y), ptr(mask), ptr(dz), dy.numel(), slope, current_stream()),
          "cer_act_mask_bwd")
    return dz


def tblock_tail_bwd(dout, out, a2, mask2=None, slope=LEAKY_SLOPE):
    for t, n in ((dout, "dout"), (out, "out"), (a2, "a2"), (mask2, "mask2")):
        _dev_f32(t, n)
    du, dz2 = torch.empty_like(dout), torch.empty_like(dout)
    check(_lib.load().cer_tblock_tail_bwd(ptr(dout), ptr(out), ptr(a2), ptr(mask2), ptr(du), ptr(dz2),
                                          dout.numel(), slope, current_stream()), "cer_tblock_tail_bwd")
    return du, dz2


def bn_rows_fwd(x, w, b, running_mean, running_var, train, eps=1e-5, momentum=0.1, out=None):
    """BatchNorm over the rows of x [R,C].  Returns (y, save_mean, save_invstd); the saves are
    None in eval mode.  Running stats are updated in place when ``train``."""
    r, c, x_ld = _rows(x, "x")
    for t, n in ((w, "w"), (b, "b"), (running_mean, "running_mean"), (running_var, "running_var")):
        _dev_f32(t, n)
    if out is None:
        out = _empty((r, c), x)
    _, _, y_ld = _rows(out, "out")
    sm = _empty((c,), x) if train else None
    si = _empty((c,), x) if train else None
    lib = _lib.load()
    nbytes = lib.cer_bn_rows_fwd_workspace_bytes(r, c) if train else 0
    ws = _empty((nbytes // 4,), x) if nbytes else None
    check(lib.cer_bn_rows_fwd(ptr(x), x_ld, ptr(w), ptr(b), ptr(running_mean), ptr(running_var), ptr(sm),
                              ptr(si), ptr(out), y_ld, r, c, 1 if train else 0, eps, momentum, ptr(ws), nbytes,
                              current_stream()), "cer_bn_rows_fwd")
    return out, sm, si


def bn_rows_stats(x, running_mean, running_var, eps=1e-5, momentum=0.1):
    """Train-mode statistics of the rows of x [R,C] alone: (save_mean, save_invstd), running buffers updated in place like
    ``bn_rows_fwd(train=True)``; no output tensor is written."""
    r, c, x_ld = _rows(x, "x")
    _dev_f32(running_mean, "running_mean")
    _dev_f32(running_var, "running_var")
    sm, si = _empty((c,), x), _empty((c,), x)
    lib = _lib.load()
    nbytes = lib.cer_bn_rows_fwd_workspace_bytes(r, c)
    ws = _empty((nbytes // 4,), x) if nbytes else None
    check(lib.cer_bn_rows_fwd(ptr(x), x_ld, None, None, ptr(running_mean), ptr(running_var), ptr(sm), ptr(si), None, 0, r, c, 1,
                              eps, momentum, ptr(ws), nbytes, current_stream()), "cer_bn_rows_fwd")
    return sm, si


def bn_rows_bwd(dy, x, save_mean, save_invstd, w, train=True, split_out=False, add=None):
    """``split_out`` (train mode, dense rows): dx as a Split tensor written by the apply pass itself.  ``add`` (train mode,
    dense rows, C % 4 == 0): dx = BatchNorm-backward(dy) + add in the same pass."""
    if add is not None:
        for t, nme in ((dy, "dy"), (x, "x"), (add, "add")):
            _dev_f32(t, nme)
        r, c = dy.shape
        if split_out or tuple(add.shape) != (r, c) or c % 4:
            raise ValueError("bn_rows_bwd(add=...): fp32 result, add of the rows' shape, C % 4 == 0")
        dx, dw, db = _empty((r, c), x), _empty((c,), x), _empty((c,), x)
        ws, nbytes = _col_ws(r, c, x)
        check(_lib.load().cer_bn_rows_bwd_add(ptr(dy), ptr(x), ptr(save_mean), ptr(save_invstd), ptr(w), ptr(add), ptr(dx), ptr(dw),
                                              ptr(db), r, c, ptr(ws), nbytes, current_stream()), "cer_bn_rows_bwd_add")
        return dx, dw, db
    if split_out:
        _dev_f32(dy, "dy")
        _dev_f32(x, "x")
        r, c = dy.shape
        dx, dw, db = Split.empty((r, c), x.device), _empty((c,), x), _empty((c,), x)
        ws, nbytes = _col_ws(r, c, x)
        check(_lib.load().cer_bn_rows_bwd_split(ptr(dy), ptr(x), ptr(save_mean), ptr(save_invstd), ptr(w), ptr(dx.hi), ptr(dx.lo),
                                                ptr(dw), ptr(db), r, c, ptr(ws), nbytes, current_stream()), "cer_bn_rows_bwd_split")
        return dx, dw, db
    r, c, dy_ld = _rows(dy, "dy")
    _, _, x_ld = _rows(x, "x")
    dx, dw, db = _empty((r, c), x), _empty((c,), x), _empty((c,), x)
    ws, nbytes = _col_ws(r, c, x)
    check(_lib.load().cer_bn_rows_bwd(ptr(dy), dy_ld, ptr(x), x_ld, ptr(save_mean), ptr(save_invstd), ptr(w),
                                      ptr(dx), ptr(dw), ptr(db), r, c, 1 if train else 0, ptr(ws), nbytes,
                                      current_stream()), "cer_bn_rows_bwd")
    return dx, dw, db


def _ptr_array(tensors):
    arr = (ctypes.c_void_p * len(tensors))()
    for i, t in enumerate(tensors):
        arr[i] = t.data_ptr()
    return arr


def lfan_attn_fwd(qkv_list, num_heads, head_dim):
    """qkv_list: per modality [R, H*3*hd] -> (vals [R, H*M*hd], probs [R,H,M,M])."""
    for t in qkv_list:
        _dev_f32(t, "qkv")
    r, m = qkv_list[0].shape[0], len(qkv_list)
    vals = _empty((r, num_heads * m * head_dim), qkv_list[0])
    probs = _empty((r, num_heads, m, m), qkv_list[0])
    check(_lib.load().cer_lfan_attn_fwd(_ptr_array(qkv_list), ptr(vals), ptr(probs), r, num_heads, m, head_dim,
                                        current_stream()), "cer_lfan_attn_fwd")
    return vals, probs


def lfan_attn_bwd(qkv_list, dvals, probs, num_heads, head_dim):
    _dev_f32(dvals, "dvals")
    r, m = qkv_list[0].shape[0], len(qkv_list)
    dqkv = [torch.empty_like(t) for t in qkv_list]
    check(_lib.load().cer_lfan_attn_bwd(_ptr_array(qkv_list), ptr(dvals), ptr(probs), _ptr_array(dqkv), r,
                                        num_heads, m, head_dim, current_stream()), "cer_lfan_attn_bwd")
    return dqkv


def layernorm_fwd(x, gamma, beta, mask=None, eps=1e-5, out=None, save=True):
    _dev_f32(x, "x")
    _dev_f32(mask, "mask")
    r, c = x.shape
    if out is None:
        out = _empty((r, c), x)
    _, _, y_ld = _rows(out, "out")
    mean = _empty((r,), x) if save else None
    rstd = _empty((r,), x) if save else None
    check(_lib.load().cer_layernorm_fwd(ptr(x), ptr(mask), ptr(gamma), ptr(beta), ptr(out), y_ld, ptr(mean),
                                        ptr(rstd), r, c, eps, current_stream()), "cer_layernorm_fwd")
    return out, mean, rstd


def layernorm_bwd(dy, x, gamma, mean, rstd, mask=None):
    r, c, dy_ld = _rows(dy, "dy")
    dx, scratch = _empty((r, c), x), _empty((r, c), x)
    dg, db = _empty((c,), x), _empty((c,), x)
    ws, nbytes = _col_ws(r, c, x)
    check(_lib.load().cer_layernorm_bwd(ptr(dy), dy_ld, ptr(x), ptr(mask), ptr(gamma), ptr(mean), ptr(rstd), ptr(dx),
                                        ptr(dg), ptr(db), ptr(scratch), r, c, ptr(ws), nbytes, current_stream()),
          "cer_layernorm_bwd")
    return dx, dg, db


class _LabelCheck:
    """Deferred check of the kernel's bad-label counter: the count travels to pinned host memory asynchronously and is
    looked at on the NEXT call (or by ``flush()``), so validating the labels costs no device synchronisation per step."""
    pending = []

    @classmethod
    def push(cls, count_dev):
        host = torch.empty((1,), dtype=torch.int32, pin_memory=True)
        host.copy_(count_dev, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        cls.pending.append((host, ev))

    @classmethod
    def poll(cls, wait=False):
        keep = []
        for host, ev in cls.pending:
            if wait:
                ev.synchronize()
            if ev.query():
                if int(host[0]) != 0:
                    cls.pending = []
                    raise IndexError(f"cross_entropy: {int(host[0])} target(s) out of bounds (valid: 0 .. n_classes-1, or "
                                     "-100 = ignore_index), as nn.CrossEntropyLoss raises; the step's loss is NaN")
            else:
                keep.append((host, ev))
        cls.pending = keep


def flush_label_check():
    """Wait for the outstanding label checks (call before trusting a finished epoch; tests call it)."""
    _LabelCheck.poll(wait=True)


def cross_entropy(logits2d, labels_f32, want_grad=True, check_labels=True):
    """Mean CE over rows; labels are float32 class ids (cast like ``.long()``), -100 = ignore_index.  Returns
    (loss scalar tensor, dlogits or None).  Out-of-range labels raise IndexError -- from this call when the counter of an
    earlier call has arrived, at the latest from ``flush_label_check()``; the affected step's loss is NaN either way."""
    _dev_f32(logits2d, "logits")
    _dev_f32(labels_f32, "labels")
    r, c = logits2d.shape
    if labels_f32.numel() != r:
        raise ValueError(f"cross_entropy: {labels_f32.numel()} labels for {r} rows")
    loss = _empty((), logits2d)
    dl = torch.empty_like(logits2d) if want_grad else None
    bad = torch.empty((1,), device=logits2d.device, dtype=torch.int32) if check_labels else None
    if check_labels:
        _LabelCheck.poll()
    check(_lib.load().cer_cross_entropy(ptr(logits2d), ptr(labels_f32), ptr(loss), ptr(dl), ptr(bad), r, c, current_stream()),
          "cer_cross_entropy")
    if check_labels:
        _LabelCheck.push(bad)
    return loss, dl


def dropout_mask(shape, p, seed, offset, device):
    m = torch.empty(shape, device=device, dtype=torch.float32)
    check(_lib.load().cer_dropout_mask(ptr(m), m.numel(), p, seed, offset, current_stream()), "cer_dropout_mask")
    return m


def leaky_relu(x, slope=LEAKY_SLOPE):
    _dev_f32(x, "x")
    y = torch.empty_like(x)
    check(_lib.load().cer_leaky_relu_fwd(ptr(x), ptr(y), x.numel(), slope, current_stream()), "cer_leaky_relu_fwd")
    return y


def softmax_gate_fwd(z, c):
    _dev_f32(z, "z")
    _dev_f32(c, "c")
    out, prob = torch.empty_like(z), torch.empty_like(z)
    check(_lib.load().cer_softmax_gate_fwd(ptr(z), ptr(c), ptr(out), ptr(prob), z.shape[0], z.shape[1], current_stream()),
          "cer_softmax_gate_fwd")
    return out, prob


def softmax_gate_bwd(dout, prob, c):
    for t, n in ((dout, "dout"), (prob, "prob"), (c, "c")):
        _dev_f32(t, n)
    dz, dc = torch.empty_like(dout), torch.empty_like(dout)
    check(_lib.load().cer_softmax_gate_bwd(ptr(dout), ptr(prob), ptr(c), ptr(dz), ptr(dc), dout.shape[0], dout.shape[1],
                                           current_stream()), "cer_softmax_gate_bwd")
    return dz, dc


def copy_cols(x, out):
    r, c, x_ld = _rows(x, "x")
    _, _, y_ld = _rows(out, "out")
    check(_lib.load().cer_copy_cols(ptr(x), x_ld, ptr(out), y_ld, r, c, current_stream()), "cer_copy_cols")
    return out


# ------------------------------------------------------------------ audio / text encoders
def logmel(pcm_int16, pad_samples, mel_matrix_f64, log_offset=0.01):
    """pcm [clips, S] int16 (16 kHz) -> log-mel [clips, frames, 64] float32."""
    if not (pcm_int16.is_cuda and pcm_int16.dtype == torch.int16 and pcm_int16.is_contiguous() and pcm_int16.dim() == 2):
        raise ValueError("pcm: expected a contiguous [clips, samples] int16 GPU tensor")
    if not (mel_matrix_f64.is_cuda and mel_matrix_f64.dtype == torch.float64 and tuple(mel_matrix_f64.shape) == (257, 64)
            and mel_matrix_f64.is_contiguous()):
        raise ValueError("mel matrix: expected a contiguous [257, 64] float64 GPU tensor")
    lib = _lib.load()
    clips, n = pcm_int16.shape
    frames = lib.cer_logmel_num_frames(n, pad_samples)
    out = torch.empty((clips, frames, 64), device=pcm_int16.device, dtype=torch.float32)
    check(lib.cer_logmel_fwd(ptr(pcm_int16), clips, n, pad_samples, ptr(mel_matrix_f64), log_offset, ptr(out),
                             current_stream()), "cer_logmel_fwd")
    return out


def frame_examples(logmel_t, starts_i32, win=96):
    """[clips, frames, 64] -> [clips, n_examples, win, 64] at the given start rows."""
    _dev_f32(logmel_t, "logmel")
    if not (starts_i32.is_cuda and starts_i32.dtype == torch.int32 and starts_i32.is_contiguous()):
        raise ValueError("starts: expected a contiguous int32 GPU tensor")
    clips, frames, _ = logmel_t.shape
    n = starts_i32.numel()
    out = torch.empty((clips, n, win, 64), device=logmel_t.device, dtype=torch.float32)
    check(_lib.load().cer_frame_examples(ptr(logmel_t), ptr(starts_i32), ptr(out), clips, frames, n, win,
                                         current_stream()), "cer_frame_examples")
    return out


def bert_embed_ln(ids, word, pos, typ, gamma, beta, eps=1e-12):
    if not (ids.is_cuda and ids.dtype == torch.int64 and ids.is_contiguous() and ids.dim() == 2):
        raise ValueError("ids: expected a contiguous [B, S] int64 GPU tensor")
    for t, n in ((word, "word"), (pos, "pos"), (typ, "type"), (gamma, "gamma"), (beta, "beta")):
        _dev_f32(t, n)
    b, s = ids.shape
    hd = word.shape[1]
    y = torch.empty((b, s, hd), device=ids.device, dtype=torch.float32)
    check(_lib.load().cer_bert_embed_ln(ptr(ids), ptr(word), ptr(pos), ptr(typ), ptr(gamma), ptr(beta), ptr(y), b, s, hd,
                                        word.shape[0], pos.shape[0], eps, current_stream()), "cer_bert_embed_ln")
    return y


# bench.py sets this to a list to time the MFMA attention launches: (kind, algorithmic FLOPs, start event, end event)
ATTN_TRACE = None


def attention(q, k, v, out, batch, heads, sq, sk, d, q_strides, k_strides, v_strides, o_strides, scale, key_mask=None,
              lse=None):
    """Strided attention: element (b, s, h, :) at base + b*st[0] + s*st[1] + h*st[2].  q/k/v/out are
    (views into) float32 GPU buffers; only their data pointers are used."""
    for t, n in ((q, "q"), (k, "k"), (v, "v"), (out, "out")):
        if not (t.is_cuda and t.dtype == torch.float32):
            raise ValueError(f"{n}: expected a float32 GPU tensor")
    if key_mask is not None and not (key_mask.is_cuda and key_mask.dtype == torch.int32 and key_mask.is_contiguous()):
        raise ValueError("key_mask: expected a contiguous int32 GPU tensor [B, Sk]")
    LL3 = ctypes.c_longlong * 3
    if ATTN_TRACE is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    check(_lib.load().cer_attention_fwd(ptr(q), ptr(k), ptr(v), ptr(key_mask), ptr(out), ptr(lse), batch, heads, sq, sk, d,
                                        LL3(*q_strides), LL3(*k_strides), LL3(*v_strides), LL3(*o_strides), scale,
                                        current_stream()), "cer_attention_fwd")
    if ATTN_TRACE is not None:
        e1.record()
        ATTN_TRACE.append(("fwd" if sq * sk < (1 << 18) else f"fwd_{sq}x{sk}", 4.0 * batch * heads * sq * sk * d, e0, e1))
    return out


def attention_bwd(q, k, v, out, dout, lse, dq, dk, dv, batch, heads, sq, sk, d, q_strides, k_strides, v_strides,
                  o_strides, do_strides, dq_strides, dk_strides, dv_strides, scale, key_mask=None):
    """Gradients of ``attention`` written into dq/dk/dv (views allowed, strides given explicitly)."""
    for t, n in ((q, "q"), (k, "k"), (v, "v"), (out, "out"), (dout, "dout"), (lse, "lse"), (dq, "dq"), (dk, "dk"), (dv, "dv")):
        if not (t.is_cuda and t.dtype == torch.float32):
            raise ValueError(f"{n}: expected a float32 GPU tensor")
    LL3 = ctypes.c_longlong * 3
    delta = torch.empty_like(lse)
    if ATTN_TRACE is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    check(_lib.load().cer_attention_bwd(ptr(q), ptr(k), ptr(v), ptr(out), ptr(dout), ptr(lse), ptr(key_mask), ptr(delta),
                                        ptr(dq), ptr(dk), ptr(dv), batch, heads, sq, sk, d, LL3(*q_strides),
                                        LL3(*k_strides), LL3(*v_strides), LL3(*o_strides), LL3(*do_strides),
                                        LL3(*dq_strides), LL3(*dk_strides), LL3(*dv_strides), scale, current_stream()),
          "cer_attention_bwd")
    if ATTN_TRACE is not None:
        e1.record()
        ATTN_TRACE.append(("bwd" if sq * sk < (1 << 18) else f"bwd_{sq}x{sk}", 8.0 * batch * heads * sq * sk * d, e0, e1))


def add_inplace(y, x):
    _dev_f32(y, "y")
    _dev_f32(x, "x")
    check(_lib.load().cer_add_inplace(ptr(y), ptr(x), y.numel(), current_stream()), "cer_add_inplace")
    return y
