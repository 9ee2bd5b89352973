"""Oracle: what 16-bit STORAGE alone does to the IR-50 embedding (test infrastructure only).

Not a restatement of the reference -- the reference's narrow arithmetic is torch's autocast
(/root/reference/trainer.py:341,367), which ``autocast_lfan_forward`` below runs through the fp32 oracle -- but
of the storage rule the HIP narrow modes follow (``visual_backbone._forward_batch_stats_n16``): every activation
tensor and every weight matrix is rounded ONCE to the storage type, everything else (accumulation, batch
statistics taken before the rounding, BatchNorm arithmetic, residual adds) is fp32.  It exists to give the
narrow-mode parity bars a stated origin: the error of this emulation against the fp32 oracle is the floor of
the storage format on a given batch, independent of any kernel.
"""
import torch
import torch.nn.functional as F

from .ir50 import BN_EPS, ir50_block_plan
from .lfan import lfan_forward


def ir50_forward_narrow_storage(x, sd, prefix, dtype, train=True, head_dropout_mask=None, round_residual_stream=True):
    """x [N,3,H,W] -> [N,512]; ``dtype`` torch.bfloat16 / torch.float16; batch statistics (``train``) only."""
    if not train:
        raise NotImplementedError("the emulation covers the batch-statistics forward (the mode cfg5 and --amp time)")
    p = prefix

    def r(t):
        return t.to(dtype).float()

    def affine(v, key):
        dims = [0] + list(range(2, v.dim()))
        s = sd[key + ".weight"] * torch.rsqrt(v.var(dims, unbiased=False) + BN_EPS)
        return s, sd[key + ".bias"] - v.mean(dims) * s

    def c(v):
        return v.view(1, -1, 1, 1)
    z = F.conv2d(x, sd[p + "input_layer.0.weight"], None, 1, 1)          # the input layer runs on fp32 operands
    s, t = affine(z, p + "input_layer.1")
    y32 = z * c(s) + c(t)
    y32 = torch.where(y32 >= 0, y32, y32 * c(sd[p + "input_layer.2.weight"]))
    ys = r(y32)
    for i, (cin, depth, stride) in enumerate(ir50_block_plan()):
        b = f"{p}body.{i}."
        s1, t1 = affine(y32, b + "res_layer.0")                            # statistics of the fp32 values, before the store
        w1 = sd[b + "res_layer.1.weight"]
        # conv(BN(x)) with zero padding == conv_{w*s1}(x) + (conv of the constant shift image, border dependent)
        b9 = F.conv2d(c(t1).expand(1, cin, ys.shape[2], ys.shape[3]).contiguous(), w1, None, 1, 1)
        tt = F.conv2d(ys, r(w1 * c(s1)), None, 1, 1) + b9
        tt = r(torch.where(tt >= 0, tt, tt * c(sd[b + "res_layer.2.weight"])))
        z32 = F.conv2d(tt, r(sd[b + "res_layer.3.weight"]), None, stride, 1)
        s2, t2 = affine(z32, b + "res_layer.4")
        if cin != depth:
            sc32 = F.conv2d(ys, r(sd[b + "shortcut_layer.0.weight"]), None, stride, 0)
            ss, st = affine(sc32, b + "shortcut_layer.1")
            res = r(sc32) * c(ss) + c(st)
        else:
            res = ys[:, :, ::stride, ::stride]
        y32 = r(z32) * c(s2) + c(t2) + res
        ys = r(y32) if round_residual_stream else y32
    s0, t0 = affine(y32, p + "output_layer.0")
    h = y32 * c(s0) + c(t0)
    if head_dropout_mask is not None:
        h = h * head_dropout_mask
    h = r(h)
    e = F.linear(h.reshape(h.shape[0], -1), r(sd[p + "output_layer.3.weight"]), sd[p + "output_layer.3.bias"])
    e = (e - e.mean(0)) * torch.rsqrt(e.var(0, unbiased=False) + BN_EPS) * sd[p + "output_layer.4.weight"] \
        + sd[p + "output_layer.4.bias"]
    return e / e.norm(dim=1, keepdim=True)


def autocast_lfan_forward(inputs, sd, modalities, dtype, **kw):
    """The reference's own narrow arithmetic: ``with autocast(enabled=args.amp): model(inputs)``
    (/root/reference/trainer.py:367) -- the fp32 oracle under ``torch.autocast("cpu", dtype)``; returns fp32 logits."""
    with torch.autocast("cpu", dtype=dtype):
        out = lfan_forward(inputs, sd, modalities, **kw)
    return out.float()


def autocast_ir50_forward(x, sd, prefix, dtype, **kw):
    from .ir50 import ir50_forward
    with torch.autocast("cpu", dtype=dtype):
        out = ir50_forward(x, sd, prefix, **kw)
    return out.float()
