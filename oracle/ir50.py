"""Oracle: IR-ResNet50 vision encoder (test infrastructure only).

Follows /root/reference/models/arcface_model.py:17-20 (l2_norm), :44-60
(bottleneck_IR), :95-102 (stage table), :130-151 (Backbone) and
/root/reference/models/backbone.py:99-103 (the 5x5 output head that
VisualBackbone re-installs).  Functional form over a flat state dict whose keys
are the reference's (``<prefix>input_layer.0.weight`` ...).
"""
import torch
import torch.nn.functional as F

# (in_channel, depth, num_units, stride of first unit) -- arcface_model.py:96-102
IR50_STAGES = ((64, 64, 3, 1), (64, 128, 4, 2), (128, 256, 14, 2), (256, 512, 3, 2))
BN_EPS = 1e-5
BN_MOMENTUM = 0.1


def ir50_block_plan():
    """[(in_channel, depth, stride)] for the 24 units, in body order."""
    plan = []
    for cin, depth, n, stride in IR50_STAGES:
        plan.append((cin, depth, stride))
        plan.extend((depth, depth, 1) for _ in range(n - 1))
    return plan


def _bn(x, sd, key, train, new_buffers):
    """BatchNorm over dim 1.  Eval: running stats.  Train: batch stats, and the
    running-stat update torch performs (unbiased variance, momentum 0.1)."""
    w, b = sd[key + ".weight"], sd[key + ".bias"]
    rm, rv = sd[key + ".running_mean"], sd[key + ".running_var"]
    dims = [0] + list(range(2, x.dim()))
    shape = [1, -1] + [1] * (x.dim() - 2)
    if train:
        mean = x.mean(dims)
        var = x.var(dims, unbiased=False)
        if new_buffers is not None:
            n = x.numel() // x.shape[1]
            new_buffers[key + ".running_mean"] = (1 - BN_MOMENTUM) * rm + BN_MOMENTUM * mean
            new_buffers[key + ".running_var"] = (1 - BN_MOMENTUM) * rv + BN_MOMENTUM * var * n / max(n - 1, 1)
    else:
        mean, var = rm, rv
    inv = torch.rsqrt(var + BN_EPS)
    return (x - mean.view(shape)) * (inv * w).view(shape) + b.view(shape)


def _prelu(x, a):
    return torch.where(x >= 0, x, x * a.view(1, -1, 1, 1))


def ir50_forward(x, sd, prefix="", train=False, head_dropout_mask=None,
                 new_buffers=None, return_features=False):
    """x: [N,3,H,W] float32 (NCHW like the reference) -> [N,512] unit-norm rows.

    ``head_dropout_mask`` ([N,512,h,w], already scaled by 1/(1-p)) stands in for
    Dropout(0.4) in train mode; None means identity.
    """
    p = prefix
    x = F.conv2d(x, sd[p + "input_layer.0.weight"], None, 1, 1)
    x = _bn(x, sd, p + "input_layer.1", train, new_buffers)
    x = _prelu(x, sd[p + "input_layer.2.weight"])
    for i, (cin, depth, stride) in enumerate(ir50_block_plan()):
        b = f"{p}body.{i}."
        if cin == depth:
            shortcut = x[:, :, ::stride, ::stride]  # MaxPool2d(1, stride)
        else:
            shortcut = F.conv2d(x, sd[b + "shortcut_layer.0.weight"], None, stride, 0)
            shortcut = _bn(shortcut, sd, b + "shortcut_layer.1", train, new_buffers)
        r = _bn(x, sd, b + "res_layer.0", train, new_buffers)
        r = F.conv2d(r, sd[b + "res_layer.1.weight"], None, 1, 1)
        r = _prelu(r, sd[b + "res_layer.2.weight"])
        r = F.conv2d(r, sd[b + "res_layer.3.weight"], None, stride, 1)
        r = _bn(r, sd, b + "res_layer.4", train, new_buffers)
        x = r + shortcut
    feat = x
    x = _bn(x, sd, p + "output_layer.0", train, new_buffers)
    if head_dropout_mask is not None:
        x = x * head_dropout_mask
    x = x.reshape(x.shape[0], -1)  # channel-major flatten (c, h, w)
    x = F.linear(x, sd[p + "output_layer.3.weight"], sd[p + "output_layer.3.bias"])
    x = _bn(x, sd, p + "output_layer.4", train, new_buffers)
    x = x / torch.norm(x, 2, 1, True)  # no epsilon (arcface_model.py:17-20)
    if return_features:
        return x, feat
    return x
