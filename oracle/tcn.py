"""Oracle: weight-normed causal dilated temporal conv net (test infrastructure only).

Follows /root/reference/models/temporal_convolutional_model.py:12-18 (Chomp1d),
:21-56 (TemporalBlock), :59-75 (TemporalConvNet).  The reference pads both sides
by (k-1)*d and chomps the tail; that equals a left-only (causal) pad, which is
what is written here.
"""
import torch
import torch.nn.functional as F

LEAKY_SLOPE = 0.01


def weight_norm_weight(g, v):
    """torch.nn.utils.weight_norm(dim=0): w = g * v / ||v|| per output channel."""
    norm = v.reshape(v.shape[0], -1).norm(dim=1).view(-1, *([1] * (v.dim() - 1)))
    return g * v / norm


def _causal_conv(x, w, b, dilation):
    k = w.shape[-1]
    x = F.pad(x, ((k - 1) * dilation, 0))
    return F.conv1d(x, w, b, dilation=dilation)


def tcn_num_levels(sd, prefix):
    n = 0
    while f"{prefix}network.{n}.conv1.weight_v" in sd:
        n += 1
    return n


def tcn_forward(x, sd, prefix, dropout_masks=None):
    """x: [B,Cin,L] -> [B,Cout,L].

    ``dropout_masks``: optional list (one per level) of (mask1, mask2) tensors,
    already scaled by 1/(1-p), shaped like the conv outputs; None = eval.
    """
    for i in range(tcn_num_levels(sd, prefix)):
        b = f"{prefix}network.{i}."
        d = 2 ** i
        w1 = weight_norm_weight(sd[b + "conv1.weight_g"], sd[b + "conv1.weight_v"])
        w2 = weight_norm_weight(sd[b + "conv2.weight_g"], sd[b + "conv2.weight_v"])
        h = F.leaky_relu(_causal_conv(x, w1, sd[b + "conv1.bias"], d), LEAKY_SLOPE)
        if dropout_masks is not None:
            h = h * dropout_masks[i][0]
        h = F.leaky_relu(_causal_conv(h, w2, sd[b + "conv2.bias"], d), LEAKY_SLOPE)
        if dropout_masks is not None:
            h = h * dropout_masks[i][1]
        if (b + "downsample.weight") in sd:
            res = F.conv1d(x, sd[b + "downsample.weight"], sd[b + "downsample.bias"])
        else:
            res = x
        x = F.leaky_relu(h + res, LEAKY_SLOPE)
    return x
