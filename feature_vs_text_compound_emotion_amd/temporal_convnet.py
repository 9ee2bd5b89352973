"""Weight-normed causal dilated TCN on the HIP kernels (forward + hand-written backward).

Mirror of the reference's ``TemporalConvNet`` / ``TemporalBlock``
(models/temporal_convolutional_model.py:21-75): same constructor, same
state-dict keys (including the duplicated ``net.0`` / ``net.4`` entries that
alias ``conv1`` / ``conv2``).  Activations are channels-last rows [B*L, C]:

  h1  = m1 * leaky(conv_k,d(x) + b1)              one implicit-GEMM launch (KH=k, dil=d, left pad)
  out = leaky(m2 * leaky(conv_k,d(h1) + b2) + res)    one launch, residual + both activations fused
  res = x  or  conv1x1(x)

The reference pads both sides and chomps the tail; the kernel pads left only, so
none of the (k-1)*d discarded outputs is ever computed.
"""
import torch
from torch import nn

from . import ops


class _WNConv1d(nn.Module):
    """Parameter holder with weight_norm's key layout: bias, weight_g, weight_v."""

    def __init__(self, cin, cout, k):
        super().__init__()
        conv = nn.Conv1d(cin, cout, k)  # default init, exactly what survives in the reference
        v = conv.weight.detach().clone()
        self.bias = nn.Parameter(conv.bias.detach().clone())
        self.weight_g = nn.Parameter(v.reshape(cout, -1).norm(dim=1).view(cout, 1, 1))
        self.weight_v = nn.Parameter(v)


class _NoParam(nn.Module):
    pass


class TemporalBlock(nn.Module):
    def __init__(self, n_inputs, n_outputs, kernel_size, stride, dilation, padding, dropout=0.2):
        super().__init__()
        if stride != 1 or padding != (kernel_size - 1) * dilation:
            raise NotImplementedError("only the causal stride-1 configuration of the reference is supported")
        self.cin, self.cout, self.k, self.dilation, self.p = n_inputs, n_outputs, kernel_size, dilation, dropout
        self.conv1 = _WNConv1d(n_inputs, n_outputs, kernel_size)
        self.chomp1, self.relu1, self.dropout1 = _NoParam(), _NoParam(), _NoParam()
        self.conv2 = _WNConv1d(n_outputs, n_outputs, kernel_size)
        self.chomp2, self.relu2, self.dropout2 = _NoParam(), _NoParam(), _NoParam()
        self.net = nn.Sequential(self.conv1, self.chomp1, self.relu1, self.dropout1,
                                 self.conv2, self.chomp2, self.relu2, self.dropout2)
        self.downsample = nn.Conv1d(n_inputs, n_outputs, 1) if n_inputs != n_outputs else None
        self.relu = _NoParam()
        if self.downsample is not None:
            nn.init.xavier_uniform_(self.downsample.weight, gain=2 ** 0.5)

    def params(self):
        ps = [self.conv1.bias, self.conv1.weight_g, self.conv1.weight_v,
              self.conv2.bias, self.conv2.weight_g, self.conv2.weight_v]
        if self.downsample is not None:
            ps += [self.downsample.weight, self.downsample.bias]
        return ps


def _conv_rows(x_rows, w_oihw3, bsz, length, k, dil, packed=None, **kw):
    """Causal conv over rows [B*L, Cin] with a [Cout,Cin,k] filter (``packed``: its packed form, if the caller has it)."""
    cout, cin, _ = w_oihw3.shape
    wp = packed if packed is not None else ops.pack_conv_weight(w_oihw3.view(cout, cin, k, 1))
    y = ops.conv2d(x_rows.view(bsz, length, 1, cin), wp, k, 1, dil=(dil, 1), pad=((k - 1) * dil, 0),
                   out_hw=(length, 1), split_k=ops.auto_split_k(bsz * length, cout, cin * k), **kw)
    return y.view(bsz * length, cout)


def _dgrad_rows(dz_rows, w_oihw3, bsz, length, k, dil, residual=None, packed=None):
    """dX of the causal conv: an anti-causal conv of dZ with the flipped, transposed filter (``packed``: that filter packed)."""
    cout, cin, _ = w_oihw3.shape
    wt = packed if packed is not None else ops.pack_conv_weight(w_oihw3.view(cout, cin, k, 1), flip=True, transpose=True)
    res = residual.view(bsz, length, 1, cin) if residual is not None else None
    y = ops.conv2d(dz_rows.view(bsz, length, 1, cout), wt, k, 1, dil=(dil, 1), pad=(0, 0), out_hw=(length, 1),
                   residual=res, split_k=ops.auto_split_k(bsz * length, cin, cout * k))
    return y.view(bsz * length, cin)


class TCNFunction(torch.autograd.Function):
    """Whole TemporalConvNet as one autograd node.  args = (x_rows, bsz, length, cfg, masks, *params)
    with cfg = [(k, dil, has_downsample)] per level and masks = [(m1, m2)] or None."""

    @staticmethod
    def forward(ctx, x, bsz, length, cfg, masks, *params):
        saved, pi = [], 0
        x = x.contiguous()
        for lvl, (k, dil, has_ds) in enumerate(cfg):
            b1, g1, v1, b2, g2, v2 = params[pi:pi + 6]
            pi += 6
            m1, m2 = masks[lvl] if masks is not None else (None, None)
            # weight-norm + both packed layouts of a conv in ONE launch (the weights change every step: 4 pack launches per level)
            w1, n1, w1p, w1t = ops.weight_norm_fwd_packed(v1, g1)
            w2, n2, w2p, w2t = ops.weight_norm_fwd_packed(v2, g2)
            h1 = _conv_rows(x, w1, bsz, length, k, dil, packed=w1p, bias=b1, act1=ops.ACT_LEAKY, mask=m1)
            if has_ds:
                dsw, dsb = params[pi:pi + 2]
                pi += 2
                res = _conv_rows(x, dsw, bsz, length, 1, 1, bias=dsb)
            else:
                dsw, res = None, x
            a2 = torch.empty_like(h1)
            out = _conv_rows(h1, w2, bsz, length, k, dil, packed=w2p, bias=b2, act1=ops.ACT_LEAKY, mask=m2,
                             residual=res.view(bsz, length, 1, -1), act2=ops.ACT_LEAKY, aux=a2)
            saved.append((x, h1, a2, out, w1, n1, w2, n2, m1, m2, w1t, w2t))
            x = out
        ctx.cfg, ctx.bsz, ctx.length, ctx.saved_levels, ctx.params = cfg, bsz, length, saved, params
        return x

    @staticmethod
    def backward(ctx, dout):
        cfg, bsz, length, params = ctx.cfg, ctx.bsz, ctx.length, ctx.params
        grads = [None] * len(params)
        # walk levels in reverse; find each level's slice of the flat parameter list
        offsets, pi = [], 0
        for (_, _, has_ds) in cfg:
            offsets.append(pi)
            pi += 8 if has_ds else 6
        dout = dout.contiguous()
        for lvl in range(len(cfg) - 1, -1, -1):
            k, dil, has_ds = cfg[lvl]
            o = offsets[lvl]
            b1, g1, v1, b2, g2, v2 = params[o:o + 6]
            x, h1, a2, out, w1, n1, w2, n2, m1, m2, w1t, w2t = ctx.saved_levels[lvl]
            du, dz2 = ops.tblock_tail_bwd(dout, out, a2, m2)
            grads[o + 3] = ops.col_sum(dz2)
            # weight gradient -> (dv, dg): the weight-norm backward folds the split partial sums itself (no fold launch)
            dv2, dg2 = ops.conv1d_wgrad_weight_norm_bwd(dz2, h1, length, k, dil, v2, g2, n2)
            grads[o + 4], grads[o + 5] = dg2, dv2
            dh1 = _dgrad_rows(dz2, w2, bsz, length, k, dil, packed=w2t)
            dz1 = ops.act_mask_bwd(dh1, h1, m1)
            grads[o + 0] = ops.col_sum(dz1)
            dv1, dg1 = ops.conv1d_wgrad_weight_norm_bwd(dz1, x, length, k, dil, v1, g1, n1)
            grads[o + 1], grads[o + 2] = dg1, dv1
            if has_ds:
                dsw = params[o + 6]
                grads[o + 6] = ops.conv1d_wgrad(du, x, length, 1, 1)
                grads[o + 7] = ops.col_sum(du)
                dres = _dgrad_rows(du, dsw, bsz, length, 1, 1)
            else:
                dres = du
            need_dx = lvl > 0 or ctx.needs_input_grad[0]
            dout = _dgrad_rows(dz1, w1, bsz, length, k, dil, residual=dres, packed=w1t) if need_dx else None
        return (dout, None, None, None, None, *grads)


class TemporalConvNet(nn.Module):
    def __init__(self, num_inputs, num_channels, kernel_size=2, dropout=0.2, max_length=200, attention=0):
        super().__init__()
        if attention:
            raise NotImplementedError("the reference's AttentionBlock is dead code (attention=0 everywhere)")
        layers = []
        for i, cout in enumerate(num_channels):
            cin = num_inputs if i == 0 else num_channels[i - 1]
            layers.append(TemporalBlock(cin, cout, kernel_size, stride=1, dilation=2 ** i,
                                        padding=(kernel_size - 1) * 2 ** i, dropout=dropout))
        self.network = nn.Sequential(*layers)
        self.dropout = dropout

    def forward_rows(self, x_rows, bsz, length, masks=None, seed=None):
        """x_rows [B*L, Cin] channels-last -> [B*L, Cout].  In train mode dropout masks are drawn
        on the device from ``seed`` unless ``masks`` ([(m1, m2)] per level, pre-scaled) is given."""
        cfg = [(b.k, b.dilation, b.downsample is not None) for b in self.network]
        if self.training and masks is None and self.dropout > 0:
            if seed is None:
                seed = int(torch.randint(0, 2 ** 31 - 1, (1,)).item())
            # all 2 * levels masks from ONE launch over one flat buffer: every mask owns its own counter range of the
            # (seed, counter) generator -- per-level offsets (2i) * n_i overlapped when the channel count changes between
            # levels, which made some masks element-for-element copies of others -- and it is 1 launch instead of 8
            sizes = [bsz * length * b.cout for b in self.network for _ in (0, 1)]
            flat = ops.dropout_mask((sum(sizes),), self.dropout, seed, 0, x_rows.device)
            views, off = [], 0
            for n in sizes:
                views.append(flat[off:off + n])
                off += n
            masks = [(views[2 * i].view(bsz * length, b.cout), views[2 * i + 1].view(bsz * length, b.cout))
                     for i, b in enumerate(self.network)]
        if not self.training:
            masks = None
        params = [p for b in self.network for p in b.params()]
        return TCNFunction.apply(x_rows, bsz, length, cfg, masks, *params)

    def forward(self, x):
        """Reference layout: x [B, Cin, L] -> [B, Cout, L]."""
        bsz, cin, length = x.shape
        rows = x.transpose(1, 2).contiguous().view(bsz * length, cin)
        y = self.forward_rows(rows, bsz, length)
        return y.view(bsz, length, -1).transpose(1, 2)
