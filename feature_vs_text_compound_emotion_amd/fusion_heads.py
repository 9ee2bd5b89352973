"""CAN and JMT / MT fusion models on the HIP kernels (forward + hand-written backward).

Mirrors of the reference's ``CAN`` (models/model.py:571-684), ``AttentionFusion`` (:529-568),
``TransformerEncoderLayer/Block`` (:716-750), ``JMTFusion`` (:895-979), ``MTFusion`` (:982-1048) and
``JMT`` (:1051-1167): same constructors, same state-dict keys, same ``forward(dict)``.

Everything stays in the b-major row layout [B*L, C] that the TCNs produce.  The reference permutes to
sequence-first [L, B, C] for ``nn.MultiheadAttention``; here the attention kernel takes explicit
(batch, token) strides instead, so no tensor is ever permuted.  The reference's final stage views
the stacked cross-attention outputs as [L*B, n_stack, 128] and attends over all L*B (frame, clip)
tokens per stack slot (clips of a batch are mixed); attention is permutation
equivariant over tokens, so running it over the b-major token order gives the same rows.

Each compute step is a small ``autograd.Function`` over C-ABI calls (GEMMs on the fp32 matrix
cores incl. all weight/data gradients, flash attention forward/backward, LayerNorm, BatchNorm);
torch only routes gradients between them.
"""
from os.path import join

import torch
from torch import nn

from . import ops
from .lfan import CLASSIFICATION, REGRESSION, TASKS, _linear_T, _packed  # noqa: F401
from .temporal_convnet import TemporalConvNet
from .visual_backbone import VisualBackbone

E = 128
BN_EPS, BN_MOMENTUM, LN_EPS = 1e-5, 0.1, 1e-5


class LinearFn(torch.autograd.Function):
    """y = act(x W^T + b) + residual   (act in {none, relu}); x may be a column slice."""

    @staticmethod
    def forward(ctx, x, w, b, act, residual):
        y = ops.linear(x, _packed(w), bias=b, act=act, residual=residual)
        ctx.act, ctx.has_res = act, residual is not None
        ctx.saved = (x, w, y if act != ops.ACT_NONE else None)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w, y = ctx.saved
        dy = dy.contiguous()
        dres = dy if ctx.has_res else None
        if ctx.act != ops.ACT_NONE:
            if ctx.has_res:
                raise RuntimeError("LinearFn: activation and residual together are not supported in the backward")
            dy = ops.act_mask_bwd(dy, y, None, 0.0)  # ReLU
        dx = _linear_T(dy, w) if ctx.needs_input_grad[0] else None
        dw = ops.conv1d_wgrad(dy, x, dy.shape[0], 1, 1).view_as(w)
        return dx, dw, ops.col_sum(dy), None, dres


def linear(x, lin, act=ops.ACT_NONE, residual=None):
    return LinearFn.apply(x, lin.weight, lin.bias, act, residual)


class LayerNormFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, gamma, beta):
        x = x.contiguous()
        y, mean, rstd = ops.layernorm_fwd(x, gamma, beta, eps=LN_EPS)
        ctx.saved = (x, gamma, mean, rstd)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, gamma, mean, rstd = ctx.saved
        return ops.layernorm_bwd(dy.contiguous(), x, gamma, mean, rstd)


class BNRowsFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, b, running_mean, running_var, train):
        x = x.contiguous()
        y, sm, si = ops.bn_rows_fwd(x, w, b, running_mean, running_var, train, BN_EPS, BN_MOMENTUM)
        if not train:
            sm, si = running_mean, torch.rsqrt(running_var + BN_EPS)
        ctx.saved, ctx.train = (x, w, sm, si), train
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w, sm, si = ctx.saved
        dx, dw, db = ops.bn_rows_bwd(dy.contiguous(), x, sm, si, w, ctx.train)
        return dx, dw, db, None, None, None


def batchnorm_rows(x, bn, train):
    y = BNRowsFn.apply(x, bn.weight, bn.bias, bn.running_mean, bn.running_var, train)
    if train:
        bn.num_batches_tracked += 1
    return y


class LeakyReLUFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        y = ops.leaky_relu(x.contiguous())
        ctx.y = y
        return y

    @staticmethod
    def backward(ctx, dy):
        return ops.act_mask_bwd(dy.contiguous(), ctx.y, None)


class SoftmaxGateFn(torch.autograd.Function):
    """out = softmax(z) * c (CAN's AttentionFusion gate)."""

    @staticmethod
    def forward(ctx, z, c):
        out, prob = ops.softmax_gate_fwd(z.contiguous(), c.contiguous())
        ctx.saved = (prob, c)
        return out

    @staticmethod
    def backward(ctx, dout):
        prob, c = ctx.saved
        return ops.softmax_gate_bwd(dout.contiguous(), prob, c)


class MHAFn(torch.autograd.Function):
    """nn.MultiheadAttention(128, 1 head) on row tensors.

    xq [Rq,E], xkv [Rk,E]; token (b, s) of the query side is row b*geo.q_sb + s*geo.q_ss (same for
    the key side).  geo = (B, Sq, Sk, q_sb, q_ss, k_sb, k_ss) in ROWS.  residual (optional) is added
    in the out-projection epilogue (the encoder layer's ``x + attn_output``).
    """

    @staticmethod
    def forward(ctx, xq, xkv, in_w, in_b, out_w, out_b, geo, residual, self_attn):
        B, Sq, Sk, q_sb, q_ss, k_sb, k_ss = geo
        xq, xkv = xq.contiguous(), xkv.contiguous()
        scale = 1.0 / E ** 0.5
        if self_attn:
            qkv = ops.linear(xq, in_w, bias=in_b)  # [R, 3E]
            q, k, v, pitch_q, pitch_k = qkv, qkv[:, E:], qkv[:, 2 * E:], 3 * E, 3 * E
        else:
            q = ops.linear(xq, in_w[:E], bias=in_b[:E])
            kv = ops.linear(xkv, in_w[E:], bias=in_b[E:])  # [Rk, 2E]
            k, v, pitch_q, pitch_k = kv, kv[:, E:], E, 2 * E
        ctxt = torch.empty((xq.shape[0], E), device=xq.device, dtype=torch.float32)
        lse = torch.empty((B, 1, Sq), device=xq.device, dtype=torch.float32)
        qs = (q_sb * pitch_q, q_ss * pitch_q, 0)
        ks = (k_sb * pitch_k, k_ss * pitch_k, 0)
        os_ = (q_sb * E, q_ss * E, 0)
        ops.attention(q, k, v, ctxt, B, 1, Sq, Sk, E, qs, ks, ks, os_, scale, lse=lse)
        y = ops.linear(ctxt, out_w, bias=out_b, residual=residual)
        ctx.geo, ctx.self_attn, ctx.has_res = geo, self_attn, residual is not None
        ctx.saved = (xq, xkv, in_w, out_w, q, k, v, ctxt, lse, qs, ks, os_)
        return y

    @staticmethod
    def backward(ctx, dy):
        B, Sq, Sk, q_sb, q_ss, k_sb, k_ss = ctx.geo
        xq, xkv, in_w, out_w, q, k, v, ctxt, lse, qs, ks, os_ = ctx.saved
        dy = dy.contiguous()
        scale = 1.0 / E ** 0.5
        d_out_w = ops.conv1d_wgrad(dy, ctxt, dy.shape[0], 1, 1).view_as(out_w)
        d_out_b = ops.col_sum(dy)
        dctx = _linear_T(dy, out_w)
        if ctx.self_attn:
            dqkv = torch.empty((xq.shape[0], 3 * E), device=dy.device, dtype=torch.float32)
            ops.attention_bwd(q, k, v, ctxt, dctx, lse, dqkv, dqkv[:, E:], dqkv[:, 2 * E:], B, 1, Sq, Sk, E, qs, ks, ks,
                              os_, os_, qs, ks, ks, scale)
            d_in_w = ops.conv1d_wgrad(dqkv, xq, dqkv.shape[0], 1, 1).view_as(in_w)
            d_in_b = ops.col_sum(dqkv)
            dxq, dxkv = _linear_T(dqkv, in_w), None
        else:
            dq = torch.empty((xq.shape[0], E), device=dy.device, dtype=torch.float32)
            dkv = torch.empty((xkv.shape[0], 2 * E), device=dy.device, dtype=torch.float32)
            ops.attention_bwd(q, k, v, ctxt, dctx, lse, dq, dkv, dkv[:, E:], B, 1, Sq, Sk, E, qs, ks, ks, os_, os_, os_,
                              ks, ks, scale)
            d_in_w = torch.cat([ops.conv1d_wgrad(dq, xq, dq.shape[0], 1, 1).view(E, E),
                                ops.conv1d_wgrad(dkv, xkv, dkv.shape[0], 1, 1).view(2 * E, E)], 0)
            d_in_b = torch.cat([ops.col_sum(dq), ops.col_sum(dkv)])
            dxq = _linear_T(dq, in_w[:E])
            dxkv = _linear_T(dkv, in_w[E:])
        return dxq, dxkv, d_in_w, d_in_b, d_out_w, d_out_b, None, (dy if ctx.has_res else None), None


def mha(xq, xkv, m, geo, residual=None):
    """m: nn.MultiheadAttention parameter holder.  Self-attention when xkv is xq."""
    self_attn = xkv is xq
    return MHAFn.apply(xq, xkv, m.in_proj_weight, m.in_proj_bias, m.out_proj.weight, m.out_proj.bias, geo, residual,
                       self_attn)


class TransformerEncoderLayer(nn.Module):
    def __init__(self, input_dim, num_heads, hidden_dim):
        super().__init__()
        if input_dim != E or num_heads != 1:
            raise NotImplementedError("the reference only builds 128-d single-head encoder layers")
        self.attention = nn.MultiheadAttention(input_dim, num_heads)  # parameter holder (never called)
        self.feed_forward = nn.Sequential(nn.Linear(input_dim, hidden_dim), nn.ReLU(), nn.Linear(hidden_dim, input_dim))
        self.layer_norm1 = nn.LayerNorm(input_dim)
        self.layer_norm2 = nn.LayerNorm(input_dim)

    def forward_rows(self, x, geo):
        x = LayerNormFn.apply(mha(x, x, self.attention, geo, residual=x), self.layer_norm1.weight, self.layer_norm1.bias)
        h = linear(x, self.feed_forward[0], act=ops.ACT_RELU)
        y = linear(h, self.feed_forward[2], residual=x)
        return LayerNormFn.apply(y, self.layer_norm2.weight, self.layer_norm2.bias)


class TransformerEncoderBlock(nn.Module):
    def __init__(self, input_dim, num_heads, hidden_dim, num_layers):
        super().__init__()
        self.layers = nn.Sequential(*[TransformerEncoderLayer(input_dim, num_heads, hidden_dim) for _ in range(num_layers)])

    def forward_rows(self, x, geo):
        for layer in self.layers:
            x = layer.forward_rows(x, geo)
        return x


class _JMTFusionBase(nn.Module):
    joint = True

    def __init__(self, num_feats_modality=None, num_out_feats=256):
        super().__init__()
        self.visual_encoder = TransformerEncoderBlock(E, 1, E, 1)
        self.audio_encoder = TransformerEncoderBlock(E, 1, E, 1)
        if self.joint:
            self.jr_encoder = TransformerEncoderBlock(E, 1, E, 1)
        # the reference assigns final_encoder here first (and re-assigns it below), which fixes its
        # position in the parameter order
        self.final_encoder = TransformerEncoderBlock(E, 1, E, 1)
        self.CA_va = nn.MultiheadAttention(E, 1)
        self.CA_av = nn.MultiheadAttention(E, 1)
        if self.joint:
            self.CA_jra = nn.MultiheadAttention(E, 1)
            self.CA_ajr = nn.MultiheadAttention(E, 1)
            self.CA_vjr = nn.MultiheadAttention(E, 1)
            self.CA_jrv = nn.MultiheadAttention(E, 1)
        self.reduce_feats_dim = nn.Linear(2 * E, E)
        self.augment_audio_feats_dim = nn.Linear(64, E)
        self.final_self_attention = nn.MultiheadAttention(E, 1)

    def forward_rows(self, v, a, bsz, length):
        """v [B*L,128], a [B*L,64] (b-major rows) -> [B*L,128]."""
        geo = (bsz, length, length, length, 1, length, 1)  # attention over the L frames of each clip
        a = linear(a, self.augment_audio_feats_dim)
        ev = self.visual_encoder.forward_rows(v, geo)
        ea = self.audio_encoder.forward_rows(a, geo)
        if self.joint:
            jr = linear(torch.cat((v, a), dim=1), self.reduce_feats_dim)
            ej = self.jr_encoder.forward_rows(jr, geo)
            feats = [mha(ev, ea, self.CA_va, geo), mha(ea, ev, self.CA_av, geo), mha(ej, ev, self.CA_jrv, geo),
                     mha(ev, ej, self.CA_vjr, geo), mha(ej, ea, self.CA_jra, geo), mha(ea, ej, self.CA_ajr, geo)]
        else:
            feats = [mha(ev, ea, self.CA_va, geo), mha(ea, ev, self.CA_av, geo)]
        n = len(feats)
        rows = bsz * length
        st = torch.stack(feats, dim=1).reshape(rows * n, E)  # row = token*n + slot
        geo_f = (n, rows, rows, 1, n, 1, n)                   # "batch" = stack slot, sequence = all L*B tokens
        st = self.final_encoder.forward_rows(st, geo_f)
        out = mha(st, st, self.final_self_attention, geo_f)
        return out.view(rows, n, E)[:, -1, :]


class JMTFusion(_JMTFusionBase):
    joint = True


class MTFusion(_JMTFusionBase):
    joint = False


def _load_visual(root_dir, backbone_settings, head_hw, load_backbone):
    resnet = VisualBackbone(mode="ir", use_pretrained=False, head_hw=head_hw)
    if load_backbone:
        sd = torch.load(join(root_dir, backbone_settings["visual_state_dict"] + ".pth"), map_location="cpu",
                        weights_only=True)
        resnet.load_state_dict(sd)
    for p in resnet.parameters():
        p.requires_grad = False
    return resnet


class _TailModel(nn.Module):
    """Shared front: IR-50 on frames, one TCN + BatchNorm1d per modality (rows layout)."""

    def _build_front(self, modalities, tcn_settings, backbone_settings, root_dir, head_hw, load_backbone):
        self.modalities = list(modalities)
        self.temporal, self.bn, self.spatial = nn.ModuleDict(), nn.ModuleDict(), nn.ModuleDict()
        for m in modalities:
            self.temporal[m] = TemporalConvNet(num_inputs=tcn_settings[m]["input_dim"],
                                               num_channels=tcn_settings[m]["channel"],
                                               kernel_size=tcn_settings[m]["kernel_size"])
            self.bn[m] = nn.BatchNorm1d(tcn_settings[m]["channel"][-1])
        self._front_args = (backbone_settings, root_dir, head_hw, load_backbone)

    def _attach_visual(self):
        backbone_settings, root_dir, head_hw, load_backbone = self._front_args
        if "video" in self.modalities:
            self.spatial["visual"] = _load_visual(root_dir, backbone_settings, head_hw, load_backbone)
        if "logmel" in self.modalities:   # model.py:626-638 / 1120-1132 (dead in main.py; the on-model VGGish)
            from .audio_backbone import AudioBackbone
            audio = AudioBackbone()
            if load_backbone:
                audio.backbone.load_state_dict(torch.load(join(root_dir, backbone_settings["audio_state_dict"] + ".pth"),
                                                          map_location="cpu", weights_only=True))
            self.spatial["audio"] = audio

    def _front(self, X):
        # model.py:651-672 / 1134-1155: the caller's dict is walked in its own key order (any order works: the fusion picks
        # the modalities by name) and 'video' / 'logmel' are overwritten in place with their embeddings [B, 1, L, C]
        for m in X:
            if m not in self.temporal:
                raise KeyError(m)
        for m in self.modalities:
            if m not in X:
                raise KeyError(m)
        if self.training:
            self.dropout_seed += 1
        feats, bsz, length = {}, None, None
        for m in X:
            i = self.modalities.index(m)               # dropout streams are tied to the model's modality order
            x = X[m]
            if m == "video":
                bsz, length = x.shape[0], x.shape[1]
                # frozen encoder (the reference's default): no autograd graph.  After a gradual release
                # (base/parameter_control.py) the released encoder parameters need the gradient of the embedding.
                released = any(p.requires_grad for p in self.spatial["visual"].parameters())
                with torch.set_grad_enabled(released and torch.is_grad_enabled()):
                    vis = self.spatial["visual"]
                    vis.backbone.dropout_seed = self.dropout_seed
                    rows = vis(x.reshape(-1, *x.shape[2:]))
                X[m] = rows.detach().view(bsz, length, -1).unsqueeze(1)
            elif m == "logmel":
                bsz, height, length, width = x.shape
                with torch.no_grad():
                    rows = self.spatial["audio"](x.permute(0, 2, 3, 1).contiguous().view(-1, width, height))
                X[m] = rows.view(bsz, length, -1).unsqueeze(1)
            else:
                bsz, length = x.shape[0], x.shape[2]
                rows = x.reshape(bsz * length, x.shape[-1])
            t = self.temporal[m].forward_rows(rows, bsz, length, seed=self.dropout_seed * 16 + i)
            feats[m] = batchnorm_rows(t, self.bn[m], self.training)
        return feats, bsz, length

    def _head(self, c, bsz, length):
        c = linear(c, self.fc1)
        c = batchnorm_rows(c, self.bn1, self.training)
        c = linear(LeakyReLUFn.apply(c), self.fc2)
        c = c.view(bsz, length, -1)
        return torch.tanh(c) if self.task == REGRESSION else c


class JMT(_TailModel):
    def __init__(self, task, modalities, tcn_settings, backbone_settings, output_dim, root_dir, device, model_name,
                 head_hw=5, load_backbone=True):
        super().__init__()
        assert task in TASKS, task
        self.device, self.task, self.dropout_seed = device, task, 0
        if list(modalities) != ["video", "vggish"] and set(modalities) != {"video", "vggish"}:
            raise ValueError("JMT / MT fuse exactly the 'video' and 'vggish' modalities (models/model.py:940-941)")
        self._build_front(modalities, tcn_settings, backbone_settings, root_dir, head_hw, load_backbone)
        if model_name == "JMT":
            self.fuse = JMTFusion()
        elif model_name == "MT":
            self.fuse = MTFusion()
        else:
            raise NotImplementedError(model_name)
        self.bn1 = nn.BatchNorm1d(E)
        self.fc1 = nn.Linear(E, E)
        self.fc2 = nn.Linear(E, output_dim)
        self._attach_visual()

    def forward(self, X):
        feats, bsz, length = self._front(X)
        c = self.fuse.forward_rows(feats["video"], feats["vggish"], bsz, length)
        return self._head(c, bsz, length)


class AttentionFusion(nn.Module):
    def __init__(self, num_feats_modality, num_out_feats=256):
        super().__init__()
        self.attn = nn.ModuleList([nn.Linear(n, num_out_feats) for n in num_feats_modality])
        d = num_out_feats * len(num_feats_modality)
        self.weights = nn.Linear(d, d)
        self.num_features = d

    def forward_rows(self, feats):
        cat = torch.cat([linear(x, self.attn[i]) for i, x in enumerate(feats)], dim=1)
        return SoftmaxGateFn.apply(linear(cat, self.weights), cat)


class CAN(_TailModel):
    def __init__(self, task, modalities, tcn_settings, backbone_settings, output_dim, root_dir, device, head_hw=5,
                 load_backbone=True):
        super().__init__()
        assert task in TASKS, task
        self.device, self.task, self.dropout_seed = device, task, 0
        self.up_sample = nn.ModuleDict()
        self._build_front(modalities, tcn_settings, backbone_settings, root_dir, head_hw, load_backbone)
        m = len(modalities)
        self.fuse = AttentionFusion([tcn_settings[mod]["channel"][-1] for mod in modalities], num_out_feats=E)
        self.conv_c = nn.Conv1d(E * m, E, 1)  # defined by the reference, never used by its forward
        self.bn1 = nn.BatchNorm1d(E * m)
        self.fc1 = nn.Linear(E * m, E * m)
        self.fc2 = nn.Linear(E * m, output_dim)
        self._attach_visual()

    def forward(self, X):
        feats, bsz, length = self._front(X)
        # model.py:558-560 pairs attn[i] with the i-th VALUE of the caller's dict: the caller's key order is part of the
        # contract here (another order meets other projection weights, or a shape error), exactly as in the reference
        c = self.fuse.forward_rows(list(feats.values()))
        return self._head(c, bsz, length)
