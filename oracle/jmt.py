"""Oracle: CAN and JMT / MT fusion heads (test infrastructure only).

Follows /root/reference/models/model.py:529-568 (AttentionFusion), :571-684 (CAN),
:716-750 (TransformerEncoderLayer/Block), :895-979 (JMTFusion), :982-1048 (MTFusion),
:1051-1167 (JMT).  nn.MultiheadAttention(128, 1) is restated explicitly (single head, seq-first
tensors, scale 1/sqrt(128), no dropout).  Note the reference's final stage views the stacked
cross-attention outputs as ``[L*B, n_stack, 128]`` and runs seq-first attention on it, i.e. it
attends over ALL L*B (frame, clip) tokens for each stack slot -- clips of a batch are mixed
(SURVEY.md F7); the oracle reproduces that.
"""
import math

import torch
import torch.nn.functional as F

from .ir50 import ir50_forward
from .lfan import _bn1d
from .tcn import tcn_forward

E = 128


def mha(xq, xkv, sd, p):
    """xq [Sq,B,E], xkv [Sk,B,E] -> [Sq,B,E] (1 head)."""
    w, b = sd[p + "in_proj_weight"], sd[p + "in_proj_bias"]
    q = F.linear(xq, w[:E], b[:E])
    k = F.linear(xkv, w[E:2 * E], b[E:2 * E])
    v = F.linear(xkv, w[2 * E:], b[2 * E:])
    att = torch.softmax(torch.einsum("sbe,tbe->bst", q, k) / math.sqrt(E), dim=-1)
    ctx = torch.einsum("bst,tbe->sbe", att, v)
    return F.linear(ctx, sd[p + "out_proj.weight"], sd[p + "out_proj.bias"])


def encoder_layer(x, sd, p):
    x = F.layer_norm(x + mha(x, x, sd, p + "attention."), (E,), sd[p + "layer_norm1.weight"], sd[p + "layer_norm1.bias"])
    ff = F.linear(F.relu(F.linear(x, sd[p + "feed_forward.0.weight"], sd[p + "feed_forward.0.bias"])),
                  sd[p + "feed_forward.2.weight"], sd[p + "feed_forward.2.bias"])
    return F.layer_norm(x + ff, (E,), sd[p + "layer_norm2.weight"], sd[p + "layer_norm2.bias"])


def jmt_fusion(x, sd, p="fuse.", mt=False):
    """x: {'video': [B,128,L], 'vggish': [B,64,L]} -> [B,L,128]."""
    v = x["video"].permute(2, 0, 1)
    a = F.linear(x["vggish"].permute(2, 0, 1), sd[p + "augment_audio_feats_dim.weight"], sd[p + "augment_audio_feats_dim.bias"])
    ev = encoder_layer(v, sd, p + "visual_encoder.layers.0.")
    ea = encoder_layer(a, sd, p + "audio_encoder.layers.0.")
    if mt:
        feats = [mha(ev, ea, sd, p + "CA_va."), mha(ea, ev, sd, p + "CA_av.")]
    else:
        jr = F.linear(torch.cat((v, a), dim=2), sd[p + "reduce_feats_dim.weight"], sd[p + "reduce_feats_dim.bias"])
        ej = encoder_layer(jr, sd, p + "jr_encoder.layers.0.")
        feats = [mha(ev, ea, sd, p + "CA_va."), mha(ea, ev, sd, p + "CA_av."), mha(ej, ev, sd, p + "CA_jrv."),
                 mha(ev, ej, sd, p + "CA_vjr."), mha(ej, ea, sd, p + "CA_jra."), mha(ea, ej, sd, p + "CA_ajr.")]
    st = torch.stack(feats, dim=2)  # [L,B,n,128]
    length, bsz, n, _ = st.shape
    st = st.view(-1, n, E)          # seq = L*B, batch = n  (the cross-clip quirk)
    st = encoder_layer(st, sd, p + "final_encoder.layers.0.")
    out = mha(st, st, sd, p + "final_self_attention.")
    return out.view(length, bsz, n, E)[:, :, -1, :].permute(1, 0, 2)


def _encode(inputs, sd, modalities, train, backbone_train, new_buffers, tcn_masks=None):
    feats = {}
    for m in modalities:
        x = inputs[m]
        if m == "video":
            bsz, length = x.shape[:2]
            x = ir50_forward(x.reshape(-1, *x.shape[2:]), sd, "spatial.visual.backbone.", train=backbone_train,
                             new_buffers=new_buffers).view(bsz, length, -1)
        else:
            x = x.squeeze(1)
        x = tcn_forward(x.transpose(1, 2), sd, f"temporal.{m}.", (tcn_masks or {}).get(m))
        feats[m] = _bn1d(x, sd, f"bn.{m}", train, new_buffers)  # [B,C,L]
    return feats


def jmt_forward(inputs, sd, modalities, model_name="JMT", train=False, backbone_train=None, new_buffers=None):
    if backbone_train is None:
        backbone_train = train
    feats = _encode(inputs, sd, modalities, train, backbone_train, new_buffers)
    c = jmt_fusion(feats, sd, "fuse.", mt=(model_name == "MT"))
    c = F.linear(c, sd["fc1.weight"], sd["fc1.bias"]).transpose(1, 2)
    c = _bn1d(c, sd, "bn1", train, new_buffers).transpose(1, 2)
    return F.linear(F.leaky_relu(c), sd["fc2.weight"], sd["fc2.bias"])


def can_forward(inputs, sd, modalities, train=False, backbone_train=None, new_buffers=None):
    if backbone_train is None:
        backbone_train = train
    feats = _encode(inputs, sd, modalities, train, backbone_train, new_buffers)
    proj = [F.linear(feats[m].transpose(1, 2), sd[f"fuse.attn.{i}.weight"], sd[f"fuse.attn.{i}.bias"])
            for i, m in enumerate(modalities)]
    cat = torch.cat(proj, -1)
    c = torch.softmax(F.linear(cat, sd["fuse.weights.weight"], sd["fuse.weights.bias"]), dim=-1) * cat
    c = F.linear(c, sd["fc1.weight"], sd["fc1.bias"]).transpose(1, 2)
    c = _bn1d(c, sd, "bn1", train, new_buffers).transpose(1, 2)
    return F.linear(F.leaky_relu(c), sd["fc2.weight"], sd["fc2.bias"])
