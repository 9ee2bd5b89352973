"""Oracle: LFAN cross-modal attention fusion (test infrastructure only).

Follows /root/reference/models/transformer.py:11-19 (scaled_dot_product),
:102-165 (MultimodalMultiheadAttention), :168-197 (MultiModalEncoderBlock).
Attention runs over the MODALITY axis (M x M per batch, head, frame), not time.
"""
import math

import torch
import torch.nn.functional as F

LN_EPS = 1e-5


def lfan_fusion_forward(x, sd, modalities, modal_dim, num_heads, prefix="fusion.layers.",
                        dropout_mask=None):
    """x: dict modality -> [B,L,enc_dim_m]; returns [B,L,modal_dim*M]."""
    m0 = modalities[0]
    bsz, length, _ = x[m0].shape
    hd = modal_dim // num_heads
    qs, ks, vs = [], [], []
    for m in modalities:
        qkv = F.linear(x[m], sd[f"{prefix}self_attn.qkv_proj.{m}.weight"],
                       sd[f"{prefix}self_attn.qkv_proj.{m}.bias"])
        # per head the 3*hd slice is [q | k | v]  (transformer.py:142-144)
        qkv = qkv.reshape(bsz, length, num_heads, 3 * hd)
        qs.append(qkv[..., :hd])
        ks.append(qkv[..., hd:2 * hd])
        vs.append(qkv[..., 2 * hd:])
    q = torch.stack(qs, dim=3)  # [B,L,H,M,hd]
    k = torch.stack(ks, dim=3)
    v = torch.stack(vs, dim=3)
    logits = torch.einsum("blhmd,blhnd->blhmn", q, k) / math.sqrt(hd)
    attn = torch.softmax(logits, dim=-1)
    vals = torch.einsum("blhmn,blhnd->blhmd", attn, v) + v  # values += V (:157)
    vals = vals.reshape(bsz, length, modal_dim * len(modalities))  # [.., H, M, hd] flattened
    o = F.linear(vals, sd[f"{prefix}self_attn.o_proj.weight"], sd[f"{prefix}self_attn.o_proj.bias"])
    if dropout_mask is not None:
        o = o * dropout_mask
    return F.layer_norm(o, (o.shape[-1],), sd[f"{prefix}norm1.weight"], sd[f"{prefix}norm1.bias"], LN_EPS)
