"""Oracle: BERT-base token features as the reference extracts them (test infrastructure only).

The arithmetic lives in the third-party ``transformers`` package (NOT vendored in
/root/reference, version un-pinned: it arrives as a dependency of deepmultilingualpunctuation in
create_v_env_main.sh; the build container has transformers 5.15.0).  This file restates the
published BERT encoder (Devlin et al. 2018; HF ``BertModel`` in eval mode: post-LN blocks, exact
erf GELU, LayerNorm eps 1e-12, additive -inf padding mask) over a state dict with HF's key names,
and is pinned by ``tools/gen_golden.py`` against ``BertModel(BertConfig())`` built from a local
config object with seeded random weights.  The reference's use of it is anchored at its call
sites: abaw5_pre_processing/base/preprocessing.py:171-176 (model construction,
output_hidden_states=True), speech.py:589-626 (sum of the LAST FOUR of the 13 hidden states) and
speech.py:567-586 (drop [CLS], the last attended token and the padding).
"""
import math

import torch
import torch.nn.functional as F

LN_EPS = 1e-12


def bert_hidden_states(ids, attention_mask, sd, num_layers=12, num_heads=12, prefix=""):
    """ids, attention_mask: [B,S] long -> list of 13 hidden states [B,S,768]."""
    p = prefix
    b, s = ids.shape
    pos = torch.arange(s)
    x = (sd[p + "embeddings.word_embeddings.weight"][ids]
         + sd[p + "embeddings.position_embeddings.weight"][pos][None]
         + sd[p + "embeddings.token_type_embeddings.weight"][0][None, None])
    h = x.shape[-1]
    x = F.layer_norm(x, (h,), sd[p + "embeddings.LayerNorm.weight"], sd[p + "embeddings.LayerNorm.bias"], LN_EPS)
    hd = h // num_heads
    bias = torch.zeros(b, 1, 1, s)
    bias = bias.masked_fill(attention_mask[:, None, None, :] == 0, float("-inf"))
    states = [x]
    for i in range(num_layers):
        L = f"{p}encoder.layer.{i}."

        def lin(t, name):
            return F.linear(t, sd[L + name + ".weight"], sd[L + name + ".bias"])

        def heads(t):
            return t.view(b, s, num_heads, hd).transpose(1, 2)
        q, k, v = heads(lin(x, "attention.self.query")), heads(lin(x, "attention.self.key")), heads(lin(x, "attention.self.value"))
        att = torch.softmax(q @ k.transpose(-1, -2) / math.sqrt(hd) + bias, dim=-1)
        ctx = (att @ v).transpose(1, 2).reshape(b, s, h)
        x = F.layer_norm(lin(ctx, "attention.output.dense") + x, (h,), sd[L + "attention.output.LayerNorm.weight"],
                         sd[L + "attention.output.LayerNorm.bias"], LN_EPS)
        inter = F.gelu(lin(x, "intermediate.dense"))
        x = F.layer_norm(lin(inter, "output.dense") + x, (h,), sd[L + "output.LayerNorm.weight"],
                         sd[L + "output.LayerNorm.bias"], LN_EPS)
        states.append(x)
    return states


def bert_token_features(ids, attention_mask, sd, **kw):
    """Sum of the last four hidden states, [B,S,768] (speech.py:617-624)."""
    hs = bert_hidden_states(ids, attention_mask, sd, **kw)
    return hs[-1] + hs[-2] + hs[-3] + hs[-4]


def exclude_padding(token_vecs_sum, attention_mask):
    """speech.py:567-586: per sentence keep the attended tokens except the first ([CLS]) and the
    last attended one ([SEP]); a sentence that fills every slot is an error."""
    out = []
    for vecs, mask in zip(token_vecs_sum, attention_mask):
        idx = torch.nonzero(mask == 1).flatten()
        if len(idx) == len(mask):
            raise ValueError("The sentence is too long, enlarge the token number!")
        keep = mask.clone().bool()
        keep[0] = False
        keep[int(idx.max())] = False
        out.append(vecs[keep])
    return torch.cat(out, dim=0)
