"""BERT-base token-feature extractor on the HIP kernels.

The reference builds ``BertModel.from_pretrained('bert-base-uncased', output_hidden_states=True)``
(abaw5_pre_processing/base/preprocessing.py:171-176), runs it in eval mode per sentence
(speech.py:589-606), sums the LAST FOUR of the 13 hidden states (speech.py:617-624) and drops
[CLS], the last attended token and the padding (speech.py:567-586).  This module keeps the
transformers ``BertModel`` state-dict key names (so a real checkpoint loads with strict=True,
``embeddings.position_ids``-style buffers excepted) and runs:

  embeddings  : gather + LayerNorm fused (one wave per token)
  per layer   : one fused QKV GEMM [tokens,768]x[768,2304] -> flash attention (fp32 MFMA,
                key-padding mask) -> out-proj GEMM with the residual in its epilogue -> LayerNorm
                -> FFN GEMM with exact-erf GELU epilogue -> FFN GEMM with residual epilogue -> LayerNorm
  output      : running sum of the last four hidden states.
"""
import torch
from torch import nn

from . import ops

LN_EPS = 1e-12


class _Holder(nn.Module):
    pass


def _linear_holder(i, o):
    return nn.Linear(i, o)


class BertEncoderHIP(nn.Module):
    def __init__(self, vocab_size=30522, hidden_size=768, num_hidden_layers=12, num_attention_heads=12,
                 intermediate_size=3072, max_position_embeddings=512, type_vocab_size=2):
        super().__init__()
        self.hidden, self.heads, self.layers_n = hidden_size, num_attention_heads, num_hidden_layers
        e = _Holder()
        e.word_embeddings = nn.Embedding(vocab_size, hidden_size)
        e.position_embeddings = nn.Embedding(max_position_embeddings, hidden_size)
        e.token_type_embeddings = nn.Embedding(type_vocab_size, hidden_size)
        e.LayerNorm = nn.LayerNorm(hidden_size, eps=LN_EPS)
        self.embeddings = e
        enc = _Holder()
        layers = []
        for _ in range(num_hidden_layers):
            L = _Holder()
            L.attention = _Holder()
            L.attention.self = _Holder()
            L.attention.self.query = nn.Linear(hidden_size, hidden_size)
            L.attention.self.key = nn.Linear(hidden_size, hidden_size)
            L.attention.self.value = nn.Linear(hidden_size, hidden_size)
            L.attention.output = _Holder()
            L.attention.output.dense = nn.Linear(hidden_size, hidden_size)
            L.attention.output.LayerNorm = nn.LayerNorm(hidden_size, eps=LN_EPS)
            L.intermediate = _Holder()
            L.intermediate.dense = nn.Linear(hidden_size, intermediate_size)
            L.output = _Holder()
            L.output.dense = nn.Linear(intermediate_size, hidden_size)
            L.output.LayerNorm = nn.LayerNorm(hidden_size, eps=LN_EPS)
            layers.append(L)
        enc.layer = nn.ModuleList(layers)
        self.encoder = enc
        self.pooler = _Holder()
        self.pooler.dense = nn.Linear(hidden_size, hidden_size)  # unused by the feature path, kept for the keys
        for p in self.parameters():
            p.requires_grad = False
        self._packed, self._key = None, None
        # "fp32": exact fp32 MFMA GEMMs; "bf16x3": split hi / lo bf16 operands, three MFMAs per product (<= 2^-15 relative per
        # product, 5x the fp32 matrix rate -- what VGGish and IR-50 run by default); "bf16" / "fp16": the GEMM operands
        # (activations and weights) as one 16-bit plane, one MFMA per product, fp32 accumulate -- what autocast does to
        # nn.Linear (the reference's --amp recipe); attention, LayerNorm, GELU, residuals and the hidden-state sum stay fp32
        self.precision = "fp32"
        self._packed_n16 = None

    def _pack(self):
        key = tuple((p.data_ptr(), p._version) for p in self.parameters())
        if self._packed is None or key != self._key:
            if self.embeddings.word_embeddings.weight.device.type != "cuda":
                raise RuntimeError("BertEncoderHIP runs on the HIP kernels only: move it to a GPU (no CPU fallback)")
            packed = []
            for L in self.encoder.layer:
                s = L.attention.self
                packed.append((torch.cat([s.query.weight, s.key.weight, s.value.weight], 0).detach().contiguous(),
                               torch.cat([s.query.bias, s.key.bias, s.value.bias], 0).detach().contiguous()))
            self._packed, self._key = packed, key
        return self._packed

    def _pack_n16(self, dtype):
        """``dtype``: torch.bfloat16 / torch.float16 (one plane per weight) or "split" (hi / lo bf16 planes: bf16x3)."""
        packed = self._pack()
        if self._packed_n16 is None or self._packed_n16[0] != (dtype, self._key):
            conv = ops.split_bf16 if dtype == "split" else (lambda w: ops.to_n16(w, dtype))
            layers = []
            for (wqkv, _), L in zip(packed, self.encoder.layer):
                layers.append(tuple(conv(w.detach().contiguous()) for w in
                                    (wqkv, L.attention.output.dense.weight, L.intermediate.dense.weight, L.output.dense.weight)))
            self._packed_n16 = ((dtype, self._key), layers)
        return self._packed_n16[1]

    @staticmethod
    def _linear_n16(x, w16, bias, act=ops.ACT_NONE, residual=None):
        """y = act(x @ w^T + bias) + residual with narrow operands: x [rows, K] fp32 is rounded once, the result is fp32."""
        rows, k = x.shape
        r = None if residual is None else residual.view(rows, 1, 1, -1)
        if isinstance(w16, ops.Split):      # bf16x3: both operands as split tensors
            return ops.conv2d_b3(ops.split_bf16(x).view(rows, 1, 1, k), w16, 1, 1, bias=bias, act1=act, residual=r, out_f32=True,
                                 out_split=False)["y"].view(rows, -1)
        return ops.conv2d_n16(ops.to_n16(x, w16.dtype).view(rows, 1, 1, k), w16, 1, 1, bias=bias, act1=act, residual=r,
                              out_f32=True, out_n16=False)["y"].view(rows, -1)

    def __deepcopy__(self, memo):
        import copy
        packed, self._packed = self._packed, None
        self._packed_n16 = None
        try:
            new = self.__class__.__new__(self.__class__)
            memo[id(self)] = new
            new.__dict__ = copy.deepcopy(self.__dict__, memo)
        finally:
            self._packed = packed
        return new

    @torch.no_grad()
    def forward(self, input_ids, attention_mask=None, last_n_sum=4):
        """ids [B,S] int64, mask [B,S] (1 = token) -> sum of the last ``last_n_sum`` hidden states [B,S,768]."""
        packed = self._pack()
        dev = self.embeddings.word_embeddings.weight.device
        ids = input_ids.to(dev).long().contiguous()
        b, s = ids.shape
        mask = None if attention_mask is None else attention_mask.to(dev).to(torch.int32).contiguous()
        e = self.embeddings
        x = ops.bert_embed_ln(ids, e.word_embeddings.weight, e.position_embeddings.weight, e.token_type_embeddings.weight,
                              e.LayerNorm.weight, e.LayerNorm.bias, LN_EPS).view(b * s, self.hidden)
        hd, H, hdim = self.hidden, self.heads, self.hidden // self.heads
        total = None
        n_layers = len(self.encoder.layer)
        if last_n_sum > n_layers:
            total = x.clone()
        if self.precision not in ("fp32", "bf16x3", "bf16", "fp16"):
            raise ValueError(f"unknown precision {self.precision!r}")
        n16 = None if self.precision == "fp32" else self._pack_n16(
            {"bf16x3": "split", "bf16": torch.bfloat16, "fp16": torch.float16}[self.precision])
        for i, L in enumerate(self.encoder.layer):
            wqkv, bqkv = packed[i]
            if n16 is not None:
                x, total = self._layer_n16(x, total, i, L, n16[i], bqkv, b, s, mask, n_layers, last_n_sum)
                continue
            qkv = ops.linear(x, wqkv, bias=bqkv)  # [tokens, 3*hidden]
            ctx = torch.empty((b * s, hd), device=dev, dtype=torch.float32)
            st = (s * 3 * hd, 3 * hd, hdim)
            ops.attention(qkv, qkv[:, hd:], qkv[:, 2 * hd:], ctx, b, H, s, s, hdim, st, st, st, (s * hd, hd, hdim),
                          1.0 / hdim ** 0.5, key_mask=mask)
            ao = L.attention.output
            y = ops.linear(ctx, ao.dense.weight, bias=ao.dense.bias, residual=x)
            x1, _, _ = ops.layernorm_fwd(y, ao.LayerNorm.weight, ao.LayerNorm.bias, eps=LN_EPS, save=False)
            inter = ops.linear(x1, L.intermediate.dense.weight, bias=L.intermediate.dense.bias, act=ops.ACT_GELU)
            y2 = ops.linear(inter, L.output.dense.weight, bias=L.output.dense.bias, residual=x1)
            x, _, _ = ops.layernorm_fwd(y2, L.output.LayerNorm.weight, L.output.LayerNorm.bias, eps=LN_EPS, save=False)
            if i >= n_layers - last_n_sum:
                total = x.clone() if total is None else ops.add_inplace(total, x)
        return total.view(b, s, hd)

    def _layer_n16(self, x, total, i, L, w16, bqkv, b, s, mask, n_layers, last_n_sum):
        hd, H, hdim = self.hidden, self.heads, self.hidden // self.heads
        wqkv, wo, wi, wf = w16
        qkv = self._linear_n16(x, wqkv, bqkv)
        ctx = torch.empty((b * s, hd), device=x.device, dtype=torch.float32)
        st = (s * 3 * hd, 3 * hd, hdim)
        ops.attention(qkv, qkv[:, hd:], qkv[:, 2 * hd:], ctx, b, H, s, s, hdim, st, st, st, (s * hd, hd, hdim),
                      1.0 / hdim ** 0.5, key_mask=mask)
        ao = L.attention.output
        y = self._linear_n16(ctx, wo, ao.dense.bias, residual=x)
        x1, _, _ = ops.layernorm_fwd(y, ao.LayerNorm.weight, ao.LayerNorm.bias, eps=LN_EPS, save=False)
        inter = self._linear_n16(x1, wi, L.intermediate.dense.bias, act=ops.ACT_GELU)
        y2 = self._linear_n16(inter, wf, L.output.dense.bias, residual=x1)
        x, _, _ = ops.layernorm_fwd(y2, L.output.LayerNorm.weight, L.output.LayerNorm.bias, eps=LN_EPS, save=False)
        if i >= n_layers - last_n_sum:
            total = x.clone() if total is None else ops.add_inplace(total, x)
        return x, total

    @staticmethod
    def exclude_padding(token_vecs_sum, attention_mask):
        """speech.py:567-586 (host-side indexing): keep attended tokens except the first and the last
        attended one; a sentence that fills every slot raises like the reference."""
        out = []
        for vecs, m in zip(token_vecs_sum, attention_mask.to(token_vecs_sum.device)):
            idx = torch.nonzero(m == 1).flatten()
            if len(idx) == len(m):
                raise ValueError("The sentence is too long, enlarge the token number!")
            keep = m.clone().bool()
            keep[0] = False
            keep[int(idx.max())] = False
            out.append(vecs[keep])
        return torch.cat(out, dim=0)
