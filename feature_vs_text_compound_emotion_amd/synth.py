"""Seeded synthetic weights and inputs.

There is no network for checkpoints or datasets, so tests, smoke() and bench.py
all draw weights and clips from here.  Weight tensors are generated from a
seed in a fixed key order, so a test can rebuild on the GPU box exactly the
state dict that ``tools/gen_golden.py`` loaded into the reference model when it
wrote the fixtures -- the big tensors never need to be stored.

Key names and shapes are the reference's (models/backbone.py:69-130,
models/arcface_model.py:121-151, models/temporal_convolutional_model.py:21-75,
models/transformer.py:102-197, models/model.py:451-485).
"""
from collections import OrderedDict

import torch

IR50_STAGES = ((64, 64, 3, 1), (64, 128, 4, 2), (128, 256, 14, 2), (256, 512, 3, 2))

EMBEDDING_DIM = {"video": 512, "vggish": 128, "bert": 768, "logmel": 128}
ENCODER_DIM = {"video": 128, "vggish": 32, "bert": 128, "logmel": 32}
# configs.py:61-73 (the LFAN channel table: BERT_TEMPORAL_DIM = 512)
TCN_CHANNELS = {"video": [256, 256, 128, 128], "vggish": [64, 64, 32, 32], "logmel": [64, 64, 32, 32],
                "bert": [256, 256, 128, 128]}


def ir50_units():
    units = []
    for cin, depth, n, stride in IR50_STAGES:
        units.append((cin, depth, stride))
        units.extend((depth, depth, 1) for _ in range(n - 1))
    return units


def _bn_spec(spec, key, c):
    spec[key + ".weight"] = ((c,), "bn_w")
    spec[key + ".bias"] = ((c,), "bn_b")
    spec[key + ".running_mean"] = ((c,), "bn_m")
    spec[key + ".running_var"] = ((c,), "bn_v")
    spec[key + ".num_batches_tracked"] = ((), "count")


def visual_backbone_spec(prefix="", head_hw=5, num_classes=8):
    """Ordered {key: (shape, kind)} for VisualBackbone (351 entries)."""
    s = OrderedDict()
    p = prefix + "backbone."
    s[p + "input_layer.0.weight"] = ((64, 3, 3, 3), "conv")
    _bn_spec(s, p + "input_layer.1", 64)
    s[p + "input_layer.2.weight"] = ((64,), "prelu")
    _bn_spec(s, p + "output_layer.0", 512)
    s[p + "output_layer.3.weight"] = ((512, 512 * head_hw * head_hw), "linear")
    s[p + "output_layer.3.bias"] = ((512,), "bias")
    _bn_spec(s, p + "output_layer.4", 512)
    for i, (cin, depth, stride) in enumerate(ir50_units()):
        b = f"{p}body.{i}."
        if cin != depth:
            s[b + "shortcut_layer.0.weight"] = ((depth, cin, 1, 1), "conv")
            _bn_spec(s, b + "shortcut_layer.1", depth)
        _bn_spec(s, b + "res_layer.0", cin)
        s[b + "res_layer.1.weight"] = ((depth, cin, 3, 3), "conv")
        s[b + "res_layer.2.weight"] = ((depth,), "prelu")
        s[b + "res_layer.3.weight"] = ((depth, depth, 3, 3), "conv")
        _bn_spec(s, b + "res_layer.4", depth)
    s[prefix + "logits.weight"] = ((num_classes, 512), "linear")
    s[prefix + "logits.bias"] = ((num_classes,), "bias")
    return s


def tcn_spec(prefix, cin, channels, k):
    """TemporalConvNet keys; ``net.0/net.4`` alias ``conv1/conv2`` (same tensors)."""
    s = OrderedDict()
    alias = {}
    for i, cout in enumerate(channels):
        b = f"{prefix}network.{i}."
        c_in = cin if i == 0 else channels[i - 1]
        for conv, ci in (("conv1", c_in), ("conv2", cout)):
            s[b + conv + ".bias"] = ((cout,), "bias")
            s[b + conv + ".weight_g"] = ((cout, 1, 1), "wn_g")
            s[b + conv + ".weight_v"] = ((cout, ci, k), "wn_v")
        for net, conv in (("net.0", "conv1"), ("net.4", "conv2")):
            for leaf in ("bias", "weight_g", "weight_v"):
                alias[b + net + "." + leaf] = b + conv + "." + leaf
        if c_in != cout:
            s[b + "downsample.weight"] = ((cout, c_in, 1), "conv")
            s[b + "downsample.bias"] = ((cout,), "bias")
    return s, alias


def lfan_spec(modalities, n_cls=7, head_hw=5, kernel_size=5, modal_dim=32):
    s = OrderedDict()
    alias = {}
    for m in modalities:
        t, a = tcn_spec(f"temporal.{m}.", EMBEDDING_DIM[m], TCN_CHANNELS[m], kernel_size)
        s.update(t)
        alias.update(a)
    if "video" in modalities:
        s.update(visual_backbone_spec("spatial.visual.", head_hw))
    if "logmel" in modalities:
        s.update(vggish_spec("spatial.audio.backbone."))
    for m in modalities:
        _bn_spec(s, f"bn.{m}", TCN_CHANNELS[m][-1])
    for m in modalities:
        s[f"fusion.layers.self_attn.qkv_proj.{m}.weight"] = ((3 * modal_dim, ENCODER_DIM[m]), "linear")
        s[f"fusion.layers.self_attn.qkv_proj.{m}.bias"] = ((3 * modal_dim,), "bias")
    d = modal_dim * len(modalities)
    s["fusion.layers.self_attn.o_proj.weight"] = ((d, d), "linear")
    s["fusion.layers.self_attn.o_proj.bias"] = ((d,), "bias")
    s["fusion.layers.norm1.weight"] = ((d,), "bn_w")
    s["fusion.layers.norm1.bias"] = ((d,), "bn_b")
    s["regressor.weight"] = ((n_cls, ENCODER_DIM[modalities[0]] + d), "linear")
    s["regressor.bias"] = ((n_cls,), "bias")
    return s, alias


# configs.py:75-127 (tcn_settings used by CAN / JMT / MT)
TCN_SETTINGS = {"video": {"input_dim": 512, "channel": [256, 256, 128, 128, 128], "kernel_size": 5},
                "vggish": {"input_dim": 128, "channel": [128, 128, 64, 64], "kernel_size": 5},
                "bert": {"input_dim": 768, "channel": [256, 256, 128, 128], "kernel_size": 5}}


def _mha_spec(s, p, e=128):
    s[p + "in_proj_weight"] = ((3 * e, e), "linear")
    s[p + "in_proj_bias"] = ((3 * e,), "bias")
    s[p + "out_proj.weight"] = ((e, e), "linear")
    s[p + "out_proj.bias"] = ((e,), "bias")


def _enc_layer_spec(s, p, e=128, hidden=128):
    _mha_spec(s, p + "layers.0.attention.", e)
    s[p + "layers.0.feed_forward.0.weight"] = ((hidden, e), "linear")
    s[p + "layers.0.feed_forward.0.bias"] = ((hidden,), "bias")
    s[p + "layers.0.feed_forward.2.weight"] = ((e, hidden), "linear")
    s[p + "layers.0.feed_forward.2.bias"] = ((e,), "bias")
    for ln in ("layer_norm1", "layer_norm2"):
        s[p + f"layers.0.{ln}.weight"] = ((e,), "bn_w")
        s[p + f"layers.0.{ln}.bias"] = ((e,), "bn_b")


def _tail_common_spec(modalities, head_hw):
    s, alias = OrderedDict(), {}
    for m in modalities:
        cfg = TCN_SETTINGS[m]
        t, a = tcn_spec(f"temporal.{m}.", cfg["input_dim"], cfg["channel"], cfg["kernel_size"])
        s.update(t)
        alias.update(a)
        _bn_spec(s, f"bn.{m}", cfg["channel"][-1])
    if "video" in modalities:
        s.update(visual_backbone_spec("spatial.visual.", head_hw))
    return s, alias


def jmt_spec(modalities=("video", "vggish"), model_name="JMT", n_cls=7, head_hw=5):
    """JMT / MT keys (models/model.py:895-1167).  MTFusion keeps an unused reduce_feats_dim."""
    s, alias = _tail_common_spec(modalities, head_hw)
    f = "fuse."
    encs = ["visual_encoder", "audio_encoder"] + (["jr_encoder"] if model_name == "JMT" else [])
    for e in encs:
        _enc_layer_spec(s, f + e + ".")
    cas = ["CA_va", "CA_av"] + (["CA_jra", "CA_ajr", "CA_vjr", "CA_jrv"] if model_name == "JMT" else [])
    for c in cas:
        _mha_spec(s, f + c + ".")
    s[f + "reduce_feats_dim.weight"] = ((128, 256), "linear")
    s[f + "reduce_feats_dim.bias"] = ((128,), "bias")
    s[f + "augment_audio_feats_dim.weight"] = ((128, 64), "linear")
    s[f + "augment_audio_feats_dim.bias"] = ((128,), "bias")
    _enc_layer_spec(s, f + "final_encoder.")
    _mha_spec(s, f + "final_self_attention.")
    _bn_spec(s, "bn1", 128)
    for k, shp in (("fc1", (128, 128)), ("fc2", (n_cls, 128))):
        s[k + ".weight"] = (shp, "linear")
        s[k + ".bias"] = ((shp[0],), "bias")
    return s, alias


def can_spec(modalities=("video", "vggish"), n_cls=7, head_hw=5):
    """CAN keys (models/model.py:529-684); conv_c is defined but never used by forward."""
    s, alias = _tail_common_spec(modalities, head_hw)
    m = len(modalities)
    for i, mod in enumerate(modalities):
        s[f"fuse.attn.{i}.weight"] = ((128, TCN_SETTINGS[mod]["channel"][-1]), "linear")
        s[f"fuse.attn.{i}.bias"] = ((128,), "bias")
    s["fuse.weights.weight"] = ((128 * m, 128 * m), "linear")
    s["fuse.weights.bias"] = ((128 * m,), "bias")
    s["conv_c.weight"] = ((128, 128 * m, 1), "conv")
    s["conv_c.bias"] = ((128,), "bias")
    _bn_spec(s, "bn1", 128 * m)
    s["fc1.weight"] = ((128 * m, 128 * m), "linear")
    s["fc1.bias"] = ((128 * m,), "bias")
    s["fc2.weight"] = ((n_cls, 128 * m), "linear")
    s["fc2.bias"] = ((n_cls,), "bias")
    return s, alias


VGGISH_CONVS = ((0, 1, 64), (3, 64, 128), (6, 128, 256), (8, 256, 256), (11, 256, 512), (13, 512, 512))


def vggish_spec(prefix=""):
    """VGG/VGGish keys (models/backbone.py:16-66): features.N conv + bias, embeddings.{0,2,4}."""
    s = OrderedDict()
    for idx, cin, cout in VGGISH_CONVS:
        s[f"{prefix}features.{idx}.weight"] = ((cout, cin, 3, 3), "conv_relu")
        s[f"{prefix}features.{idx}.bias"] = ((cout,), "bias")
    for idx, cin, cout in ((0, 512 * 4 * 6, 4096), (2, 4096, 4096), (4, 4096, 128)):
        s[f"{prefix}embeddings.{idx}.weight"] = ((cout, cin), "linear_relu")
        s[f"{prefix}embeddings.{idx}.bias"] = ((cout,), "bias")
    return s


def bert_spec(prefix="", layers=12, hidden=768, inter=3072, vocab=30522, max_pos=512, pooler=True):
    """HF BertModel keys for a bert-base-shaped encoder (local BertConfig(), 199 entries)."""
    s = OrderedDict()
    e = prefix + "embeddings."
    s[e + "word_embeddings.weight"] = ((vocab, hidden), "bert_w")
    s[e + "position_embeddings.weight"] = ((max_pos, hidden), "bert_w")
    s[e + "token_type_embeddings.weight"] = ((2, hidden), "bert_w")
    s[e + "LayerNorm.weight"] = ((hidden,), "bn_w")
    s[e + "LayerNorm.bias"] = ((hidden,), "bias")
    for i in range(layers):
        L = f"{prefix}encoder.layer.{i}."
        for name, (o, n) in (("attention.self.query", (hidden, hidden)), ("attention.self.key", (hidden, hidden)),
                             ("attention.self.value", (hidden, hidden)), ("attention.output.dense", (hidden, hidden))):
            s[L + name + ".weight"] = ((o, n), "bert_lin")
            s[L + name + ".bias"] = ((o,), "bias")
        s[L + "attention.output.LayerNorm.weight"] = ((hidden,), "bn_w")
        s[L + "attention.output.LayerNorm.bias"] = ((hidden,), "bias")
        s[L + "intermediate.dense.weight"] = ((inter, hidden), "bert_lin")
        s[L + "intermediate.dense.bias"] = ((inter,), "bias")
        s[L + "output.dense.weight"] = ((hidden, inter), "bert_lin")
        s[L + "output.dense.bias"] = ((hidden,), "bias")
        s[L + "output.LayerNorm.weight"] = ((hidden,), "bn_w")
        s[L + "output.LayerNorm.bias"] = ((hidden,), "bias")
    if pooler:
        s[prefix + "pooler.dense.weight"] = ((hidden, hidden), "bert_lin")
        s[prefix + "pooler.dense.bias"] = ((hidden,), "bias")
    return s


def make_audio_int16(seconds=1.0, sample_rate=16000, seed=4321):
    """int16 PCM, round(3000*N(0,1)) clipped."""
    g = torch.Generator().manual_seed(seed)
    n = int(round(seconds * sample_rate))
    return torch.clamp(torch.round(torch.randn(n, generator=g) * 3000.0), -32768, 32767).to(torch.int16)


def make_token_ids(batch, length, seed=777, pad_from=None):
    """[CLS]=101, U{1000..30521}, [SEP]=102; optional zero padding from ``pad_from``."""
    g = torch.Generator().manual_seed(seed)
    ids = torch.randint(1000, 30522, (batch, length), generator=g)
    mask = torch.ones(batch, length, dtype=torch.long)
    ids[:, 0] = 101
    for b in range(batch):
        end = length if pad_from is None else pad_from[b]
        ids[b, end - 1] = 102
        ids[b, end:] = 0
        mask[b, end:] = 0
    return ids, mask


def _draw(shape, kind, g):
    if kind == "count":
        return torch.zeros((), dtype=torch.long)
    if kind == "conv":
        fan_in = 1
        for d in shape[1:]:
            fan_in *= d
        return torch.randn(shape, generator=g) * (1.0 / fan_in) ** 0.5
    if kind == "conv_relu":  # He init keeps ReLU stacks (VGGish) at O(1)
        fan_in = 1
        for d in shape[1:]:
            fan_in *= d
        return torch.randn(shape, generator=g) * (2.0 / fan_in) ** 0.5
    if kind == "linear_relu":
        return torch.randn(shape, generator=g) * (2.0 / shape[1]) ** 0.5
    if kind == "bert_w":
        return torch.randn(shape, generator=g) * 0.05
    if kind == "bert_lin":
        return torch.randn(shape, generator=g) * (1.0 / shape[1]) ** 0.5
    if kind == "linear":
        bound = (1.0 / shape[1]) ** 0.5
        return (torch.rand(shape, generator=g) * 2 - 1) * bound
    if kind == "bias" or kind == "bn_b" or kind == "bn_m":
        return torch.randn(shape, generator=g) * 0.1
    if kind == "bn_w" or kind == "bn_v" or kind == "wn_g":
        return torch.rand(shape, generator=g) + 0.5
    if kind == "prelu":
        return torch.rand(shape, generator=g) * 0.3 + 0.1
    if kind == "wn_v":
        return torch.randn(shape, generator=g) * 0.05
    raise ValueError(kind)


def make_state_dict(spec, alias=None, seed=0):
    """Draw every tensor of ``spec`` in order from one seeded CPU generator."""
    g = torch.Generator().manual_seed(seed)
    sd = OrderedDict((k, _draw(shape, kind, g)) for k, (shape, kind) in spec.items())
    for k, src in (alias or {}).items():
        sd[k] = sd[src]
    return sd


def lfan_state_dict(modalities, n_cls=7, head_hw=5, seed=0, conditioned=False):
    """``conditioned``: the same draw with the biases of the VIDEO temporal net scaled by 0.1.  The default draw gives the
    TCN biases N(0, 0.1) against a signal of ~0.03 (unit-norm 512-d embeddings through weight-normed filters), so about
    half of its LeakyReLU channels are switched off for every frame and a few sit right at the switch: one pre-activation
    changing sign under a 1e-3 relative perturbation of the embeddings multiplies that element's gradient by 100 (slope
    0.01 -> 1) behind a batch-statistics BatchNorm1d whose gain on an almost constant channel is 1 / sqrt(eps) = 316, and
    the tail's gradient moves by tens of percent (tests/test_conditioning_cpu.py shows it on the CPU oracle alone).  With
    the biases at the scale of the signal the channels carry mixed signs and the same perturbation moves the gradient by
    a few percent -- the well-conditioned problem on which update-level comparisons between precisions are meaningful."""
    spec, alias = lfan_spec(modalities, n_cls=n_cls, head_hw=head_hw)
    sd = make_state_dict(spec, alias, seed)
    if conditioned:
        for k, v in sd.items():
            if k.startswith("temporal.video.") and k.endswith(".bias") and k not in alias:
                v.mul_(0.1)
    return sd


def make_clip_batch(modalities, batch, length, hw=40, seed=1234, n_cls=7):
    """Synthetic clips: uint8-uniform frames normalised to [-1,1],
    N(0,1) pre-computed vggish/bert features, one class per clip repeated over frames."""
    g = torch.Generator().manual_seed(seed)
    x = OrderedDict()
    for m in modalities:
        if m == "video":
            u8 = torch.randint(0, 256, (batch, length, hw, hw, 3), generator=g, dtype=torch.uint8)
            x[m] = ((u8.float() / 255.0 - 0.5) / 0.5).permute(0, 1, 4, 2, 3).contiguous()
        elif m == "logmel":   # [B, 64 mel bins, L, 96 frames] (model.py:500), values in the range of log(mel + 0.01)
            x[m] = torch.randn(batch, 64, length, 96, generator=g) * 2.0 - 1.0
        else:
            x[m] = torch.randn(batch, 1, length, EMBEDDING_DIM[m], generator=g)
    cls = torch.randint(0, n_cls, (batch,), generator=g)
    labels = cls.view(batch, 1, 1).expand(batch, length, 1).float().contiguous()
    return x, labels


def dropout_mask(shape, p, g):
    """Pre-scaled keep mask (values 0 or 1/(1-p)) drawn on the CPU generator ``g``."""
    return (torch.rand(shape, generator=g) >= p).float() / (1.0 - p)
