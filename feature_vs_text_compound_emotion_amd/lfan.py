"""LFAN (leader-follower attention network) on the HIP kernels.

Mirror of the reference's ``LFAN`` (models/model.py:375-526), its
``MultimodalTransformerEncoder`` fusion (models/transformer.py:102-215) and the
criterion ``nn.CrossEntropyLoss`` (experiment.py:133): same constructor
arguments, ``.init()``, ``forward(dict) -> [B, L, n_cls]``, ``state_dict``
keys, ``model.spatial['visual']``.

Execution plan for one forward (rows = B*L frames, channels-last):
  video frames --IR50 (frozen, fp32 MFMA convs)--> [rows,512]
  per modality: TCN (one autograd node) -> [rows, C_m]
  LFAN head (one autograd node): BatchNorm1d per modality -> QKV GEMMs -> M x M cross-modal
     attention -> o_proj GEMM -> dropout -> LayerNorm -> [leader | follower] -> regressor GEMM.
  The leader/follower concat is never materialised by a copy: BatchNorm and LayerNorm write
  straight into column slices of the regressor's input buffer.
"""
from os.path import join

import torch
from torch import nn

from . import ops
from .temporal_convnet import TemporalConvNet
from .audio_backbone import AudioBackbone
from .visual_backbone import VisualBackbone

CLASSIFICATION, REGRESSION = "CLASSIFICATION", "REGRESSION"  # reference constants.py:17-20
TASKS = [CLASSIFICATION, REGRESSION]
LN_EPS, BN_EPS, BN_MOMENTUM = 1e-5, 1e-5, 0.1


def _linear_T(dy, w, residual=None):
    """dX = dY @ W for a Linear with weight W [out, in]."""
    wt = ops.pack_conv_weight(w.view(w.shape[0], w.shape[1], 1, 1), transpose=True)
    return ops.linear(dy, wt, residual=residual)


def _packed(w):
    """Linear weight [out, in] -> kernel layout (only pads K to a multiple of 32)."""
    return w if w.shape[1] % 32 == 0 and w.is_contiguous() else ops.pack_conv_weight(w.view(*w.shape, 1, 1))


class LFANHeadFunction(torch.autograd.Function):
    """BN1d x M -> cross-modal attention fusion -> LayerNorm -> concat -> regressor, one node.

    args: (meta, buffers, fusion_mask, *tensors) where tensors =
      [t_m for each modality] + per modality [bn_w, bn_b, qkv_w, qkv_b] + [o_w, o_b, ln_w, ln_b, r_w, r_b]
    meta = (M, H, hd, train, sink); buffers = [(running_mean, running_var)] per modality (updated in place); ``sink``: None or a
    list that receives the M BatchNorm outputs (detached views; what the reference leaves in the caller's dict).
    """

    @staticmethod
    def forward(ctx, meta, buffers, fmask, *ts):
        M, H, hd, train, sink = meta
        t = [x.contiguous() for x in ts[:M]]
        per = [ts[M + 4 * i:M + 4 * i + 4] for i in range(M)]
        o_w, o_b, ln_w, ln_b, r_w, r_b = ts[M + 4 * M:]
        rows, enc0, d = t[0].shape[0], t[0].shape[1], H * hd * M
        z = torch.empty((rows, enc0 + d), device=t[0].device, dtype=torch.float32)
        ys, stats, qkvs = [], [], []
        for i in range(M):
            bn_w, bn_b, q_w, q_b = per[i]
            out = z[:, :enc0] if i == 0 else None
            y, sm, si = ops.bn_rows_fwd(t[i], bn_w, bn_b, buffers[i][0], buffers[i][1], train, BN_EPS, BN_MOMENTUM,
                                        out=out)
            if not train:  # eval-mode backward (rare) needs the statistics actually used
                sm, si = buffers[i][0], torch.rsqrt(buffers[i][1] + BN_EPS)
            ys.append(y)
            stats.append((sm, si))
            qkvs.append(ops.linear(y, _packed(q_w), bias=q_b))
        vals, probs = ops.lfan_attn_fwd(qkvs, H, hd)
        o = ops.linear(vals, _packed(o_w), bias=o_b)
        _, ln_mean, ln_rstd = ops.layernorm_fwd(o, ln_w, ln_b, mask=fmask, eps=LN_EPS, out=z[:, enc0:])
        logits = ops.linear(z, _packed(r_w), bias=r_b)
        if sink is not None:
            sink.extend(y.detach() for y in ys)
        ctx.meta, ctx.ts = meta[:4], ts
        ctx.saved = (t, ys, stats, qkvs, vals, probs, o, ln_mean, ln_rstd, z, fmask, enc0)
        return logits

    @staticmethod
    def backward(ctx, dlogits):
        M, H, hd, train = ctx.meta
        ts = ctx.ts
        t, ys, stats, qkvs, vals, probs, o, ln_mean, ln_rstd, z, fmask, enc0 = ctx.saved
        per = [ts[M + 4 * i:M + 4 * i + 4] for i in range(M)]
        o_w, o_b, ln_w, ln_b, r_w, r_b = ts[M + 4 * M:]
        rows = z.shape[0]
        dlogits = dlogits.contiguous()
        g = [None] * len(ts)
        base = M + 4 * M
        g[base + 4] = ops.conv1d_wgrad(dlogits, z, rows, 1, 1).view_as(r_w)
        g[base + 5] = ops.col_sum(dlogits)
        dz = _linear_T(dlogits, r_w)  # [rows, enc0 + d]
        do, g[base + 2], g[base + 3] = ops.layernorm_bwd(dz[:, enc0:], o, ln_w, ln_mean, ln_rstd, mask=fmask)
        g[base + 0] = ops.conv1d_wgrad(do, vals, rows, 1, 1).view_as(o_w)
        g[base + 1] = ops.col_sum(do)
        dvals = _linear_T(do, o_w)
        dqkv = ops.lfan_attn_bwd(qkvs, dvals, probs, H, hd)
        for i in range(M):
            bn_w, bn_b, q_w, q_b = per[i]
            g[M + 4 * i + 2] = ops.conv1d_wgrad(dqkv[i], ys[i], rows, 1, 1).view_as(q_w)
            g[M + 4 * i + 3] = ops.col_sum(dqkv[i])
            lead = None
            if i == 0:  # the leader also feeds the regressor directly
                lead = ops.copy_cols(dz[:, :enc0], torch.empty((rows, enc0), device=z.device, dtype=torch.float32))
            dy = _linear_T(dqkv[i], q_w, residual=lead)
            dt, g[M + 4 * i + 0], g[M + 4 * i + 1] = ops.bn_rows_bwd(dy, t[i], stats[i][0], stats[i][1], bn_w, train)
            g[i] = dt if ctx.needs_input_grad[3 + i] else None
        return (None, None, None, *g)


class CrossEntropyFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits2d, labels_f32):
        loss, dl = ops.cross_entropy(logits2d.contiguous(), labels_f32.contiguous(), want_grad=True)
        ctx.dl = dl
        return loss

    @staticmethod
    def backward(ctx, gout):
        return ctx.dl * gout, None


def cross_entropy_loss(outputs, labels):
    """``nn.CrossEntropyLoss(reduction='mean')`` as the trainer applies it (trainer.py:380-383):
    outputs [B,L,C] or [B*L,C]; labels float or long with B*L elements."""
    c = outputs.shape[-1]
    return CrossEntropyFunction.apply(outputs.reshape(-1, c), labels.reshape(-1).float())


class MultimodalMultiheadAttention(nn.Module):
    """Parameter holder: qkv_proj per modality + o_proj (transformer.py:102-131)."""

    def __init__(self, modalities, input_dim, modal_dim, num_heads):
        super().__init__()
        assert modal_dim % num_heads == 0, "Embedding dimension must be 0 modulo number of heads."
        self.modalities, self.embed_dim, self.num_heads = list(modalities), modal_dim, num_heads
        self.head_dim = modal_dim // num_heads
        self.qkv_proj = nn.ModuleDict({m: nn.Linear(input_dim[m], 3 * modal_dim) for m in modalities})
        self.o_proj = nn.Linear(modal_dim * len(modalities), modal_dim * len(modalities))
        for m in modalities:
            nn.init.xavier_uniform_(self.qkv_proj[m].weight)
            self.qkv_proj[m].bias.data.fill_(0)
        nn.init.xavier_uniform_(self.o_proj.weight)
        self.o_proj.bias.data.fill_(0)


class MultiModalEncoderBlock(nn.Module):
    def __init__(self, modalities, input_dim, modal_dim, num_heads, dropout=0.0):
        super().__init__()
        self.self_attn = MultimodalMultiheadAttention(modalities, input_dim, modal_dim, num_heads)
        self.norm1 = nn.LayerNorm(modal_dim * len(modalities))
        self.dropout = nn.Dropout(dropout)  # holder for p; the mask is applied inside the LayerNorm kernel


class MultimodalTransformerEncoder(nn.Module):
    def __init__(self, modalities, input_dim, modal_dim, num_heads, dropout=0.0):
        super().__init__()
        self.layers = MultiModalEncoderBlock(modalities, input_dim, modal_dim, num_heads, dropout)


class LFAN(nn.Module):
    def __init__(self, backbone_settings, output_dim: int, task: str, modality=("frame",), kernel_size=5,
                 example_length=300, tcn_attention=0,
                 tcn_channel={'video': [512, 256, 256, 128], 'cnn_res50': [512, 256, 256, 128],
                              'mfcc': [32, 32, 32, 32], 'vggish': [32, 32, 32, 32], 'logmel': [32, 32, 32, 32]},
                 embedding_dim={'video': 512, 'bert': 768, 'cnn_res50': 512, 'mfcc': 39, 'vggish': 128,
                                'logmel': 128, 'egemaps': 88},
                 encoder_dim={'video': 128, 'bert': 128, 'cnn_res50': 128, 'mfcc': 32, 'vggish': 32, 'logmel': 32,
                              'egemaps': 32},
                 modal_dim=32, num_heads=2, root_dir='', device='cuda', head_hw=5):
        super().__init__()
        assert task in TASKS, task
        self.task, self.output_dim = task, output_dim
        self.backbone_settings, self.root_dir, self.device = backbone_settings, root_dir, device
        self.modality = list(modality)
        self.kernel_size, self.example_length = kernel_size, example_length
        self.tcn_channel, self.tcn_attention = tcn_channel, tcn_attention
        self.embedding_dim, self.encoder_dim = embedding_dim, encoder_dim
        self.outputs = {}
        self.temporal, self.fusion = nn.ModuleDict(), None
        self.num_heads, self.modal_dim = num_heads, modal_dim
        self.final_dim = self.encoder_dim[self.modality[0]] + self.modal_dim * len(self.modality)
        self.spatial = nn.ModuleDict()
        self.bn = nn.ModuleDict()
        self.head_hw = head_hw
        self.dropout_seed = 0  # advanced every training forward; masks are a pure function of it
        self.test_masks = None  # parity tests inject the reference's dropout masks here

    def load_visual_backbone(self, backbone_settings):
        resnet = VisualBackbone(mode='ir', use_pretrained=False, head_hw=self.head_hw)
        state_dict = torch.load(join(self.root_dir, backbone_settings['visual_state_dict'] + ".pth"),
                                map_location='cpu', weights_only=True)
        resnet.load_state_dict(state_dict)
        for param in resnet.parameters():
            param.requires_grad = False
        return resnet

    def load_audio_backbone(self, backbone_settings):
        """model.py:436-448"""
        vggish = AudioBackbone()
        state_dict = torch.load(join(self.root_dir, backbone_settings['audio_state_dict'] + ".pth"), map_location='cpu',
                                weights_only=True)
        vggish.backbone.load_state_dict(state_dict)
        for param in vggish.parameters():
            param.requires_grad = False
        return vggish

    def init(self, load_backbone=True):
        if 'video' in self.modality:
            if load_backbone:
                self.spatial["visual"] = self.load_visual_backbone(self.backbone_settings)
            else:  # synthetic weights arrive through load_state_dict
                self.spatial["visual"] = VisualBackbone(mode='ir', use_pretrained=False, head_hw=self.head_hw)
                for p in self.spatial["visual"].parameters():
                    p.requires_grad = False
        if 'logmel' in self.modality:   # model.py:458-461 (dead in the reference's main.py, parseit.py:329-331, but it is the one
            # place the reference runs VGGish INSIDE forward): weights arrive through load_state_dict / load_audio_backbone
            self.spatial["audio"] = self.load_audio_backbone(self.backbone_settings) if load_backbone else AudioBackbone()
        for modal in self.modality:
            self.temporal[modal] = TemporalConvNet(num_inputs=self.embedding_dim[modal],
                                                   max_length=self.example_length,
                                                   num_channels=self.tcn_channel[modal],
                                                   attention=self.tcn_attention, kernel_size=self.kernel_size,
                                                   dropout=0.1)
            self.bn[modal] = nn.BatchNorm1d(self.tcn_channel[modal][-1])
        self.fusion = MultimodalTransformerEncoder(modalities=self.modality, input_dim=self.encoder_dim,
                                                   modal_dim=self.modal_dim, num_heads=self.num_heads, dropout=0.1)
        self.regressor = nn.Linear(self.final_dim, self.output_dim)

    def forward(self, X):
        """model.py:487-526.  Like the reference, the per-modality stage walks the CALLER's dict in its own key order, the
        fusion walks the model's ``modality`` list, and the caller's dict is overwritten with the per-modality features
        ([B, L, C_m] after TemporalConvNet + BatchNorm1d, model.py:511-515) -- so the keys may come in any order, and a key
        the model was not built for (or a missing one) is a KeyError."""
        for m in X:
            if m not in self.temporal:
                raise KeyError(m)                       # self.temporal[modal], model.py:514
        for m in self.modality:
            if m not in X:
                raise KeyError(m)                       # x[modal] in the fusion, transformer.py:137
        mods = list(self.modality)
        masks = self.test_masks or {}
        if self.training:
            self.dropout_seed += 1
        rows_in, bsz, length = {}, None, None
        for m in X:
            x = X[m]
            if m == "video":
                bsz, length = x.shape[0], x.shape[1]
                # frozen encoder (the reference's default): no autograd graph.  After a gradual release
                # (base/parameter_control.py) the released encoder parameters need the gradient of the embedding.
                released = any(p.requires_grad for p in self.spatial["visual"].parameters())
                with torch.set_grad_enabled(released and torch.is_grad_enabled()):
                    vis = self.spatial["visual"]
                    vis.backbone.dropout_seed = self.dropout_seed
                    emb = vis(x.reshape(-1, *x.shape[2:]), masks.get("head"))
                rows_in[m] = emb
            elif m == "logmel":
                # model.py:500-508: [B, 64 mel bins, L, 96 frames] -> one 96 x 64 example per clip frame -> VGGish
                bsz, height, length, width = x.shape
                with torch.no_grad():
                    rows_in[m] = self.spatial["audio"](x.permute(0, 2, 3, 1).contiguous().view(-1, width, height))
            else:
                bsz, length = x.shape[0], x.shape[2]
                rows_in[m] = x.reshape(bsz * length, x.shape[-1])  # [B,1,L,C] -> rows
        if length != self.example_length:
            raise ValueError(f"clip length {length} != example_length {self.example_length} (model.py:521)")
        t = []
        for i, m in enumerate(mods):
            t.append(self.temporal[m].forward_rows(rows_in[m], bsz, length, masks=(masks.get("tcn") or {}).get(m),
                                                   seed=self.dropout_seed * 16 + i))
        attn = self.fusion.layers.self_attn
        M, H, hd = len(mods), attn.num_heads, attn.head_dim
        fmask = None
        if self.training:
            fmask = masks.get("fusion")
            p = self.fusion.layers.dropout.p
            if fmask is None and p > 0:
                fmask = ops.dropout_mask((bsz * length, M * H * hd), p, self.dropout_seed * 16 + 15, 0, t[0].device)
            elif fmask is not None:
                fmask = fmask.reshape(bsz * length, -1).contiguous()
        ts = list(t)
        for m in mods:
            ts += [self.bn[m].weight, self.bn[m].bias, attn.qkv_proj[m].weight, attn.qkv_proj[m].bias]
        ts += [attn.o_proj.weight, attn.o_proj.bias, self.fusion.layers.norm1.weight, self.fusion.layers.norm1.bias,
               self.regressor.weight, self.regressor.bias]
        buffers = [(self.bn[m].running_mean, self.bn[m].running_var) for m in mods]
        sink = []
        logits = LFANHeadFunction.apply((M, H, hd, self.training, sink), buffers, fmask, *ts)
        if self.training:
            for m in mods:
                self.bn[m].num_batches_tracked += 1
        for m, y in zip(mods, sink):                    # the reference's in-place dict updates (model.py:511-515)
            X[m] = y.view(bsz, length, -1)
        out = logits.view(bsz, self.example_length, -1)
        if self.task == REGRESSION:
            out = torch.tanh(out)
        return out
