"""VGGish audio encoder + log-mel front end on the HIP kernels.

Mirror of the reference's ``VGG`` / ``VGGish`` / ``AudioBackbone`` (models/backbone.py:16-66,
133-145; pre-processing twin abaw5_pre_processing/base/vggish/vggish.py) and of the numpy front
end ``waveform_to_examples`` / ``wavfile_to_examples`` (vggish_input.py:37-98): same state-dict
keys (``features.N`` / ``embeddings.N``), ``forward(x [n,96,64]) -> [n,128]``.

Pipeline: the 1-channel stem conv reads the log-mel patch directly (small-Cin gather path), every
conv fuses bias + ReLU, max-pools are NHWC kernels, and the reference's two transposes before the
flatten are free because the activations already are (H, W, C).
"""
import numpy as np
import torch
from torch import nn

from . import ops

SAMPLE_RATE = 16000
CONV_IDX = (0, 3, 6, 8, 11, 13)
POOL_AFTER = (0, 3, 8, 13)


def _make_layers():
    layers, cin = [], 1
    for v in [64, "M", 128, "M", 256, 256, "M", 512, 512, "M"]:
        if v == "M":
            layers.append(nn.MaxPool2d(kernel_size=2, stride=2))
        else:
            layers += [nn.Conv2d(cin, v, kernel_size=3, padding=1), nn.ReLU(inplace=True)]
            cin = v
    return nn.Sequential(*layers)


def hertz_to_mel(f):
    return 1127.0 * np.log(1.0 + f / 700.0)


def mel_matrix(num_mel_bins=64, num_bins=257, sample_rate=16000, lower=125.0, upper=7500.0):
    """HTK mel weights, DC row zeroed (mel_features.py:134-204) -- a host-side constant."""
    bins_mel = hertz_to_mel(np.linspace(0.0, sample_rate / 2.0, num_bins))
    edges = np.linspace(hertz_to_mel(lower), hertz_to_mel(upper), num_mel_bins + 2)
    w = np.empty((num_bins, num_mel_bins))
    for i in range(num_mel_bins):
        lo, ce, up = edges[i:i + 3]
        w[:, i] = np.maximum(0.0, np.minimum((bins_mel - lo) / (ce - lo), (up - bins_mel) / (up - ce)))
    w[0, :] = 0.0
    return w


def example_starts(num_frames, window_frames, hop_frames):
    """my_frame (mel_features.py:21-49): Python round() -> half to even, fractional hop."""
    n = 1 + int(np.floor((num_frames - window_frames) / hop_frames))
    return [round(hop_frames * i) for i in range(max(n, 0))]


class VGGish(nn.Module):
    def __init__(self):
        super().__init__()
        self.features = _make_layers()
        self.embeddings = nn.Sequential(nn.Linear(512 * 4 * 6, 4096), nn.ReLU(True), nn.Linear(4096, 4096), nn.ReLU(True),
                                        nn.Linear(4096, 128))
        self._packed, self._key = None, None
        self._mel = None
        # "bf16x3": convs 2-6 and the three FCs on the split-bf16 kernels (<= 2^-15 relative per product); "fp32": exact fp32;
        # "bf16" / "fp16": narrow storage (one 16-bit plane per tensor, one MFMA per product, fp32 accumulate) -- what the
        # reference's autocast computes (trainer.py:367) and BASELINE cfg5's "bf16" asks of the whole tri-modal step
        self.precision = "bf16x3"

    def _pack(self):
        key = tuple((p.data_ptr(), p._version) for p in self.parameters())
        if self._packed is None or key != self._key:
            if self.features[0].weight.device.type != "cuda":
                raise RuntimeError("VGGish runs on the HIP kernels only: move the module to a GPU (no CPU fallback)")
            convs = [ops.pack_conv_weight(self.features[i].weight.detach().contiguous()) for i in CONV_IDX]
            self._packed = {"convs": convs, "convs_b3": [None] + [ops.split_bf16(w) for w in convs[1:]],
                            "fc_b3": [ops.split_bf16(self.embeddings[i].weight.detach().contiguous()) for i in (0, 2, 4)]}
            self._key = key
        return self._packed

    def _pack_n16(self, dtype):
        packed = self._pack()
        if packed.get("n16_dtype") != dtype:
            packed["convs_n16"] = [None] + [ops.to_n16(w, dtype) for w in packed["convs"][1:]]
            packed["fc_n16"] = [ops.to_n16(self.embeddings[i].weight.detach().contiguous(), dtype) for i in (0, 2, 4)]
            packed["n16_dtype"] = dtype
        return packed

    def __deepcopy__(self, memo):
        import copy
        packed, self._packed = self._packed, None
        try:
            new = self.__class__.__new__(self.__class__)
            memo[id(self)] = new
            new.__dict__ = copy.deepcopy(self.__dict__, memo)
        finally:
            self._packed = packed
        return new

    def forward(self, x, fs=None):
        """x: [n,96,64] log-mel examples (tensor or numpy, like the reference) -> [n,128]."""
        dev = self.features[0].weight.device
        x = torch.as_tensor(x).to(dev).float().contiguous()
        packed = self._pack()
        n = x.shape[0]
        if self.precision == "bf16x3":
            return self._forward_b3(x, packed, n)
        if self.precision in ("bf16", "fp16"):
            return self._forward_n16(x, n, torch.bfloat16 if self.precision == "bf16" else torch.float16)
        if self.precision != "fp32":
            raise ValueError(f"unknown precision {self.precision!r}")
        packed = packed["convs"]
        y = x.view(n, 1, x.shape[1], x.shape[2])  # NCHW with C = 1
        for j, i in enumerate(CONV_IDX):
            y = ops.conv2d(y, packed[j], 3, 3, pad=(1, 1), bias=self.features[i].bias.detach(), act1=ops.ACT_RELU,
                           x_nchw=(j == 0))
            if i in POOL_AFTER:
                y = ops.maxpool2x2_nhwc(y)
        e = y.view(n, -1)  # (H, W, C) flatten == the reference's transposes + view
        fc = self.embeddings
        split = max(1, min(8, 512 // max(1, (n + 127) // 128 * 32)))
        e = ops.linear(e, fc[0].weight.detach(), bias=fc[0].bias.detach(), act=ops.ACT_RELU, split_k=split)
        e = ops.linear(e, fc[2].weight.detach(), bias=fc[2].bias.detach(), act=ops.ACT_RELU, split_k=split)
        return ops.linear(e, fc[4].weight.detach(), bias=fc[4].bias.detach(), split_k=split)

    def _forward_b3(self, x, packed, n):
        """Layer 1 (Cin = 1) on the fp32 small-Cin kernel, everything else on the bf16x3 kernels.  Max-pooling
        needs the fp32 value, so a conv that feeds a pool writes fp32 and the pooled map is re-split."""
        feats, fc = self.features, self.embeddings
        y = ops.conv2d(x.view(n, 1, x.shape[1], x.shape[2]), packed["convs"][0], 3, 3, pad=(1, 1),
                       bias=feats[0].bias.detach(), act1=ops.ACT_RELU, x_nchw=True)
        cur = ops.split_bf16(ops.maxpool2x2_nhwc(y))
        for j, i in list(enumerate(CONV_IDX))[1:]:
            pooled = i in POOL_AFTER
            r = ops.conv2d_b3(cur, packed["convs_b3"][j], 3, 3, pad=(1, 1), bias=feats[i].bias.detach(), act1=ops.ACT_RELU,
                              out_f32=pooled, out_split=not pooled)
            cur = ops.split_bf16(ops.maxpool2x2_nhwc(r["y"])) if pooled else r["split"]
        k = cur.hi.numel() // n
        e = cur.view(n, 1, 1, k)  # (H, W, C) flatten == the reference's transposes + view
        split = max(1, min(8, 512 // max(1, (n + 127) // 128 * 32)))
        e = ops.conv2d_b3(e, packed["fc_b3"][0], 1, 1, bias=fc[0].bias.detach(), act1=ops.ACT_RELU, split_k=split)["split"]
        e = ops.conv2d_b3(e, packed["fc_b3"][1], 1, 1, bias=fc[2].bias.detach(), act1=ops.ACT_RELU, split_k=split)["split"]
        return ops.conv2d_b3(e, packed["fc_b3"][2], 1, 1, bias=fc[4].bias.detach(), split_k=split, out_f32=True,
                             out_split=False)["y"].view(n, -1)

    def _forward_n16(self, x, n, dtype):
        """Narrow storage: layer 1 (Cin = 1) on the fp32 small-Cin kernel, convs 2-6 and the FCs on the narrow kernels.  A
        conv that feeds a max-pool writes fp32 (pooling wants the un-rounded value) and the pooled map is rounded once."""
        packed = self._pack_n16(dtype)
        feats, fc = self.features, self.embeddings
        y = ops.conv2d(x.view(n, 1, x.shape[1], x.shape[2]), packed["convs"][0], 3, 3, pad=(1, 1),
                       bias=feats[0].bias.detach(), act1=ops.ACT_RELU, x_nchw=True)
        cur = ops.to_n16(ops.maxpool2x2_nhwc(y), dtype)
        for j, i in list(enumerate(CONV_IDX))[1:]:
            pooled = i in POOL_AFTER
            r = ops.conv2d_n16(cur, packed["convs_n16"][j], 3, 3, pad=(1, 1), bias=feats[i].bias.detach(), act1=ops.ACT_RELU,
                               out_f32=pooled, out_n16=not pooled)
            cur = ops.to_n16(ops.maxpool2x2_nhwc(r["y"]), dtype) if pooled else r["n16"]
        k = cur.numel() // n
        e = cur.view(n, 1, 1, k)
        split = max(1, min(8, 512 // max(1, (n + 127) // 128 * 32)))
        e = ops.conv2d_n16(e, packed["fc_n16"][0], 1, 1, bias=fc[0].bias.detach(), act1=ops.ACT_RELU, split_k=split)["n16"]
        e = ops.conv2d_n16(e, packed["fc_n16"][1], 1, 1, bias=fc[2].bias.detach(), act1=ops.ACT_RELU, split_k=split)["n16"]
        return ops.conv2d_n16(e, packed["fc_n16"][2], 1, 1, bias=fc[4].bias.detach(), split_k=split, out_f32=True,
                              out_n16=False)["y"].view(n, -1)

    # ---------------------------------------------------------------- front end
    def wav_int16_to_examples(self, pcm_int16, sample_rate, window_sec=0.96, hop_sec=0.96):
        """wavfile_to_examples for PCM already in memory: pcm [clips, S] (or [S]) int16 at 16 kHz ->
        [clips, n_examples, 96, 64] float32 on the GPU."""
        if sample_rate != SAMPLE_RATE:
            raise ValueError("no resampler on the HIP path: feed 16 kHz PCM (the reference resamples with resampy)")
        dev = self.features[0].weight.device
        pcm = torch.as_tensor(pcm_int16).to(dev)
        if pcm.dim() == 1:
            pcm = pcm[None]
        if self._mel is None or self._mel.device != dev:
            self._mel = torch.from_numpy(mel_matrix()).to(dev).contiguous()
        lm = ops.logmel(pcm.contiguous(), sample_rate, self._mel, 0.01)  # pad = one second of edge samples
        win = int(round(window_sec * 100.0))
        starts = example_starts(lm.shape[1], win, hop_sec * 100.0)
        st = torch.tensor(starts, dtype=torch.int32, device=dev)
        return ops.frame_examples(lm, st, win)


class AudioBackbone(nn.Module):
    def __init__(self):
        super().__init__()
        self.backbone = VGGish()
        for p in self.backbone.parameters():
            p.requires_grad = False

    def forward(self, x, extract_vggish=False):
        return self.backbone(x)
