"""On-GPU video-frame input transform (the reference Dataset's ``get_3D_transforms``).

Mirrors /root/reference/base/dataset.py:487-508 + base/transforms3D.py: ``GroupScale(48)`` ->
``GroupRandomCrop(48, 40)`` / ``GroupCenterCrop(40)`` -> ``GroupRandomHorizontalFlip`` -> ``Stack`` ->
``ToTorchFormatTensor`` -> ``GroupNormalize(.5, .5)``, as ONE HIP kernel (csrc/frames.hip) over uint8
frames that are already resident in HBM.  The resize is bit-identical to ``PIL.Image.resize(BILINEAR)``
(what torchvision's Resize runs for a PIL image): the host computes Pillow's integer coefficient tables
(``precompute_coeffs`` + ``normalize_coeffs_8bpc`` of Pillow's Resample.c, third-party, restated) once per
geometry and the kernel does the two fixed-point passes.
"""
import math
import random

import numpy as np
import torch

from . import _lib
from ._lib import check, current_stream, ptr

PRECISION_BITS = 32 - 8 - 2


def resample_tables(in_size, out_size):
    """Pillow's bilinear (triangle, support 1) tables: bounds int32 [out,2], coefficients int32 [out,ksize]."""
    scale = float(np.float32(in_size)) / out_size      # the box edges are C floats in Pillow
    filterscale = max(scale, 1.0)
    support = 1.0 * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    bounds = np.zeros((out_size, 2), np.int32)
    coef = np.zeros((out_size, ksize), np.int32)
    inv = 1.0 / filterscale
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        xmin = max(int(center - support + 0.5), 0)
        cnt = min(int(center + support + 0.5), in_size) - xmin
        w = [max(0.0, 1.0 - abs((x + xmin - center + 0.5) * inv)) for x in range(cnt)]
        total = 0.0
        for v in w:
            total += v
        for x in range(cnt):
            v = (w[x] / total if total != 0.0 else w[x]) * (1 << PRECISION_BITS)
            coef[xx, x] = int(v - 0.5) if v < 0 else int(v + 0.5)
        bounds[xx] = (xmin, cnt)
    return bounds, coef


class FrameTransform:
    """``FrameTransform(size=48, crop=40, train=True)(frames_u8[B,L,H,W,3]) -> float32 [B,L,3,crop,crop]``.

    train=True draws one (x1, y1) crop offset and one flip per clip with Python's ``random`` in the
    reference's call order (transforms3D.py:52-53,80); train=False uses GroupCenterCrop's offset.
    """

    def __init__(self, size=48, crop=40, train=True, mean=0.5, std=0.5):
        self.size, self.crop, self.train, self.mean, self.std = size, crop, train, float(mean), float(std)
        self._tables = {}

    def _get_tables(self, h, w, device):
        key = (h, w, str(device))
        if key not in self._tables:
            hb, hk = resample_tables(w, self.size)
            vb, vk = resample_tables(h, self.size)
            band = _lib.load().cer_frames_band_rows()
            # the largest input-row span of any band, over every crop offset
            span = 0
            for y0 in range(self.size):
                y1 = min(self.size, y0 + band) - 1
                span = max(span, int(vb[y1, 0] + vb[y1, 1] - vb[y0, 0]))
            dev = [torch.from_numpy(a).to(device) for a in (hb, hk, vb, vk)]
            self._tables[key] = (dev, hk.shape[1], vk.shape[1], span)
        return self._tables[key]

    def draw(self, n_clips, rng=None):
        """(x1, y1, flip) per clip -- the reference's random calls, in its order.  ``rng``: a ``random.Random`` instance to
        draw from (the prefetcher's own stream); default the global ``random`` module the reference uses."""
        rng = random if rng is None else rng
        out = np.zeros((n_clips, 3), np.int32)
        for i in range(n_clips):
            if self.train:
                out[i, 0] = rng.randint(0, self.size - self.crop)
                out[i, 1] = rng.randint(0, self.size - self.crop)
                out[i, 2] = 1 if rng.random() < 0.5 else 0
            else:
                out[i, 0] = out[i, 1] = int(round((self.size - self.crop) / 2.0))
        return out

    def __call__(self, frames, crop_xyf=None, return_u8=False):
        if not frames.is_cuda or frames.dtype != torch.uint8:
            raise RuntimeError("FrameTransform needs a uint8 CUDA tensor [B,L,H,W,3] (there is no CPU fallback)")
        if frames.dim() == 4:
            frames = frames.unsqueeze(0)
        b, l, h, w, c = frames.shape
        if c != 3:
            raise RuntimeError("FrameTransform: frames must be RGB, channels last")
        frames = frames.contiguous()
        if crop_xyf is None:
            crop_xyf = self.draw(b)
        crop_xyf = np.ascontiguousarray(crop_xyf, dtype=np.int32).reshape(b, 3)
        if (crop_xyf[:, :2] < 0).any() or (crop_xyf[:, :2] + self.crop > self.size).any():
            raise RuntimeError("FrameTransform: crop offset outside the resized image")
        (hb, hk, vb, vk), hks, vks, span = self._get_tables(h, w, frames.device)
        cx = torch.from_numpy(crop_xyf).to(frames.device)
        out = torch.empty((b, l, 3, self.crop, self.crop), dtype=torch.float32, device=frames.device)
        u8 = torch.empty((b, l, self.crop, self.crop, 3), dtype=torch.uint8, device=frames.device) if return_u8 else None
        lib = _lib.load()
        n = b * l
        step = 65535 // l * l if l <= 65535 else 0
        if step == 0:
            raise RuntimeError("FrameTransform: clip longer than 65535 frames")
        flat_in, flat_out = frames.view(n, h, w, 3), out.view(n, 3, self.crop, self.crop)
        flat_u8 = u8.view(n, self.crop, self.crop, 3) if return_u8 else None
        for s in range(0, n, step):
            e = min(n, s + step)
            check(lib.cer_frames_transform(ptr(flat_in[s:e]), e - s, h, w, ptr(hb), ptr(hk), hks, ptr(vb), ptr(vk), vks,
                                           self.size, self.crop, ptr(cx[s // l:]), l, span, self.mean, self.std,
                                           ptr(flat_out[s:e]), ptr(flat_u8[s:e]) if return_u8 else None,
                                           current_stream()), "cer_frames_transform")
        return (out, u8) if return_u8 else out
