"""Oracle: the reference's video-frame input transform (test infrastructure only).

Follows /root/reference/base/dataset.py:487-508 and base/transforms3D.py: uint8 frames
[L,256,256,3] -> GroupScale(48) (torchvision Resize == PIL ``Image.resize((48,48), BILINEAR)``:
antialiased triangle filter evaluated in Pillow's 8-bit fixed-point arithmetic, horizontal pass
then vertical pass with a uint8 intermediate) -> GroupRandomCrop(48,40) / GroupCenterCrop(40) ->
GroupRandomHorizontalFlip -> ToTorchFormatTensor (/255, CHW) -> Normalize(0.5, 0.5).

The resize is a restatement of Pillow's ``src/libImaging/Resample.c`` (third-party, not vendored
in the reference; Pillow 12.2.0 in the build container) and is pinned BIT-EXACTLY against
``PIL.Image.resize`` by tools/gen_golden_frames.py.
"""
import math

import numpy as np

PRECISION_BITS = 32 - 8 - 2


def precompute_coeffs(in_size, out_size, support=1.0):
    """(bounds [out,2] = (xmin, count), integer coefficients [out, ksize]) for the triangle filter."""
    scale = float(np.float32(in_size) - np.float32(0)) / out_size
    filterscale = max(scale, 1.0)
    sup = support * filterscale
    ksize = int(math.ceil(sup)) * 2 + 1
    bounds = np.zeros((out_size, 2), dtype=np.int32)
    kk = np.zeros((out_size, ksize), dtype=np.int32)
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        ss = 1.0 / filterscale
        xmin = max(int(center - sup + 0.5), 0)
        xmax = min(int(center + sup + 0.5), in_size) - xmin
        w = np.array([max(0.0, 1.0 - abs((x + xmin - center + 0.5) * ss)) for x in range(xmax)])
        ww = w.sum()
        if ww != 0.0:
            w = w / ww
        for x in range(xmax):
            v = w[x] * (1 << PRECISION_BITS)
            kk[xx, x] = int(-0.5 + v) if w[x] < 0 else int(0.5 + v)
        bounds[xx] = (xmin, xmax)
    return bounds, kk


def _resample_axis(img, bounds, kk):
    """img [A, in, C] uint8 -> [A, out, C] uint8 along axis 1 (Pillow's 8bpc inner loop)."""
    out = np.empty((img.shape[0], bounds.shape[0], img.shape[2]), dtype=np.uint8)
    src = img.astype(np.int64)
    for xx in range(bounds.shape[0]):
        xmin, n = bounds[xx]
        acc = (1 << (PRECISION_BITS - 1)) + np.tensordot(src[:, xmin:xmin + n, :], kk[xx, :n].astype(np.int64), axes=([1], [0]))
        out[:, xx, :] = np.clip(acc >> PRECISION_BITS, 0, 255).astype(np.uint8)
    return out


def resize_bilinear_u8(frame, out_size):
    """[H,W,C] uint8 -> [out,out,C] uint8, horizontal pass first (like Pillow)."""
    bh, kh = precompute_coeffs(frame.shape[1], out_size)
    bv, kv = precompute_coeffs(frame.shape[0], out_size)
    tmp = _resample_axis(frame, bh, kh)                       # along W
    return _resample_axis(tmp.transpose(1, 0, 2), bv, kv).transpose(1, 0, 2)  # along H


def center_crop_offset(size, crop):
    return int(round((size - crop) / 2.0))


def frames_transform(frames_u8, size=48, crop=40, x1=None, y1=None, flip=False):
    """[L,H,W,3] uint8 -> [L,3,crop,crop] float32 in [-1,1].  x1/y1 None -> centre crop."""
    if x1 is None:
        x1 = y1 = center_crop_offset(size, crop)
    out = []
    for f in frames_u8:
        r = resize_bilinear_u8(f, size)[y1:y1 + crop, x1:x1 + crop]
        if flip:
            r = r[:, ::-1]
        out.append(((r.astype(np.float32) / np.float32(255.0)) - np.float32(0.5)) / np.float32(0.5))
    return np.stack(out).transpose(0, 3, 1, 2).copy()
