"""Stride-2 3x3 convs of IR-50: the flat tap-gather kernel vs the window-resident kernel on a space-to-depth input
(algorithmic FLOPs / HIP-event time).  --n16 for the narrow twins."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from feature_vs_text_compound_emotion_amd import ops  # noqa: E402


def timed(fn, iters):
    for _ in range(2):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=1024)
    ap.add_argument("--hw", type=int, default=224, help="input frame size (the four stride-2 layers see hw, hw/2, hw/4, hw/8)")
    ap.add_argument("--iters", type=int, default=5)
    ap.add_argument("--n16", choices=["bf16", "fp16"], default=None, help="narrow kernels (one 16-bit plane per operand)")
    a = ap.parse_args()
    if a.n16:
        dt = torch.bfloat16 if a.n16 == "bf16" else torch.float16
        for i, c in enumerate((128, 256, 512)):
            h = a.hw >> i
            if h % 2:
                continue
            x = ops.to_n16(torch.randn(a.frames, h, h, c, device="cuda"), dt)
            w = ops.to_n16(torch.randn(c, 9 * c, device="cuda") * 0.02, dt)
            xs, ws = ops.space_to_depth(x), ops.pack_s2d_weight(w, c)
            flops = 2.0 * a.frames * (h // 2) ** 2 * c * c * 9
            kw = dict(stride=2, pad=(1, 1), want_stats=True)
            t_flat = timed(lambda: ops.conv2d_n16(x, w, 3, 3, **kw), a.iters)
            t_s2d = timed(lambda: ops.conv2d_n16(xs, ws, 3, 3, x_s2d=True, **kw), a.iters)
            print(f"{c:4d}->{c:<4d} @{h:3d}^2 s2 {a.n16}: flat {t_flat:7.3f} ms {flops / t_flat / 1e9:6.1f} TF/s | "
                  f"s2d {t_s2d:7.3f} ms {flops / t_s2d / 1e9:6.1f} TF/s  x{t_flat / t_s2d:.2f}", flush=True)
            del x, w, xs, ws
        return
    for i, c in enumerate((64, 128, 256, 512)):
        h = a.hw >> i
        if h % 2:
            continue
        x = ops.split_bf16(torch.randn(a.frames, h, h, c, device="cuda"))
        w = ops.split_bf16(torch.randn(c, 9 * c, device="cuda") * 0.02)
        xs, ws = ops.space_to_depth(x), ops.pack_s2d_weight(w, c)
        flops = 2.0 * a.frames * (h // 2) ** 2 * c * c * 9
        kw = dict(stride=2, pad=(1, 1), out_f32=True, out_split=False, want_stats=True)
        t_flat = timed(lambda: ops.conv2d_b3(x, w, 3, 3, **kw), a.iters)
        t_s2d = timed(lambda: ops.conv2d_b3(xs, ws, 3, 3, x_s2d=True, **kw), a.iters)
        print(f"{c:4d}->{c:<4d} @{h:3d}^2 s2: flat {t_flat:7.3f} ms {3 * flops / t_flat / 1e9:6.1f} TF/s(x3) | "
              f"s2d {t_s2d:7.3f} ms {3 * flops / t_s2d / 1e9:6.1f} TF/s(x3)  x{t_flat / t_s2d:.2f}", flush=True)
        del x, w, xs, ws


if __name__ == "__main__":
    main()
