"""Throughput of the encoder weight-gradient kernels on the released units' shapes (1024 frames):
   python tools/bench_wgrad.py [--hw 40]"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from feature_vs_text_compound_emotion_amd import ops  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--hw", type=int, default=40)
    ap.add_argument("--frames", type=int, default=1024)
    a = ap.parse_args()
    shapes = [("s1 64->64", 64, 64, a.hw, 3, 1), ("s3 256->256", 256, 256, a.hw // 4, 3, 1), ("s4 256->512", 256, 512, a.hw // 4, 3, 1),
              ("s4 512->512 s2", 512, 512, a.hw // 4, 3, 2), ("s4 512->512", 512, 512, a.hw // 8, 3, 1), ("s4 256->512 1x1 s2", 256, 512, a.hw // 4, 1, 2)]
    for name, cin, cout, h, k, s in shapes:
        ho = (h + 2 * (k // 2) - k) // s + 1
        x = torch.randn(a.frames, h, h, cin, device="cuda")
        dz = torch.randn(a.frames, ho, ho, cout, device="cuda")
        xs, dzs = ops.split_bf16(x), ops.split_bf16(dz)
        flops = 2.0 * a.frames * ho * ho * cout * cin * k * k
        for label, fn in (("fp32 kernel", lambda: ops.conv2d_wgrad(dz, x, k, k, stride=s, pad=(k // 2, k // 2))),
                          ("b3, fp32 in", lambda: ops.conv2d_wgrad(dz, x, k, k, stride=s, pad=(k // 2, k // 2), b3=True)),
                          ("b3, split in", lambda: ops.conv2d_wgrad(dzs, xs, k, k, stride=s, pad=(k // 2, k // 2), b3=True))):
            fn()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                fn()
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / 5
            print(f"{name:20s} H={h:3d} {label:13s} {ms:8.3f} ms  {flops / ms / 1e9:7.1f} TFLOP/s", flush=True)


if __name__ == "__main__":
    main()
