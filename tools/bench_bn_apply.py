"""GB/s of the batch-statistics BatchNorm apply pass (encoder_bn.hip) at the IR-50 unit shapes (1024 frames).
   python tools/bench_bn_apply.py [--hw 224] [--narrow fp16|bf16]"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from feature_vs_text_compound_emotion_amd import ops  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--hw", type=int, default=224)
    ap.add_argument("--frames", type=int, default=1024)
    ap.add_argument("--narrow", default=None)
    ap.add_argument("--iters", type=int, default=5)
    a = ap.parse_args()
    dt = {"fp16": torch.float16, "bf16": torch.bfloat16, None: None}[a.narrow]
    shapes = [("stem", a.hw, 64, False), ("s1", a.hw // 2, 64, True), ("s2", a.hw // 4, 128, True),
              ("s3", a.hw // 8, 256, True), ("s4", a.hw // 16, 512, True)]
    for name, hw, c, with_res in shapes:
        n = a.frames
        y = torch.randn(n, hw, hw, c, device="cuda")
        sc, sh, al = torch.rand(c, device="cuda") + 0.5, torch.randn(c, device="cuda"), torch.rand(c, device="cuda")
        if dt is None:
            res = ops.split_bf16(torch.randn(n, hw, hw, c, device="cuda")) if with_res else None
            yin, bytes_per = y, 4 + (4 if with_res else 0) + 4
            fn = lambda: ops.bn_apply_nhwc_b3(yin, sc, sh, alpha=None if with_res else al, res=res, want_stats=True)
        else:
            res = torch.randn(n, hw, hw, c, device="cuda").to(dt) if with_res else None
            yin, bytes_per = y.to(dt), 2 + (2 if with_res else 0) + 2
            fn = lambda: ops.bn_apply_nhwc_n16(yin, sc, sh, dtype=dt, alpha=None if with_res else al, res=res, want_stats=True)
        fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(a.iters):
            fn()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / a.iters
        gb = y.numel() * bytes_per / 1e9
        print(f"{name:5s} {n}x{hw}x{hw}x{c}  {ms:8.3f} ms  {gb / ms * 1e3:8.1f} GB/s  ({gb:.2f} GB)", flush=True)
        del y, res, yin


if __name__ == "__main__":
    main()
