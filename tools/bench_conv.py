"""Per-shape throughput of the implicit-GEMM conv kernel (algorithmic FLOPs / HIP-event time)."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from feature_vs_text_compound_emotion_amd import _lib  # noqa: E402
if os.environ.get("CER_EXP_LIB"):          # A/B runs against an experimental build (tools/exp_*.py): never set in product use
    _lib.LIB_PATH = os.environ["CER_EXP_LIB"]
from feature_vs_text_compound_emotion_amd import ops  # noqa: E402

# (name, Cin, Cout, H, k, stride) for a 40x40 IR-50; H scales with --hw/40
SHAPES = [
    ("s1 64->64", 64, 64, 40, 3, 1),
    ("s2 64->128", 64, 128, 40, 3, 1),
    ("s2 128->128 s2", 128, 128, 40, 3, 2),
    ("s2 128->128", 128, 128, 20, 3, 1),
    ("s3 128->256", 128, 256, 20, 3, 1),
    ("s3 256->256 s2", 256, 256, 20, 3, 2),
    ("s3 256->256", 256, 256, 10, 3, 1),
    ("s4 256->512", 256, 512, 10, 3, 1),
    ("s4 512->512 s2", 512, 512, 10, 3, 2),
    ("s4 512->512", 512, 512, 5, 3, 1),
]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=1024)
    ap.add_argument("--hw", type=int, default=40)
    ap.add_argument("--tiles", default="0")
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--only", default="")
    ap.add_argument("--b3", action="store_true", help="bf16x3 kernel (split hi/lo operands)")
    ap.add_argument("--n16", choices=["bf16", "fp16"], default=None, help="narrow kernel (one 16-bit plane per operand)")
    ap.add_argument("--stats", action="store_true", help="narrow path: request the batch-statistics rows (the conv2 launches of a training-mode unit)")
    ap.add_argument("--custom", default="", help="cin,cout,h,k,stride[;...]: these layers (h as given) instead of the IR-50 list")
    a = ap.parse_args()
    scale = a.hw / 40
    shapes = SHAPES
    if a.custom:
        scale = 1.0
        shapes = [("%s->%s k%s s%s" % tuple(t.split(",")[i] for i in (0, 1, 3, 4)),) + tuple(int(v) for v in t.split(","))
                  for t in a.custom.split(";")]
    for name, cin, cout, h, k, stride in shapes:
        if a.only and a.only not in name:
            continue
        h = int(h * scale)
        x = torch.randn(a.frames, h, h, cin, device="cuda")
        w = torch.randn(cout, ops.conv_kpad(k, k, cin), device="cuda") * 0.02
        ho = (h + 2 * (k // 2) - k) // stride + 1
        flops = 2.0 * a.frames * ho * ho * cout * cin * k * k
        if a.n16:
            dt = torch.bfloat16 if a.n16 == "bf16" else torch.float16
            xs, ws = ops.to_n16(x, dt), ops.to_n16(w, dt)
            del x
            for tile in [int(t) for t in a.tiles.split(",")]:
                run = lambda: ops.conv2d_n16(xs, ws, k, k, stride=stride, pad=(k // 2, k // 2), tile=tile, want_stats=a.stats)  # noqa: E731
                try:
                    for _ in range(2):
                        run()
                except RuntimeError as err:
                    if "UNSUPPORTED" in str(err) or "status -2" in str(err):
                        print(f"{name:18s} H={h:3d} {a.n16} tile={tile}  unsupported", flush=True)
                        continue
                    raise
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(a.iters):
                    run()
                e1.record()
                torch.cuda.synchronize()
                ms = e0.elapsed_time(e1) / a.iters
                gb = (xs.numel() * 2 + a.frames * ho * ho * cout * 2) / 1e9
                print(f"{name:18s} H={h:3d} {a.n16} tile={tile} {ms:8.3f} ms  {flops / ms / 1e9:7.1f} TFLOP/s  "
                      f"{gb / ms * 1e3:6.0f} GB/s (in+out)", flush=True)
            continue
        if a.b3:
            xs, ws = ops.split_bf16(x), ops.split_bf16(w)
            for tile in [int(t) for t in a.tiles.split(",")]:
                run = lambda: ops.conv2d_b3(xs, ws, k, k, stride=stride, pad=(k // 2, k // 2), tile=tile)  # noqa: E731
                try:
                    for _ in range(2):
                        run()
                except RuntimeError as err:
                    if "UNSUPPORTED" in str(err) or "status -2" in str(err):
                        print(f"{name:18s} H={h:3d} b3 tile={tile}  unsupported", flush=True)
                        continue
                    raise
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(a.iters):
                    run()
                e1.record()
                torch.cuda.synchronize()
                ms = e0.elapsed_time(e1) / a.iters
                print(f"{name:18s} H={h:3d} b3 tile={tile} {ms:8.3f} ms  {flops / ms / 1e9:7.1f} TFLOP/s (effective)", flush=True)
            continue
        for tile in [int(t) for t in a.tiles.split(",")]:
            y = torch.empty(a.frames, ho, ho, cout, device="cuda")
            for _ in range(2):
                ops.conv2d(x, w, k, k, stride=stride, pad=(k // 2, k // 2), tile=tile, out=y)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(a.iters):
                ops.conv2d(x, w, k, k, stride=stride, pad=(k // 2, k // 2), tile=tile, out=y)
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / a.iters
            print(f"{name:18s} H={h:3d} tile={tile} {ms:8.3f} ms  {flops / ms / 1e9:7.1f} TFLOP/s", flush=True)


if __name__ == "__main__":
    main()
