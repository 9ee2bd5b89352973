"""Does the HBM-bound bn_apply pass hide under an MFMA-bound conv launch when both run at once (two streams)?
Serial vs concurrent time of one 256->256 @56x56 window conv (1024 frames) and one bn_apply over a tensor of that size."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from feature_vs_text_compound_emotion_amd import ops  # noqa: E402


def main():
    n, h, c = 1024, 56, 256
    x = ops.split_bf16(torch.randn(n, h, h, c, device="cuda"))
    w = ops.split_bf16(torch.randn(c, 9 * c, device="cuda") * 0.02)
    z = torch.randn(n, h, h, c, device="cuda")
    sc, sh = torch.rand(c, device="cuda") + 0.5, torch.randn(c, device="cuda")
    res = ops.split_bf16(torch.randn(n, h, h, c, device="cuda"))
    conv = lambda: ops.conv2d_b3(x, w, 3, 3, pad=(1, 1), out_f32=True, out_split=False, want_stats=True)  # noqa: E731
    bn = lambda: ops.bn_apply_nhwc_b3(z, sc, sh, res=res, want_stats=True)  # noqa: E731
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    for _ in range(3):
        conv()
        bn()
    torch.cuda.synchronize()

    def timed(fn, iters=10):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        fn(iters)
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / iters

    def serial(iters):
        for _ in range(iters):
            conv()
            bn()

    def only_conv(iters):
        for _ in range(iters):
            conv()

    def only_bn(iters):
        for _ in range(iters):
            bn()

    def both(iters):
        cur = torch.cuda.current_stream()
        s1.wait_stream(cur)
        s2.wait_stream(cur)
        for _ in range(iters):
            with torch.cuda.stream(s1):
                conv()
            with torch.cuda.stream(s2):
                bn()
        cur.wait_stream(s1)
        cur.wait_stream(s2)

    print(f"conv {timed(only_conv):.3f} ms, bn_apply {timed(only_bn):.3f} ms, serial {timed(serial):.3f} ms, "
          f"two streams {timed(both):.3f} ms", flush=True)


if __name__ == "__main__":
    main()
