"""Eval-mode (inference) throughput of the IR-50 encoder on the bf16x3 kernels: frames/s and effective TFLOP/s.
Trainer.inference (reference trainer.py:436-523) runs the model in eval(): every BatchNorm is folded into the convs."""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.modules.setdefault("triton", None)
from bench import ir50_forward_flops  # noqa: E402
from feature_vs_text_compound_emotion_amd import synth  # noqa: E402
from feature_vs_text_compound_emotion_amd.visual_backbone import VisualBackbone  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--hw", type=int, default=224)
    ap.add_argument("--frames", type=int, default=1024)
    ap.add_argument("--iters", type=int, default=5)
    a = ap.parse_args()
    vsd = synth.make_state_dict(synth.visual_backbone_spec("", a.hw // 8), seed=0)
    vb = VisualBackbone(use_pretrained=False, head_hw=a.hw // 8)
    vb.load_state_dict(vsd)
    vb = vb.cuda().eval()
    x = torch.randn(a.frames, 3, a.hw, a.hw, device="cuda")
    with torch.no_grad():
        for _ in range(2):
            vb(x)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(a.iters):
            vb(x)
        e1.record()
        torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / a.iters
    fl = ir50_forward_flops(a.hw) * a.frames
    print(json.dumps({"mode": "eval", "hw": a.hw, "frames": a.frames, "ms": ms, "frames_per_s": a.frames / ms * 1e3,
                      "effective_tflops": fl / ms / 1e9, "frac_of_833": fl / ms / 1e9 / 833.33}))


if __name__ == "__main__":
    main()
