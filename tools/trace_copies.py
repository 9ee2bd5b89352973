"""Where do the device-to-device copies and stock torch kernels of a step come from?  (launch-count work, 40x40 recipe)
   python tools/trace_copies.py [bench args]   -> per call site: aten op, count per step"""
import collections
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.modules.setdefault("triton", None)
import torch  # noqa: E402

import bench  # noqa: E402


def main():
    prec = sys.argv[1] if len(sys.argv) > 1 else "fp16"
    enc = sys.argv[2] if len(sys.argv) > 2 else "off"
    cfg = {"model": "LFAN", "modalities": bench.ALL_MODS, "hw": 40, "batch": 32, "length": 32, "n_cls": 7, "precision": prec,
           "release": 0, "encoders": enc, "act_mem": "raw"}
    wl = bench.Workload(cfg, 0, 1, torch.device("cuda", 0))
    for _ in range(2):
        wl.step()
    torch.cuda.synchronize()
    sites = collections.Counter()
    orig = {}

    def wrap(name):
        fn = getattr(torch.Tensor, name)
        orig[name] = fn

        def inner(self, *a, **k):
            if self.is_cuda:
                import traceback
                fr = [f for f in traceback.extract_stack()[:-1] if "feature_vs_text" in f.filename or "bench.py" in f.filename]
                where = " <- ".join(f"{os.path.basename(f.filename)}:{f.lineno}" for f in fr[-3:])
                sites[(name, where)] += 1
            return fn(self, *a, **k)
        setattr(torch.Tensor, name, inner)
    for n in ("clone", "copy_", "contiguous", "add_", "__iadd__", "__add__", "add", "to", "float", "mul_", "zero_"):
        wrap(n)
    from torch.profiler import ProfilerActivity, profile
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
        wl.step()
        torch.cuda.synchronize()
    for n, fn in orig.items():
        setattr(torch.Tensor, n, fn)
    print("---- python-level tensor methods on GPU tensors (one step)")
    for (name, where), c in sites.most_common(40):
        print(f"{c:5d}  {name:12s} {where}")
    print("---- device activities by name (one step)")
    ev = collections.Counter()
    for e in prof.events():
        if e.device_type == torch.autograd.DeviceType.CUDA:
            ev[e.name[:90]] += 1
    for n, c in ev.most_common(25):
        print(f"{c:5d}  {n}")
    print("total device activities:", sum(ev.values()))
    print("---- aten ops that launch copies")
    ops = collections.Counter(e.name for e in prof.events() if e.name.startswith("aten::") and e.device_type != torch.autograd.DeviceType.CUDA)
    for n, c in ops.most_common(25):
        print(f"{c:5d}  {n}")


if __name__ == "__main__":
    main()
