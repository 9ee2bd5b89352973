"""In-kernel timeline of the persistent narrow patch kernel (tile 79, conv_n16_p64.hip): per wave and phase, s_memrealtime at the
top of the phase, after its work (K phase: 18 half taps; E phase: window DMA, epilogue, the vmcnt wait), after the block barrier.
    python tools/exp_stamp_p64.py build && gpurun -- python tools/exp_stamp_p64.py run"""
import argparse
import ctypes
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "feature_vs_text_compound_emotion_amd")
SRC, EXP = os.path.join(PKG, "csrc"), os.path.join(PKG, "csrc_exp")
NIT = 64

PATCHES = [
    ("namespace cer {\n\nnamespace {", "namespace cer {\n__device__ unsigned long long *cer_dbg_buf = nullptr;\n"
     "#define STAMP(i) do { __builtin_amdgcn_sched_barrier(0); if (cer_dbg_buf && ph < 64 && lane == 0) "
     "cer_dbg_buf[16 + ((size_t)(blockIdx.x * 8 + wave) * 64 + ph) * 4 + i] = __builtin_amdgcn_s_memrealtime(); __builtin_amdgcn_sched_barrier(0); } while (0)\n\nnamespace {"),
    ("            const bool kph = ((ph ^ grp) & 1) == 0;\n", "            STAMP(0);\n            const bool kph = ((ph ^ grp) & 1) == 0;\n"),
    ("            __builtin_amdgcn_sched_barrier(0);\n            __builtin_amdgcn_s_barrier();\n            __builtin_amdgcn_sched_barrier(0);\n        }\n        if constexpr (STATS) {\n            // the block's totals",
     "            STAMP(1);\n            __builtin_amdgcn_s_barrier();\n            STAMP(2);\n        }\n        if constexpr (STATS) {\n            // the block's totals"),
]


def build():
    if os.path.isdir(EXP):
        shutil.rmtree(EXP)
    shutil.copytree(SRC, EXP, ignore=shutil.ignore_patterns("*.o", "*.sha", ".build_stamp"))
    path = os.path.join(EXP, "conv_n16_p64.hip")
    s = open(path).read()
    for old, new in PATCHES:
        assert s.count(old) == 1, old[:80]
        s = s.replace(old, new, 1)
    s += ('\nextern "C" int cer_dbg_set_buf(void *p) { return (int)hipMemcpyToSymbol(HIP_SYMBOL(cer::cer_dbg_buf), &p, sizeof(p)); }\n')
    open(path, "w").write(s)
    flags = ["-O3", "--offload-arch=gfx950", "-fPIC", "-std=c++17", "-Wno-unused-function", "-Wno-unused-result"]
    objs, procs = [], []
    for f in sorted(os.listdir(EXP)):
        if f.endswith(".hip"):
            obj = os.path.join(EXP, f[:-4] + ".o")
            objs.append(obj)
            procs.append(subprocess.Popen(["/opt/rocm/bin/hipcc", *flags, "-c", os.path.join(EXP, f), "-o", obj]))
    for p in procs:
        assert p.wait() == 0
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", os.path.join(EXP, "libcer_hip.so"), *objs])
    print(os.path.join(EXP, "libcer_hip.so"))


def run(a):
    sys.path.insert(0, ROOT)
    from feature_vs_text_compound_emotion_amd import _lib
    _lib.LIB_PATH = os.path.join(EXP, "libcer_hip.so")
    import numpy as np
    import torch
    from feature_vs_text_compound_emotion_amd import ops
    _lib.load()
    raw = ctypes.CDLL(_lib.LIB_PATH)
    h, cin, cout = a.hw, 64, a.cout
    dt = torch.bfloat16
    x = ops.to_n16(torch.randn(a.frames, h, h, cin, device="cuda"), dt)
    w = ops.to_n16(torch.randn(cout, ops.conv_kpad(3, 3, cin), device="cuda") * 0.02, dt)
    buf = torch.zeros(16 + 256 * 8 * NIT * 4, dtype=torch.int64, device="cuda")
    assert raw.cer_dbg_set_buf(ctypes.c_void_p(buf.data_ptr())) == 0
    run_ = lambda: ops.conv2d_n16(x, w, 3, 3, stride=1, pad=(1, 1), tile=79, want_stats=bool(a.stats))  # noqa: E731
    for _ in range(2):
        run_()
    torch.cuda.synchronize()
    buf[16:] = 0
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    run_()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1)
    flops = 2.0 * a.frames * h * h * cout * cin * 9
    print(f"launch {ms:.3f} ms  {flops / ms / 1e9:.0f} TF/s")
    t = buf[16:].view(256, 8, NIT, 4).cpu().numpy().astype(np.float64) * 0.01     # [block][wave][phase][stamp] in us
    ph = np.arange(8, NIT - 2)                                   # steady state
    for g in (0, 1):
        w = t[:, 4 * g:4 * g + 4]
        kph = ph[(ph % 2) == g]                                  # the group's K phases; its E phases follow
        eph = kph + 1
        kwork = w[:, :, kph, 1] - w[:, :, kph, 0]
        ework = w[:, :, eph, 1] - w[:, :, eph, 0]
        kbar = w[:, :, kph, 2] - w[:, :, kph, 1]
        ebar = w[:, :, eph, 2] - w[:, :, eph, 1]
        print(f"group {g}: K phase (18 half taps) {kwork.mean():5.2f} us, then barrier wait {kbar.mean():5.2f};  "
              f"E phase (window DMA + epilogue + vmcnt) {ework.mean():5.2f} us, then barrier wait {ebar.mean():5.2f}")
    per = t[:, 0, 10:NIT - 2, 0] - t[:, 0, 9:NIT - 3, 0]
    print(f"phase period {per.mean():5.2f} us = one patch per CU (three stamps per phase cost ~0.1 us each)")


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("cmd", choices=["build", "run"])
    ap.add_argument("--frames", type=int, default=256)
    ap.add_argument("--hw", type=int, default=224)
    ap.add_argument("--cout", type=int, default=64)
    ap.add_argument("--stats", type=int, default=0)
    a = ap.parse_args()
    build() if a.cmd == "build" else run(a)
