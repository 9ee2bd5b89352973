"""In-kernel timeline of the persistent narrow patch kernel (tile 79, conv_n16_p64.hip): per wave and patch, s_memrealtime at
the top of the patch, after its 18 half taps (before the vmcnt wait), after the wait, after the block barrier.
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
     "#define STAMP(i) do { __builtin_amdgcn_sched_barrier(0); if (cer_dbg_buf && it < 64 && lane == 0) "
     "cer_dbg_buf[16 + ((size_t)(blockIdx.x * 4 + wave) * 64 + it) * 4 + i] = __builtin_amdgcn_s_memrealtime(); __builtin_amdgcn_sched_barrier(0); } while (0)\n\nnamespace {"),
    ("        unsigned xo[XPW];\n        if (has_next) window_offsets(nxt, xo);", "        STAMP(0);\n        unsigned xo[XPW];\n        if (has_next) window_offsets(nxt, xo);"),
    ("    const int emode = epi_mode(p);\n", "    const int emode = epi_mode(p);\n    const unsigned dbgf = cer_dbg_buf ? (unsigned)cer_dbg_buf[0] : 0u;\n"),
    ("            if constexpr (HAS_PREV && h >= 10) {\n", "            if (HAS_PREV && h >= 10 && !(dbgf & 1u)) {\n"),
    ("                if (has_next) {\n#pragma unroll\n                    for (int i = first; i < first + cnt; ++i) issue_x(i, nxt, xo, cur ^ 1);",
     "                if (has_next && !(dbgf & 2u)) {\n#pragma unroll\n                    for (int i = first; i < first + cnt; ++i) issue_x(i, nxt, xo, cur ^ 1);"),
    ("        // window `nxt` has landed (the stores", "        STAMP(1);\n        // window `nxt` has landed (the stores"),
    ("        __builtin_amdgcn_s_barrier();\n        __builtin_amdgcn_sched_barrier(0);\n        if constexpr (HAS_PREV && STATS) stats_final(it & 1, prv);",
     "        STAMP(2);\n        __builtin_amdgcn_s_barrier();\n        STAMP(3);\n        if constexpr (HAS_PREV && STATS) stats_final(it & 1, prv);"),
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
    buf = torch.zeros(16 + 256 * 4 * NIT * 4, dtype=torch.int64, device="cuda")
    buf[0] = a.flags
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
    t = buf[16:].view(256, 4, NIT, 4).cpu().numpy().astype(np.float64) * 0.01
    ok = t[:, :, 2:NIT - 1, :]                                   # steady state: patches 2 .. 62
    k = ok[..., 1] - ok[..., 0]
    wv = ok[..., 2] - ok[..., 1]
    bar = ok[..., 3] - ok[..., 2]
    nxt = t[:, :, 3:NIT, 0] - t[:, :, 2:NIT - 1, 3]
    per = t[:, :, 3:NIT, 0] - t[:, :, 2:NIT - 1, 0]
    for nm, d in (("18 half taps (+ DMA issue, epilogue of the previous patch)", k), ("vmcnt wait (next window)", wv), ("block barrier", bar),
                  ("barrier -> next patch's top (statistics, loop)", nxt), ("patch period", per)):
        print(f"{nm:62s} mean {d.mean():6.2f} us  p10 {np.percentile(d, 10):6.2f}  p50 {np.percentile(d, 50):6.2f}  p90 {np.percentile(d, 90):6.2f}")
    b = 100
    print("block 100, per wave: top, taps done, window landed, barrier passed (us from patch 2's top of wave 0)")
    base = t[b, 0, 2, 0]
    for it in range(2, 8):
        print("  " + " | ".join(" ".join(f"{v - base:7.2f}" for v in t[b, wv_, it]) for wv_ in range(4)))


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("cmd", choices=["build", "run"])
    ap.add_argument("--frames", type=int, default=256)
    ap.add_argument("--hw", type=int, default=224)
    ap.add_argument("--cout", type=int, default=64)
    ap.add_argument("--stats", type=int, default=0)
    ap.add_argument("--flags", type=int, default=0)
    a = ap.parse_args()
    build() if a.cmd == "build" else run(a)
