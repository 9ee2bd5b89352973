"""In-kernel timeline of the narrow 16x16-patch conv kernel (tile 71: 64 couts, 4 waves, two blocks per CU).

`build`: copies csrc/ to csrc_exp/ (git-ignored), patches conv_n16_patch.hip so that wave 0 of every block records
s_memrealtime (100 MHz) at its phase boundaries plus HW_ID / XCC_ID into a global buffer, and links csrc_exp/libcer_hip.so.
`run` (GPU): loads THAT library instead of the product's, runs the 64 -> 64 @224x224 layer and prints, per phase, the mean
and the distribution over blocks, and the timeline of one CU.  The product sources are not touched (kernel_source_sha stays).

    python tools/exp_stamp.py build && gpurun -- python tools/exp_stamp.py run --frames 256
"""
import argparse
import ctypes
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "feature_vs_text_compound_emotion_amd")
SRC, EXP = os.path.join(PKG, "csrc"), os.path.join(PKG, "csrc_exp")

HEAD = (
    "namespace cer {\n__device__ unsigned long long *cer_dbg_buf = nullptr;\n"
    "#define STAMP(i) do { if (STAMPED) { __builtin_amdgcn_sched_barrier(0); st[i] = __builtin_amdgcn_s_memrealtime(); "
    "__builtin_amdgcn_sched_barrier(0); } } while (0)\n"
    "#define DUMP() do { if (STAMPED && cer_dbg_buf && tid == 0) { st[7] = __builtin_amdgcn_s_memrealtime(); "
    "unsigned long long *o_ = cer_dbg_buf + 16 + (size_t)blockIdx.x * 10; for (int i_ = 0; i_ < 8; ++i_) o_[i_] = st[i_]; "
    "o_[8] = __builtin_amdgcn_s_getreg((4) | (0 << 6) | (31 << 11)); o_[9] = __builtin_amdgcn_s_getreg((20) | (0 << 6) | (3 << 11)); } } while (0)\n")

PATCHES = [
    ("namespace cer {\n", HEAD),
    ("    constexpr int NW = WP * WC, NT = NW * 64;\n    constexpr int PH = 16,",
     "    constexpr int NW = WP * WC, NT = NW * 64;\n    constexpr bool STAMPED = BN == 64 && XBUFS == 1;\n"
     "    unsigned long long st[8] = {0, 0, 0, 0, 0, 0, 0, 0};\n    STAMP(0);\n    constexpr int PH = 16,"),
    # prologue: 1 = addresses ready, 2 = all DMA issued, 3 = this wave's pieces landed, 4 = everyone's (first step barrier)
    ("    issue_w(0, 0, 0);\n    issue_w(0, 1, 1);\n\n    // window: wave w moves",
     "    STAMP(1);\n    issue_w(0, 0, 0);\n    issue_w(0, 1, 1);\n\n    // window: wave w moves"),
    ("    for (int i = 0; i < XPW; ++i) issue_x(i, 0);\n\n", "    for (int i = 0; i < XPW; ++i) issue_x(i, 0);\n    STAMP(2);\n\n"),
    ("// prologue: the slices and the window\n", "// prologue: the slices and the window\n                    STAMP(3);\n"),
    ("                __builtin_amdgcn_s_barrier();\n            }\n            constexpr int ntap = (tap + 2) % 9, nring = (tap + 2) % 3;\n"
     "            const int ncc = cc + (tap + 2 >= 9 ? 1 : 0);\n            const unsigned char *Wr = smem + (tap % 3) * WSLICE;",
     "                __builtin_amdgcn_s_barrier();\n                if (cc == 0 && tap == 0) STAMP(4);\n"
     "            }\n            constexpr int ntap = (tap + 2) % 9, nring = (tap + 2) % 3;\n"
     "            const int ncc = cc + (tap + 2 >= 9 ? 1 : 0);\n            const unsigned char *Wr = smem + (tap % 3) * WSLICE;"),
    ("(zero fills: they return at once)\n", "(zero fills: they return at once)\n    STAMP(5);\n"),
    # ablations: bit 1 of buf[0] = no MFMAs, bit 2 = no output stores (timing only)
    ("                for (int a = 0; a < TC; ++a) acc[a][b] = mfma_n16<F16>(af[kk][a], bf[g], acc[a][b]);\n            });\n            // issue order",
     "                for (int a = 0; a < TC; ++a) if (!(dbgf & 2u)) acc[a][b] = mfma_n16<F16>(af[kk][a], bf[g], acc[a][b]);\n            });\n            // issue order"),
    ("    const int cin_steps = p.cin_steps;\n\n    // ---- DMA assignment ----",
     "    const int cin_steps = p.cin_steps;\n    const unsigned dbgf = STAMPED && cer_dbg_buf ? (unsigned)cer_dbg_buf[0] : 0u;\n\n    // ---- DMA assignment ----"),
    # direct path (patch kernel: the first occurrence): stores skipped on request, stamp after them, dump before the return
    ("            if constexpr (MODE != EPI_GENERIC) epi_direct_stores<MODE, NARROW, TC, TP>(p, acc, c0 + wc * TC * 16, kg, epx, s1, s2);\n        });\n",
     "            if constexpr (MODE != EPI_GENERIC) { if (!(dbgf & 4u)) epi_direct_stores<MODE, NARROW, TC, TP>(p, acc, c0 + wc * TC * 16, kg, epx, s1, s2); }\n        });\n"
     "        if (dbgf & 1u) { asm volatile(\"s_waitcnt vmcnt(0)\" ::: \"memory\"); }\n        STAMP(6);\n"),
    ("        if (p.stats) epi_direct_stats<TC, WP, BN>(p, s1, s2, reinterpret_cast<float *>(smem_n16p), wp, wc, kg, l15, tid, c0, (size_t)patch);\n        return;\n",
     "        if (p.stats) epi_direct_stats<TC, WP, BN>(p, s1, s2, reinterpret_cast<float *>(smem_n16p), wp, wc, kg, l15, tid, c0, (size_t)patch);\n        DUMP();\n        return;\n"),
]


def build():
    if os.path.isdir(EXP):
        shutil.rmtree(EXP)
    shutil.copytree(SRC, EXP, ignore=shutil.ignore_patterns("*.o", "*.sha", ".build_stamp"))
    path = os.path.join(EXP, "conv_n16_patch.hip")
    s = open(path).read()
    for old, new in PATCHES:
        assert s.count(old) >= 1, old[:80]
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
    h, cin, cout = a.hw, 64, 64
    dt = torch.bfloat16
    x = ops.to_n16(torch.randn(a.frames, h, h, cin, device="cuda"), dt)
    w = ops.to_n16(torch.randn(cout, ops.conv_kpad(3, 3, cin), device="cuda") * 0.02, dt)
    nblk = a.frames * (h // 16) ** 2
    buf = torch.zeros(16 + nblk * 10, dtype=torch.int64, device="cuda")
    buf[0] = a.drain
    assert raw.cer_dbg_set_buf(ctypes.c_void_p(buf.data_ptr())) == 0
    run_ = lambda: ops.conv2d_n16(x, w, 3, 3, stride=1, pad=(1, 1), tile=71, want_stats=bool(a.stats))  # noqa: E731
    for _ in range(2):
        run_()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    run_()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1)
    flops = 2.0 * a.frames * h * h * cout * cin * 9
    print(f"launch {ms:.3f} ms  {flops / ms / 1e9:.0f} TF/s, {nblk} blocks")
    s = buf[16:].view(nblk, 10).cpu().numpy()
    t = s[:, :8].astype(np.float64) * 0.01  # us
    names = ["0->1 kernarg + weight addresses", "1->2 issue 4 + window addresses + issue 11", "2->3 own pieces landed", "3->4 first barrier", "4->5 K loop",
             "5->6 flag load + stores issued", "6->7 statistics + exit"]
    for i, nm in enumerate(names):
        d = t[:, i + 1] - t[:, i]
        print(f"{nm:36s} mean {d.mean():7.2f} us   p10 {np.percentile(d, 10):6.2f}  p50 {np.percentile(d, 50):6.2f}  p90 {np.percentile(d, 90):6.2f}")
    tot = t[:, 7] - t[:, 0]
    print(f"{'block lifetime':36s} mean {tot.mean():7.2f} us   p10 {np.percentile(tot, 10):6.2f}  p50 {np.percentile(tot, 50):6.2f}  p90 {np.percentile(tot, 90):6.2f}")
    hw = s[:, 8]
    cu = (s[:, 9] & 15) * 4096 + ((hw >> 13) & 7) * 512 + ((hw >> 12) & 1) * 256 + ((hw >> 8) & 15) * 16
    ids, cnt = np.unique(cu, return_counts=True)
    print(f"{len(ids)} distinct (xcc, se, sh, cu); blocks per CU min {cnt.min()} max {cnt.max()}")
    span = t[:, 7].max() - t[:, 0].min()
    print(f"kernel span by stamps {span:.1f} us; resident blocks per CU (sum of lifetimes / CUs / span) {tot.sum() / (len(ids) * span):.2f}")
    one = np.where(cu == ids[len(ids) // 2])[0]
    one = one[np.argsort(t[one, 0])]
    base = t[one[0], 0]
    print("timeline of one CU (us from its first block): start, addresses, issued, landed, barrier, K loop, stores, end; simd/wave of wave 0")
    for b in one[4:4 + a.show]:
        r = t[b] - base
        print("  " + "  ".join(f"{v:8.2f}" for v in r) + f"   simd {(hw[b] >> 4) & 3} wave {hw[b] & 15}")


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("cmd", choices=["build", "run"])
    ap.add_argument("--frames", type=int, default=256)
    ap.add_argument("--hw", type=int, default=224)
    ap.add_argument("--drain", type=int, default=0)
    ap.add_argument("--stats", type=int, default=0)
    ap.add_argument("--show", type=int, default=16)
    a = ap.parse_args()
    build() if a.cmd == "build" else run(a)
