"""Build recipe for libcer_hip.so (gfx950 only, in-tree so it travels with gpurun)."""
import hashlib
import os
import subprocess
import sys

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG_DIR, "csrc")
LIB_PATH = os.path.join(PKG_DIR, "libcer_hip.so")
STAMP = os.path.join(PKG_DIR, "csrc", ".build_stamp")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["-O3", "--offload-arch=gfx950", "-fPIC", "-std=c++17", "-Wall", "-Wno-unused-function",
         "-Wno-unused-result"]


def sources():
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hip"))


def _digest():
    h = hashlib.sha256()
    for f in sorted(os.listdir(CSRC)):
        if f.endswith((".hip", ".h")):
            with open(os.path.join(CSRC, f), "rb") as fh:
                h.update(f.encode())
                h.update(fh.read())
    with open(os.path.join(PKG_DIR, "..", "include", "cer_hip.h"), "rb") as fh:
        h.update(fh.read())
    h.update(" ".join(FLAGS).encode())
    return h.hexdigest()


def build(force=False, verbose=True):
    """Compile every .hip translation unit and link libcer_hip.so.  Objects are
    cached per source so that iterating on one kernel recompiles one file."""
    dig = _digest()
    if not force and os.path.exists(LIB_PATH) and os.path.exists(STAMP):
        with open(STAMP) as fh:
            if fh.read().strip() == dig:
                return LIB_PATH
    objs = []
    procs = []
    for src in sources():
        obj = src[:-4] + ".o"
        objs.append(obj)
        with open(src, "rb") as fh:
            sh = hashlib.sha256(fh.read())
        # every header a translation unit may include: all of csrc/*.h plus the public ABI header
        hdrs = sorted(h for h in os.listdir(CSRC) if h.endswith(".h")) + [os.path.join("..", "..", "include", "cer_hip.h")]
        for hdr in hdrs:
            with open(os.path.join(CSRC, hdr), "rb") as fh:
                sh.update(fh.read())
        sh.update(" ".join(FLAGS).encode())
        tag = obj + ".sha"
        if not force and os.path.exists(obj) and os.path.exists(tag) and open(tag).read() == sh.hexdigest():
            continue
        cmd = [HIPCC, *FLAGS, "-c", src, "-o", obj]
        if verbose:
            print("[build]", " ".join(cmd), flush=True)
        procs.append((subprocess.Popen(cmd), tag, sh.hexdigest(), src))
    for p, tag, hx, src in procs:
        if p.wait() != 0:
            raise RuntimeError(f"hipcc failed on {src}")
        with open(tag, "w") as fh:
            fh.write(hx)
    cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB_PATH, *objs]
    if verbose:
        print("[build]", " ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    with open(STAMP, "w") as fh:
        fh.write(dig)
    return LIB_PATH


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    print(LIB_PATH)
