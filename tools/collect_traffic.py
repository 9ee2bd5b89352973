"""Reduce two rocprofv3 --pmc runs of bench.py (FETCH_SIZE and WRITE_SIZE, separate passes as
MI355X_MICROARCH.md prescribes) to HBM bytes per step of cer::conv_igemm_kernel.

gfx950 corrections (MI355X_MICROARCH.md, HBM section): FETCH_SIZE counts 64 B per 128-B request on wide
coalesced reads -> doubled; WRITE_SIZE is exact for 16-B-per-lane stores; both are in KiB.

    python tools/collect_traffic.py <fetch_dir> <write_dir> <steps_total> <out.json> [key=value ...]
"""
import csv
import glob
import json
import sys


PER_KERNEL = {}


def total(d, counter):
    f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)[0]
    s, n = 0.0, 0
    for r in csv.DictReader(open(f)):
        if any(k in r["Kernel_Name"] for k in ("conv_igemm_kernel", "conv_b3", "conv_n16", "stem_conv3x3")) and \
                r["Counter_Name"] == counter:
            s += float(r["Counter_Value"])
            n += 1
            k = PER_KERNEL.setdefault(r["Kernel_Name"].split("(")[0].replace("void ", ""), {})
            k[counter] = k.get(counter, 0.0) + float(r["Counter_Value"])
            k[counter + "_launches"] = k.get(counter + "_launches", 0) + 1
    return s, n


def main():
    fetch_dir, write_dir, steps, out = sys.argv[1], sys.argv[2], int(sys.argv[3]), sys.argv[4]
    extra = dict(kv.split("=", 1) for kv in sys.argv[5:])
    import os
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    sys.modules.setdefault("triton", None)
    from bench import kernel_source_sha
    extra["kernel_source_sha"] = kernel_source_sha()   # bench.py reports a profile only for the code it was taken on
    f, nf = total(fetch_dir, "FETCH_SIZE")
    w, nw = total(write_dir, "WRITE_SIZE")
    res = {"kernel": "all conv launches of one step (conv_b3_*, conv_n16_*, conv_igemm, stem_conv3x3)", "steps_profiled": steps,
           "launches_per_step": nf / steps,
           "fetch_bytes_per_step": 2.0 * f * 1024 / steps, "write_bytes_per_step": w * 1024 / steps,
           "hbm_bytes_per_step": (2.0 * f + w) * 1024 / steps,
           "raw": {"FETCH_SIZE_KiB_sum": f, "WRITE_SIZE_KiB_sum": w, "dispatches": [nf, nw]},
           "correction": "FETCH_SIZE x2 (gfx950 counts 64 B per 128-B request), WRITE_SIZE x1, KiB -> bytes",
           "per_kernel": {k: {"launches": v.get("FETCH_SIZE_launches", 0),
                              "hbm_bytes_per_launch": (2.0 * v.get("FETCH_SIZE", 0.0) / max(v.get("FETCH_SIZE_launches", 1), 1) +
                                                       v.get("WRITE_SIZE", 0.0) / max(v.get("WRITE_SIZE_launches", 1), 1)) * 1024}
                          for k, v in PER_KERNEL.items()}, **extra}
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps(res))


if __name__ == "__main__":
    main()
