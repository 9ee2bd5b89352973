"""MFMA utilisation of the bf16x3 conv kernels from one rocprofv3 --pmc pass over bench.py:
    python tools/collect_mfma_util.py <pmc_dir> <out.json>
SQ_BUSY_CYCLES is summed over the 32 shader engines (cycles); SQ_VALU_MFMA_BUSY_CYCLES over the 1024 SIMDs (cycles);
SQ_WAVE_CYCLES / SQ_WAIT_* are in units of 4 cycles."""
import csv
import glob
import json
import sys
from collections import defaultdict

d, out = sys.argv[1], sys.argv[2]
acc = defaultdict(lambda: defaultdict(float))
n = defaultdict(int)
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "conv_b3" in r["Kernel_Name"]:
            k = r["Kernel_Name"].split("(")[0].replace("void ", "")
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
            if r["Counter_Name"] == "SQ_BUSY_CYCLES":
                n[k] += 1
res = {}
for k, v in acc.items():
    cycles = v["SQ_BUSY_CYCLES"] / 32.0            # wall cycles summed over the kernel's launches
    res[k] = {"launches": n[k], "gpu_cycles_per_launch": cycles / max(n[k], 1),
              "mfma_busy_frac_of_simd_cycles": v["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024.0 * cycles),
              "wave_cycles_waiting_frac": v["SQ_WAIT_ANY"] / v["SQ_WAVE_CYCLES"] if v.get("SQ_WAVE_CYCLES") else None,
              "wave_cycles_issue_stalled_frac": v["SQ_WAIT_INST_ANY"] / v["SQ_WAVE_CYCLES"] if v.get("SQ_WAVE_CYCLES") else None}
json.dump(res, open(out, "w"), indent=1)
print(json.dumps(res, indent=1))
