"""Per-kernel summary of a rocprofv3 --pmc pass (SQ_* + GRBM_GUI_ACTIVE) over tools/bench_conv.py.
SQ_BUSY_CYCLES sums over the 32 shader engines (cycles); SQ_VALU_MFMA_BUSY_CYCLES over the 1024 SIMDs (cycles);
SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* are in units of 4 cycles; GRBM_GUI_ACTIVE sums over the 8 XCDs."""
import csv
import glob
import sys
from collections import defaultdict

acc = defaultdict(lambda: defaultdict(float))
n = defaultdict(int)
dur = defaultdict(float)
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if ("conv_" not in r["Kernel_Name"] and "wgrad" not in r["Kernel_Name"]) or "pack_conv" in r["Kernel_Name"]:
            continue
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Counter_Name"] == "SQ_BUSY_CYCLES":
            n[k] += 1
            if "Start_Timestamp" in r and "End_Timestamp" in r:
                dur[k] += float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
for k, v in acc.items():
    cyc = v["SQ_BUSY_CYCLES"] / 32.0
    wc = v["SQ_WAVE_CYCLES"]
    line = f"{k}\n   launches {n[k]}  gpu cycles/launch {cyc / max(n[k], 1):.0f}"
    if dur[k]:
        line += f"  avg {dur[k] / n[k] / 1e3:.1f} us  SQ clock {cyc / dur[k]:.2f} GHz  GRBM clock {v['GRBM_GUI_ACTIVE'] / 8 / dur[k]:.2f} GHz"
    line += (f"\n   MFMA busy {v['SQ_VALU_MFMA_BUSY_CYCLES'] / (1024.0 * cyc):.3f} of SIMD cycles;  wave cycles: waiting "
             f"{v['SQ_WAIT_ANY'] / wc:.3f}, issue-stalled {v['SQ_WAIT_INST_ANY'] / wc:.3f} (of which LDS {v['SQ_WAIT_INST_LDS'] / wc:.3f}), "
             f"active {v['SQ_ACTIVE_INST_ANY'] / wc:.3f};  LDS bank-conflict cycles / wave-cycle {v['SQ_LDS_BANK_CONFLICT'] / (4 * wc):.4f}"
             f";  resident waves/SIMD {4 * wc / (1024.0 * cyc):.2f}")
    print(line)
