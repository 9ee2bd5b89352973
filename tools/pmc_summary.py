"""Sum rocprofv3 --pmc counters per kernel name prefix:  python tools/pmc_summary.py <dir> [substr]"""
import csv
import glob
import sys
from collections import defaultdict

d = sys.argv[1]
sub = sys.argv[2] if len(sys.argv) > 2 else "conv_b3"
acc = defaultdict(lambda: defaultdict(float))
cnt = defaultdict(int)
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if sub in r["Kernel_Name"]:
            k = r["Kernel_Name"][:70]
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
            cnt[(k, r["Counter_Name"])] += 1
for k, v in acc.items():
    print(k)
    for c, x in sorted(v.items()):
        print(f"   {c:32s} {x:16.0f}  ({cnt[(k, c)]} dispatches)")
