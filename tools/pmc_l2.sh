#!/bin/bash
# L2 hit rate + fabric traffic of single conv layers (run through gpurun):  tools/pmc_l2.sh <tag> <bench_conv args...>
set -e
TAG=$1; shift
OUT=gpurun_out/pmcl2_${TAG}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p "$OUT"
timeout -k 10 300 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --output-format csv -d "$OUT/hit" -o h -- python3 tools/bench_conv.py "$@" > "$OUT/run_hit.log" 2>&1
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/fetch" -o f -- python3 tools/bench_conv.py "$@" > "$OUT/run_fetch.log" 2>&1
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/write" -o w -- python3 tools/bench_conv.py "$@" > "$OUT/run_write.log" 2>&1
python3 - "$OUT" <<'PY' | tee "$OUT/summary.txt"
import csv, glob, sys
from collections import defaultdict
acc = defaultdict(lambda: defaultdict(float)); n = defaultdict(lambda: defaultdict(int))
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "conv_" not in r["Kernel_Name"]: continue
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[k][r["Counter_Name"]] += 1
for k, v in acc.items():
    hit, miss = v.get("TCC_HIT_sum", 0), v.get("TCC_MISS_sum", 0)
    print(k)
    print(f"   L2 hit rate {hit / max(hit + miss, 1):.3f} (hits {hit:.3g}, misses {miss:.3g}, requests {v.get('TCC_REQ_sum', 0):.3g}, {n[k].get('TCC_HIT_sum', 0)} launches)")
    if "FETCH_SIZE" in v:
        print(f"   FETCH_SIZE x2 per launch {2 * v['FETCH_SIZE'] * 1024 / n[k]['FETCH_SIZE'] / 1e9:.3f} GB; WRITE_SIZE per launch {v.get('WRITE_SIZE', 0) * 1024 / max(n[k].get('WRITE_SIZE', 1), 1) / 1e9:.3f} GB")
PY
rm -rf "$OUT"/hit "$OUT"/fetch "$OUT"/write
