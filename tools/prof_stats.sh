#!/bin/bash
# rocprofv3 --kernel-trace --stats of one bench.py configuration (run through gpurun):  tools/prof_stats.sh <tag> <bench args...>
set -e
TAG=$1; shift
OUT=gpurun_out/prof_${TAG}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p "$OUT"
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -o s -- python3 bench.py --no-cpu-baseline --no-alt "$@" > "$OUT/bench_under_rocprof.log" 2>&1
grep "^{\"metric" "$OUT/bench_under_rocprof.log" > "$OUT/bench_under_rocprof.json" || true
rm -f "$OUT"/stats/*kernel_trace.csv "$OUT"/stats/*/*kernel_trace.csv
find "$OUT" -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} "$OUT/kernel_stats.csv"
python3 - "$OUT/kernel_stats.csv" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(int(r["Calls"]) for r in rows)
print("total launches", tot)
for r in rows[:28]:
    print(f'{int(r["Calls"]):6d} {float(r["TotalDurationNs"])/1e6:9.2f} ms {float(r["AverageNs"])/1e3:9.1f} us  {r["Name"][:100]}')
PY
