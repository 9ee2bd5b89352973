#!/bin/bash
# Collect the judged profiles of one bench configuration on the GPU box (run through gpurun):
#   tools/profile_round.sh <hw> <tag>
# 1) rocprofv3 --kernel-trace --stats of bench.py  2) separate --pmc FETCH_SIZE / WRITE_SIZE passes (HBM traffic)
set -e
HW=$1; TAG=$2; OUT=gpurun_out/prof_${TAG}_hw${HW}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p "$OUT"
ARGS="bench.py --hw $HW --steps 3 --warmup 1 --no-cpu-baseline --no-alt"
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -o s -- python3 $ARGS > "$OUT/bench_under_rocprof.log" 2>&1
timeout -k 10 500 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/fetch" -o f -- python3 $ARGS > "$OUT/fetch.log" 2>&1
timeout -k 10 500 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/write" -o w -- python3 $ARGS > "$OUT/write.log" 2>&1
python3 tools/collect_traffic.py "$OUT/fetch" "$OUT/write" 4 "$OUT/traffic.json" batch=32 length=32 encoders=on hw=$HW precision=bf16x3
grep "^{\"metric" "$OUT/bench_under_rocprof.log" > "$OUT/bench_under_rocprof.json"
rm -f "$OUT"/stats/*kernel_trace.csv "$OUT"/fetch/*.csv "$OUT"/write/*.csv   # large; the summaries (traffic.json incl. per-kernel per-launch bytes) stay
ls "$OUT" "$OUT/stats"
