#!/bin/bash
# Collect the judged profiles of one bench configuration on the GPU box (run through gpurun):
#   tools/profile_round.sh <tag> <precision> <hw> <length> [extra bench args]
# 1) rocprofv3 --kernel-trace --stats of bench.py  2) separate --pmc FETCH_SIZE / WRITE_SIZE passes (HBM traffic, per the
# guide: own passes, no trace domains)  3) one SQ pass for MFMA utilisation.  Summaries land in gpurun_out/prof_<tag>/.
set -e
TAG=$1; PREC=$2; HW=$3; LEN=$4; shift 4
OUT=gpurun_out/prof_${TAG}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p "$OUT"
ARGS="bench.py --hw $HW --length $LEN --precision $PREC --steps 3 --warmup 1 --no-cpu-baseline --no-alt $*"
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -o s -- python3 $ARGS > "$OUT/bench_under_rocprof.log" 2>&1
timeout -k 10 500 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/fetch" -o f -- python3 $ARGS > "$OUT/fetch.log" 2>&1
timeout -k 10 500 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/write" -o w -- python3 $ARGS > "$OUT/write.log" 2>&1
timeout -k 10 500 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE --output-format csv -d "$OUT/sq" -o q -- python3 $ARGS > "$OUT/sq.log" 2>&1
python3 tools/collect_traffic.py "$OUT/fetch" "$OUT/write" 4 "$OUT/traffic.json" batch=32 length=$LEN encoders=on hw=$HW precision=$PREC > /dev/null
python3 tools/pmc_conv_summary.py "$OUT/sq" > "$OUT/mfma_util.txt"
grep "^{\"metric" "$OUT/bench_under_rocprof.log" > "$OUT/bench_under_rocprof.json"
find "$OUT/stats" -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} "$OUT/kernel_stats.csv"
rm -rf "$OUT/stats" "$OUT/fetch" "$OUT/write" "$OUT/sq"   # large raw CSVs; the summaries stay
ls "$OUT"; head -12 "$OUT/kernel_stats.csv"; cat "$OUT/mfma_util.txt"
