#!/bin/bash
# PMC diagnosis of single conv layers (run through gpurun):  tools/pmc_conv.sh <tag> <bench_conv args...>
# one SQ pass (wave / wait / MFMA-busy cycles) + GRBM_GUI_ACTIVE for the clock; summary printed by tools/pmc_conv_summary.py
set -e
TAG=$1; shift
OUT=gpurun_out/pmc_${TAG}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p "$OUT"
timeout -k 10 400 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE --output-format csv -d "$OUT/sq" -o s -- python3 tools/bench_conv.py "$@" > "$OUT/run.log" 2>&1
python3 tools/pmc_conv_summary.py "$OUT/sq" | tee "$OUT/summary.txt"
rm -f "$OUT"/sq/*/*counter_collection.csv "$OUT"/sq/*counter_collection.csv
