#!/bin/bash
# Extra round-3 evidence (run through gpurun): SQ / MFMA-busy pass of the whole-encoder training step at B = 32 x 32 @224^2, and
# rocprofv3 kernel statistics of the reference's own recipe (40x40, encoders off, fp16).
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/prof_r3_extra
mkdir -p "$OUT"
ARGS="bench.py --release 4 --hw 224 --batch 32 --modalities video --steps 2 --warmup 1 --no-cpu-baseline"
timeout -k 10 500 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE --output-format csv -d "$OUT/sq" -o q -- python3 $ARGS > "$OUT/sq.log" 2>&1
python3 tools/pmc_conv_summary.py "$OUT/sq" > "$OUT/mfma_util_release4.txt"
rm -rf "$OUT/sq"
ARGS2="bench.py --hw 40 --encoders off --precision fp16 --steps 4 --warmup 2 --no-cpu-baseline --no-alt"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats40" -o s -- python3 $ARGS2 > "$OUT/bench40_under_rocprof.log" 2>&1
find "$OUT/stats40" -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} "$OUT/kernel_stats_hw40_fp16_off.csv"
rm -rf "$OUT/stats40"
grep -A2 "wgrad\|win_kernel<128" "$OUT/mfma_util_release4.txt" | head -30
