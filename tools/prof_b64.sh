#!/bin/bash
# kernel-trace + issue counters of the S=8192 backward (run ON the GPU box): bash tools/prof_b64.sh [mode]
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
MODE=${1:-none}
OUT="$GRAFT_REPO_ROOT/gpurun_out/prof_b64"; mkdir -p "$OUT"
rm -rf /tmp/prof_b64
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_b64 -o t -- python "$GRAFT_REPO_ROOT/tools/run_one.py" --S 8192 --mode $MODE --what bwd --iters 10 > "$OUT/trace_$MODE.log" 2>&1)
f=$(find /tmp/prof_b64 -name "*kernel_stats*.csv" | head -1)
[ -n "$f" ] && cp "$f" "$OUT/b64_${MODE}_kernel_stats.csv" && head -6 "$f"
bash tools/pmc.sh "--S 8192 --mode $MODE --what bwd --iters 2" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS" > "$OUT/pmc_$MODE.txt" 2>&1
grep -A9 "kv64" "$OUT/pmc_$MODE.txt" | head -40
