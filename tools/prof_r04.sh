#!/bin/bash
# round-4 profile collection (run ON the GPU box): rocprofv3 kernel-trace summaries of the bench command (cfg2) and of the S = 2048 /
# 8192 steps -- each behind bench.py's wall-clock pre-warm and with >= 50 (S = 8192: 30) launches, so the averages are warm-clock
# numbers comparable with the bench's event timings --, PMC HBM traffic per launch, PMC issue counters of the pipelined forward
# (variants) and backward.  Outputs under gpurun_out/prof_r04/; the summaries worth keeping are copied to profiles/ by hand.
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
OUT="$GRAFT_REPO_ROOT/gpurun_out/prof_r04"; mkdir -p "$OUT"
run_trace() {  # name, command...
  name=$1; shift
  rm -rf /tmp/prof_$name
  (cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$name -o t -- "$@" > "$OUT/$name.log" 2>&1)
  f=$(find /tmp/prof_$name -name "*kernel_stats*.csv" | head -1)
  [ -n "$f" ] && cp "$f" "$OUT/r04_${name}_kernel_stats.csv" && head -6 "$f"
}
run_trace bench_cfg2_rpe python "$GRAFT_REPO_ROOT/bench.py" --steps 1000 --warmup 100 --no-extras
run_trace s2048_rpe python "$GRAFT_REPO_ROOT/tools/run_one.py" --S 2048 --mode rpe --what both --iters 100 --seconds 0.5
run_trace s8192_rpe python "$GRAFT_REPO_ROOT/tools/run_one.py" --S 8192 --mode rpe --what both --iters 30 --seconds 0.5
run_trace s8192_none python "$GRAFT_REPO_ROOT/tools/run_one.py" --S 8192 --mode none --what both --iters 30 --seconds 0.5
run_trace s8192_dense python "$GRAFT_REPO_ROOT/tools/run_one.py" --S 8192 --mode dense --what both --iters 10 --seconds 0.3
run_trace s8192_rpe_fp16 python "$GRAFT_REPO_ROOT/tools/run_one.py" --S 8192 --mode rpe --what fwd --iters 30 --seconds 0.5 --dtype fp16
timeout 1500 python tools/pmc_traffic.py > "$OUT/pmc_traffic.log" 2>&1; cp gpurun_out/pmc_traffic.json "$OUT/" 2>/dev/null
C1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE"
C2="SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS"
bash tools/pmc.sh "--S 8192 --mode rpe --what fwd --iters 3" "$C1" "$C2" > "$OUT/r04_pmc_fwd64_s8192_rpe.txt" 2>&1
bash tools/pmc.sh "--S 8192 --mode none --what fwd --iters 3" "$C1" "$C2" > "$OUT/r04_pmc_fwd64_s8192_none.txt" 2>&1
bash tools/pmc.sh "--S 8192 --mode rpe --what fwd --iters 3 --dtype fp16" "$C1" "$C2" > "$OUT/r04_pmc_fwd64_s8192_rpe_fp16.txt" 2>&1
bash tools/pmc.sh "--S 8192 --mode rpe --what bwd --iters 2" "$C1" "$C2" > "$OUT/r04_pmc_bwd_s8192_rpe.txt" 2>&1
bash tools/pmc.sh "--S 512 --mode rpe --what bwd --iters 20" "$C1" "$C2" > "$OUT/r04_pmc_bwd_fused64_cfg2.txt" 2>&1
ls -la "$OUT"
