#!/bin/bash
# round-2 profile collection (run ON the GPU box): rocprofv3 kernel-trace summaries of the bench command (cfg2) and of the
# S=8192 RPE / dense steps, PMC HBM traffic, PMC issue counters of the pipelined forward.  Outputs under gpurun_out/prof_r02b/.
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
OUT="$GRAFT_REPO_ROOT/gpurun_out/prof_r02b"; mkdir -p "$OUT"
run_trace() {  # name, command...
  name=$1; shift
  rm -rf /tmp/prof_$name
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$name -o t -- "$@" > "$OUT/$name.log" 2>&1)
  f=$(find /tmp/prof_$name -name "*kernel_stats*.csv" | head -1)
  [ -n "$f" ] && cp "$f" "$OUT/r02b_${name}_kernel_stats.csv" && head -8 "$f"
}
run_trace bench_cfg2_rpe python "$GRAFT_REPO_ROOT/bench.py" --steps 200 --warmup 20 --no-extras
run_trace s8192_rpe python "$GRAFT_REPO_ROOT/tools/run_one.py" --S 8192 --mode rpe --what both --iters 10
run_trace s8192_none python "$GRAFT_REPO_ROOT/tools/run_one.py" --S 8192 --mode none --what both --iters 10
run_trace s8192_dense python "$GRAFT_REPO_ROOT/tools/run_one.py" --S 8192 --mode dense --what both --iters 5
run_trace s2048_rpe python "$GRAFT_REPO_ROOT/tools/run_one.py" --S 2048 --mode rpe --what both --iters 20
timeout 900 python tools/pmc_traffic.py > "$OUT/pmc_traffic.log" 2>&1; cp gpurun_out/pmc_traffic.json "$OUT/" 2>/dev/null
bash tools/pmc.sh "--S 8192 --mode rpe --what fwd --iters 3" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS" > "$OUT/r02b_pmc_fwd64_s8192_rpe.txt" 2>&1
bash tools/pmc.sh "--S 8192 --mode rpe --what bwd --iters 2" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS" > "$OUT/r02b_pmc_bwd_s8192_rpe.txt" 2>&1
ls -la "$OUT"
