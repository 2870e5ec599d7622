#!/bin/bash
# round-3 profile collection (run ON the GPU box): rocprofv3 kernel-trace summaries of the bench command (cfg2) and of the
# S = 2048 / 8192 steps, PMC HBM traffic, PMC issue counters of the pipelined forward and backward, and the 2-rank dry runs of
# bench.py (two ranks sharing the one GPU over gloo: the N > 1 control flow end to end).  Outputs under gpurun_out/prof_r03/.
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
OUT="$GRAFT_REPO_ROOT/gpurun_out/prof_r03"; mkdir -p "$OUT"
run_trace() {  # name, command...
  name=$1; shift
  rm -rf /tmp/prof_$name
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$name -o t -- "$@" > "$OUT/$name.log" 2>&1)
  f=$(find /tmp/prof_$name -name "*kernel_stats*.csv" | head -1)
  [ -n "$f" ] && cp "$f" "$OUT/r03_${name}_kernel_stats.csv" && head -6 "$f"
}
run_trace bench_cfg2_rpe python "$GRAFT_REPO_ROOT/bench.py" --steps 200 --warmup 20 --no-extras
run_trace s2048_rpe python "$GRAFT_REPO_ROOT/tools/run_one.py" --S 2048 --mode rpe --what both --iters 20
run_trace s8192_rpe python "$GRAFT_REPO_ROOT/tools/run_one.py" --S 8192 --mode rpe --what both --iters 10
run_trace s8192_none python "$GRAFT_REPO_ROOT/tools/run_one.py" --S 8192 --mode none --what both --iters 10
run_trace s8192_dense python "$GRAFT_REPO_ROOT/tools/run_one.py" --S 8192 --mode dense --what both --iters 5
# config 5: the whole optimizer step of FAT5-base replayed from one HIP graph (GraphedTrainStep), 25 steps
run_trace cfg5_graphed_step python "$GRAFT_REPO_ROOT/tools/graph_step_debug.py" class --full --fuse --nosync
timeout 1200 python tools/pmc_traffic.py > "$OUT/pmc_traffic.log" 2>&1; cp gpurun_out/pmc_traffic.json "$OUT/" 2>/dev/null
C1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE"
C2="SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS"
bash tools/pmc.sh "--S 8192 --mode rpe --what fwd --iters 3" "$C1" "$C2" > "$OUT/r03_pmc_fwd64_s8192_rpe.txt" 2>&1
bash tools/pmc.sh "--S 8192 --mode none --what fwd --iters 3" "$C1" "$C2" > "$OUT/r03_pmc_fwd64_s8192_none.txt" 2>&1
bash tools/pmc.sh "--S 2048 --mode rpe --what fwd --iters 5" "$C1" "$C2" > "$OUT/r03_pmc_fwd64_s2048_rpe.txt" 2>&1
bash tools/pmc.sh "--S 8192 --mode rpe --what bwd --iters 2" "$C1" "$C2" > "$OUT/r03_pmc_bwd_s8192_rpe.txt" 2>&1
for sc in weak strong; do
  FAT5_BENCH_SHARE_GPU=1 timeout 600 python bench.py --gpus 2 --steps 200 --warmup 20 --scaling $sc --no-extras > "$OUT/r03_bench_2rank_shared_gpu_$sc.log" 2>&1
  tail -2 "$OUT/r03_bench_2rank_shared_gpu_$sc.log" | cut -c1-400
done
ls -la "$OUT"
