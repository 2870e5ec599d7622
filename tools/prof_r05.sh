#!/bin/bash
# round-5 profile collection (run ON the GPU box): rocprofv3 kernel-trace summaries of the bench command (cfg2), of the S = 2048 / 8192 steps in T5-bias
# (RPE) mode and -- new this round -- of the DENSE-bias steps (the reference's own operator: (4,12,2048 / 8192) and its benchmark shape (16,12,1024) causal),
# each behind a wall-clock pre-warm; PMC HBM traffic per launch (incl. the dense kernels); PMC issue counters of the new dense backward kernels; the
# two-rank dry run of the N > 1 bench (one all-reduce per step).  Outputs under gpurun_out/prof_r05/; the summaries worth keeping are copied to profiles/.
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
OUT="$GRAFT_REPO_ROOT/gpurun_out/prof_r05"; mkdir -p "$OUT"
run_trace() {  # name, command...
  name=$1; shift
  rm -rf /tmp/prof_$name
  (cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$name -o t -- "$@" > "$OUT/$name.log" 2>&1)
  f=$(find /tmp/prof_$name -name "*kernel_stats*.csv" | head -1)
  [ -n "$f" ] && cp "$f" "$OUT/r05_${name}_kernel_stats.csv" && head -6 "$f"
}
run_trace bench_cfg2_rpe python "$GRAFT_REPO_ROOT/bench.py" --steps 1000 --warmup 100 --no-extras
run_trace s2048_rpe python "$GRAFT_REPO_ROOT/tools/run_one.py" --S 2048 --mode rpe --what both --iters 100 --seconds 0.5
run_trace s8192_rpe python "$GRAFT_REPO_ROOT/tools/run_one.py" --S 8192 --mode rpe --what both --iters 30 --seconds 0.5
run_trace s2048_dense python "$GRAFT_REPO_ROOT/tools/run_one.py" --S 2048 --mode dense --what both --iters 50 --seconds 0.5
run_trace s8192_dense python "$GRAFT_REPO_ROOT/tools/run_one.py" --S 8192 --mode dense --what both --iters 10 --seconds 0.3
run_trace refshape_b16_s1024_causal_dense python "$GRAFT_REPO_ROOT/tools/run_one.py" --B 16 --S 1024 --causal --mode dense --what both --iters 50 --seconds 0.5
run_trace b16_s1024_causal_none python "$GRAFT_REPO_ROOT/tools/run_one.py" --B 16 --S 1024 --causal --mode none --what both --iters 50 --seconds 0.5
timeout 1500 python tools/pmc_traffic.py > "$OUT/pmc_traffic.log" 2>&1; cp gpurun_out/pmc_traffic.json "$OUT/" 2>/dev/null
C1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE"
C2="SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS"
bash tools/pmc.sh "--S 8192 --mode dense --what bwd --iters 2" "$C1" "$C2" > "$OUT/r05_pmc_bwd_s8192_dense.txt" 2>&1
bash tools/pmc.sh "--S 512 --mode rpe --what bwd --iters 20" "$C1" "$C2" > "$OUT/r05_pmc_bwd_fused64_cfg2.txt" 2>&1
# N > 1 control flow on one GPU (gloo; both ranks on cuda:0): the default -- one all-reduce per step -- and the bucketed form
FAT5_BENCH_SHARE_GPU=1 timeout 600 python bench.py --gpus 2 --steps 400 --warmup 40 --no-extras > "$OUT/r05_bench_2rank_shared_gpu_per_step.log" 2>&1
FAT5_BENCH_SHARE_GPU=1 timeout 600 python bench.py --gpus 2 --steps 400 --warmup 40 --no-extras --bucket-allreduce > "$OUT/r05_bench_2rank_shared_gpu_bucketed.log" 2>&1
tail -2 "$OUT/r05_bench_2rank_shared_gpu_per_step.log" "$OUT/r05_bench_2rank_shared_gpu_bucketed.log"
ls -la "$OUT"
