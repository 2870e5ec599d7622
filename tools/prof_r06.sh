#!/bin/bash
# round-6 profile collection (run ON the GPU box) on the library with the table gradient's diagonal sums in the dQ workgroups of the one-launch backward:
# GPU tests, smoke, the bench line, rocprofv3 kernel-trace summaries of the bench command (cfg2) and of the S = 2048 / 8192 steps (T5-bias and dense mode) and the
# reference's benchmark shape, each behind a wall-clock pre-warm; the per-workgroup timeline of the cfg2 backward (FAT5_TRACE variant: tools/trace64.py); PMC HBM
# traffic per launch; PMC issue counters of the cfg2 backward; the two-rank dry run of the N > 1 bench.  Outputs under gpurun_out/prof_r06/.
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
OUT="$GRAFT_REPO_ROOT/gpurun_out/prof_r06"; mkdir -p "$OUT"
timeout 900 python -m pytest tests -m gpu -q > "$OUT/gpu_tests.log" 2>&1; echo "gpu tests rc=$?" | tee -a "$OUT/gpu_tests.log"; tail -3 "$OUT/gpu_tests.log"
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > "$OUT/smoke.log" 2>&1; echo "smoke rc=$?" | tee -a "$OUT/smoke.log"
timeout 900 python bench.py > "$OUT/bench.json" 2> "$OUT/bench.err"; echo "bench rc=$?"; cut -c1-700 "$OUT/bench.json"
run_trace() {  # name, command...
  name=$1; shift
  rm -rf /tmp/prof_$name
  (cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$name -o t -- "$@" > "$OUT/$name.log" 2>&1)
  f=$(find /tmp/prof_$name -name "*kernel_stats*.csv" | head -1)
  [ -n "$f" ] && cp "$f" "$OUT/r06_${name}_kernel_stats.csv" && head -6 "$f"
}
run_trace bench_cfg2_rpe python "$GRAFT_REPO_ROOT/bench.py" --steps 1000 --warmup 100 --no-extras
run_trace s2048_rpe python "$GRAFT_REPO_ROOT/tools/run_one.py" --S 2048 --mode rpe --what both --iters 100 --seconds 0.5
run_trace s8192_rpe python "$GRAFT_REPO_ROOT/tools/run_one.py" --S 8192 --mode rpe --what both --iters 30 --seconds 0.5
run_trace s2048_dense python "$GRAFT_REPO_ROOT/tools/run_one.py" --S 2048 --mode dense --what both --iters 50 --seconds 0.5
run_trace s8192_dense python "$GRAFT_REPO_ROOT/tools/run_one.py" --S 8192 --mode dense --what both --iters 10 --seconds 0.3
run_trace refshape_b16_s1024_causal_dense python "$GRAFT_REPO_ROOT/tools/run_one.py" --B 16 --S 1024 --causal --mode dense --what both --iters 50 --seconds 0.5
run_trace d128_b16_s1024_causal_dense python "$GRAFT_REPO_ROOT/tools/run_one.py" --D 128 --B 16 --S 1024 --causal --mode dense --what both --iters 50 --seconds 0.5
FAT5_LIB_VARIANT=trace python tools/trace64.py --S 512 --mode rpe --variant 0 --stage 3 2>&1 | grep -v amdgpu.ids > "$OUT/r06_trace64_cfg2_qdiag.log"
FAT5_LIB_VARIANT=trace python tools/trace64.py --S 512 --mode rpe --variant 16777216 --stage 3 2>&1 | grep -v amdgpu.ids > "$OUT/r06_trace64_cfg2_kvdiag.log"
cat "$OUT/r06_trace64_cfg2_qdiag.log" "$OUT/r06_trace64_cfg2_kvdiag.log"
timeout 1500 python tools/pmc_traffic.py > "$OUT/pmc_traffic.log" 2>&1; cp gpurun_out/pmc_traffic.json "$OUT/" 2>/dev/null; tail -25 "$OUT/pmc_traffic.log"
C1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE"
C2="SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS"
bash tools/pmc.sh "--S 512 --mode rpe --what bwd --iters 20" "$C1" "$C2" > "$OUT/r06_pmc_bwd_fused64_cfg2.txt" 2>&1
FAT5_BENCH_SHARE_GPU=1 timeout 600 python bench.py --gpus 2 --steps 400 --warmup 40 --no-extras > "$OUT/r06_bench_2rank_shared_gpu.log" 2>&1
tail -1 "$OUT/r06_bench_2rank_shared_gpu.log" | cut -c1-900
ls -la "$OUT"
