#!/bin/bash
# round-5, second collection (run ON the GPU box): the kernels added after prof_r05.sh ran -- the head_dim-128 pipelined forward (bias none / T5 table / dense),
# the masked blocks inside the forward sweep (plain causal) -- plus the bench command again on the final library.  Outputs under gpurun_out/prof_r05b/.
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
OUT="$GRAFT_REPO_ROOT/gpurun_out/prof_r05b"; mkdir -p "$OUT"
run_trace() {  # name, command...
  name=$1; shift
  rm -rf /tmp/prof_$name
  (cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$name -o t -- "$@" > "$OUT/$name.log" 2>&1)
  f=$(find /tmp/prof_$name -name "*kernel_stats*.csv" | head -1)
  [ -n "$f" ] && cp "$f" "$OUT/r05_${name}_kernel_stats.csv" && head -5 "$f"
}
run_trace bench_cfg2_rpe python "$GRAFT_REPO_ROOT/bench.py" --steps 1000 --warmup 100 --no-extras
run_trace d128_s8192_rpe_fwd python "$GRAFT_REPO_ROOT/tools/run_one.py" --D 128 --S 8192 --mode rpe --what fwd --iters 30 --seconds 0.5
run_trace d128_s8192_none_fwd python "$GRAFT_REPO_ROOT/tools/run_one.py" --D 128 --S 8192 --mode none --what fwd --iters 30 --seconds 0.5
run_trace d128_s2048_rpe_fwd python "$GRAFT_REPO_ROOT/tools/run_one.py" --D 128 --S 2048 --mode rpe --what fwd --iters 100 --seconds 0.5
run_trace d128_b16_s1024_causal_dense_fwd python "$GRAFT_REPO_ROOT/tools/run_one.py" --D 128 --B 16 --S 1024 --causal --mode dense --what fwd --iters 100 --seconds 0.5
run_trace b16_s1024_causal_none python "$GRAFT_REPO_ROOT/tools/run_one.py" --B 16 --S 1024 --causal --mode none --what both --iters 50 --seconds 0.5
ls -la "$OUT"
