#!/bin/bash
# round-5, closing collection (run ON the GPU box) on the library with the longest-first causal launch order (commit 295ffe9): GPU tests, smoke,
# the bench line, and kernel traces of the shapes that order changed (the reference's benchmark shape, plain causal, cfg2 for the headline).
# Outputs under gpurun_out/prof_r05c/.
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
OUT="$GRAFT_REPO_ROOT/gpurun_out/prof_r05c"; mkdir -p "$OUT"
timeout 900 python -m pytest tests -m gpu -x -q > "$OUT/gpu_tests.log" 2>&1; echo "gpu tests rc=$?" | tee -a "$OUT/gpu_tests.log"; tail -3 "$OUT/gpu_tests.log"
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > "$OUT/smoke.log" 2>&1; echo "smoke rc=$?" | tee -a "$OUT/smoke.log"
timeout 900 python bench.py > "$OUT/bench.json" 2> "$OUT/bench.err"; echo "bench rc=$?"; cut -c1-600 "$OUT/bench.json"
run_trace() {  # name, command...
  name=$1; shift
  rm -rf /tmp/prof_$name
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$name -o t -- "$@" > "$OUT/$name.log" 2>&1)
  f=$(find /tmp/prof_$name -name "*kernel_stats*.csv" | head -1)
  [ -n "$f" ] && cp "$f" "$OUT/r05c_${name}_kernel_stats.csv" && head -5 "$f"
}
run_trace bench_cfg2_rpe python "$GRAFT_REPO_ROOT/bench.py" --steps 1000 --warmup 100 --no-extras
run_trace refshape_b16_s1024_causal_dense python "$GRAFT_REPO_ROOT/tools/run_one.py" --B 16 --S 1024 --causal --mode dense --what both --iters 100 --seconds 0.5
run_trace refshape_b16_s512_causal_dense python "$GRAFT_REPO_ROOT/tools/run_one.py" --B 16 --S 512 --causal --mode dense --what both --iters 100 --seconds 0.5
run_trace d128_b16_s1024_causal_dense python "$GRAFT_REPO_ROOT/tools/run_one.py" --D 128 --B 16 --S 1024 --causal --mode dense --what both --iters 100 --seconds 0.5
run_trace b16_s1024_causal_rpe python "$GRAFT_REPO_ROOT/tools/run_one.py" --B 16 --S 1024 --causal --mode rpe --what both --iters 100 --seconds 0.5
run_trace b16_s1024_causal_none python "$GRAFT_REPO_ROOT/tools/run_one.py" --B 16 --S 1024 --causal --mode none --what both --iters 100 --seconds 0.5
ls -la "$OUT"
