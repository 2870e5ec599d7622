#!/bin/bash
# kernel-trace summaries of the bandwidth-bound kernels (RMSNorm, add+RMSNorm, cross-entropy, AdamWScale) -> gpurun_out/prof_rowwise/
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
OUT="$GRAFT_REPO_ROOT/gpurun_out/prof_rowwise"; mkdir -p "$OUT"
for name in bench_rowwise time_addnorm time_adamw; do
  rm -rf /tmp/prof_$name
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$name -o t -- python "$GRAFT_REPO_ROOT/tools/$name.py" > "$OUT/$name.log" 2>&1)
  f=$(find /tmp/prof_$name -name "*kernel_stats*.csv" | head -1)
  [ -n "$f" ] && cp "$f" "$OUT/r02b_${name}_kernel_stats.csv" && head -12 "$f" | cut -c1-160
done
