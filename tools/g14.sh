cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
rm -rf /tmp/prof_b
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_b -o t -- python "$GRAFT_REPO_ROOT/bench.py" --steps 2000 --warmup 200 --no-extras > /tmp/prof_b.log 2>&1)
f=$(find /tmp/prof_b -name "*kernel_stats*.csv" | head -1); head -5 "$f" | cut -c1-200
tail -1 /tmp/prof_b.log | cut -c1-200
