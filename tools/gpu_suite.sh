#!/bin/bash
# one gpurun call: GPU tests, smoke, bench, rocprof kernel-trace summary
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
{
echo "=== pytest -m gpu ==="
timeout 1500 python -m pytest tests -q -m gpu --maxfail=40 > gpurun_out/pytest_gpu.log 2>&1; grep -E "AssertionError|^E +assert|^FAILED|passed|failed" gpurun_out/pytest_gpu.log | head -80
echo "=== smoke ==="
timeout 300 python __graft_entry__.py smoke 2>&1 | tail -8
echo "=== bench ==="
timeout 600 python bench.py 2>&1 | tail -3
} > gpurun_out/suite.log 2>&1
if [ "$1" == "prof" ]; then
  cd /tmp && rm -rf /tmp/prof && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -o bench -- python "$GRAFT_REPO_ROOT/bench.py" --steps 50 --warmup 5 --no-extras > "$GRAFT_REPO_ROOT/gpurun_out/prof.log" 2>&1
  cd "$GRAFT_REPO_ROOT"
  find /tmp/prof -type f | head -20
  for f in $(find /tmp/prof -name "*kernel_stats*.csv"); do cp "$f" gpurun_out/bench_kernel_stats.csv; done
  head -12 gpurun_out/bench_kernel_stats.csv
fi
tail -c 5000 gpurun_out/suite.log
