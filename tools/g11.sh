cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 2400 python -m pytest tests -q -m gpu --timeout=900 > gpurun_out/pytest_gpu.log 2>&1
grep -E "^FAILED|passed|failed|^ERROR" gpurun_out/pytest_gpu.log | cut -c1-250 | head -40
timeout 900 python tools/parity_report.py gpurun_out/parity_r03.json 2>&1 | grep -v amdgpu.ids | tail -8
timeout 300 python __graft_entry__.py smoke 2>&1 | tail -3
timeout 900 python bench.py > gpurun_out/bench_r03b.json 2> gpurun_out/bench_r03b.err; tail -c 300 gpurun_out/bench_r03b.err; head -c 600 gpurun_out/bench_r03b.json
