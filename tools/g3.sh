cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 2400 python -m pytest tests -q -m gpu --timeout=900 > gpurun_out/pytest_gpu.log 2>&1
grep -E "^FAILED|passed|failed|^ERROR" gpurun_out/pytest_gpu.log | cut -c1-250 | head -60
timeout 300 python __graft_entry__.py smoke 2>&1 | tail -4
