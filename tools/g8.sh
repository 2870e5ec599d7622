cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_fused_linear_gpu.py -q --timeout=600 2>&1 | tail -5
timeout 600 python tools/time_cfg5.py 2>&1 | grep "cfg5"
