cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
timeout 600 python tools/time_cfg5.py 2>&1 | grep -v amdgpu.ids
timeout 600 python -m pytest tests/test_bwd64_gpu.py tests/test_attention_gpu.py -q --timeout=600 -x 2>&1 | tail -3
