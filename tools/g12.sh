cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_bwd64_gpu.py -q --timeout=600 -x > gpurun_out/pytest_b64.log 2>&1
grep -E "^E  |^FAILED|passed|failed|^ERROR" gpurun_out/pytest_b64.log | cut -c1-260 | head -30
timeout 600 python tools/fuzz64.py 60 7 2>&1 | tail -4
echo "=== dkdv: kv64 forced wg256 (4+8192) ==="
timeout 300 python tools/attn_time.py --S 2048,4096,8192 --modes none,rpe --what dkdv --variant 8196 2>&1 | grep -v amdgpu.ids
echo "=== default ==="
timeout 300 python tools/attn_time.py --S 2048,4096,8192 --modes rpe --what dq,dkdv,bwd 2>&1 | grep -v amdgpu.ids
