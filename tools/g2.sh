cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_fwd64_gpu.py -q --timeout=600 -x > gpurun_out/pytest_f64.log 2>&1
grep -E "^E  |^FAILED|passed|failed" gpurun_out/pytest_f64.log | cut -c1-300 | head -40
echo "=== fwd timing: default dispatch ==="
timeout 300 python tools/attn_time.py --S 1024,2048,4096,8192 --modes none,rpe --what fwd 2>&1 | grep -v amdgpu.ids
echo "=== fwd64 forced, unsplit (1+2048) ==="
timeout 300 python tools/attn_time.py --S 1024,2048,4096,8192 --modes none,rpe --what fwd --variant 2049 2>&1 | grep -v amdgpu.ids
echo "=== fwd64 forced, ksplit (1+1024) ==="
timeout 300 python tools/attn_time.py --S 1024,2048,4096,8192 --modes none,rpe --what fwd --variant 1025 2>&1 | grep -v amdgpu.ids
echo "=== 32-row body (2) ==="
timeout 300 python tools/attn_time.py --S 1024,2048,4096 --modes none,rpe --what fwd --variant 2 2>&1 | grep -v amdgpu.ids
