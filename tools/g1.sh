cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 300 python tools/calib_elem.py 2>&1 | grep -v amdgpu.ids | tail -30
timeout 1700 python -m pytest tests -q -m gpu --maxfail=60 --timeout=600 > gpurun_out/pytest_gpu.log 2>&1
grep -E "^FAILED|passed|failed" gpurun_out/pytest_gpu.log | head -60
echo "=== timing S=1024,2048,4096 default ==="
timeout 300 python tools/attn_time.py --S 1024,2048,4096 --modes none,rpe --what fwd,dq,dkdv,bwd 2>&1 | grep -v amdgpu.ids
echo "=== forced 64-wide (variant 21 = FWD64|KV64|Q64 on) ==="
timeout 300 python tools/attn_time.py --S 1024,2048,4096 --modes none,rpe --what fwd,dq,dkdv,bwd --variant 21 2>&1 | grep -v amdgpu.ids
