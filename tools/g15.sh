cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
for i in 1 2; do timeout 300 python bench.py --no-extras --steps 3000 --warmup 300 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"; done
timeout 300 python tools/attn_time.py --S 512 --modes none,rpe --what fwd,bwd --iters 100 2>&1 | grep -v amdgpu.ids
timeout 300 python tools/attn_time.py --S 8192 --modes rpe --what fwd --dtype fp16 --variant 1 2>&1 | grep -v amdgpu.ids
timeout 300 python tools/attn_time.py --S 8192 --modes rpe --what fwd --dtype fp16 --variant 2 2>&1 | grep -v amdgpu.ids
timeout 300 python tools/attn_time.py --B 16 --S 1024 --causal --modes none,rpe,dense --what fwd,bwd 2>&1 | grep -v amdgpu.ids
