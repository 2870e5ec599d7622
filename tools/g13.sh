cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_attention_gpu.py tests/test_bwd64_gpu.py -q --timeout=600 -x 2>&1 | tail -2
timeout 300 python tools/attn_time.py --S 512,2048,8192 --modes rpe --what fwd,bwd,red --iters 50 2>&1 | grep -v amdgpu.ids
timeout 300 python bench.py --no-extras --steps 2000 --warmup 200 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"
