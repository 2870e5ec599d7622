cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
for v in 0 2049 1025; do timeout 200 python tools/attn_time.py --S 512,768,1024 --modes none,rpe --what fwd --variant $v --iters 50 2>&1 | grep -v amdgpu.ids; done
