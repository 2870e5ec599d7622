cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_fused_linear_gpu.py -q --timeout=600 > gpurun_out/pytest_lin.log 2>&1
grep -E "^E  |^FAILED|passed|failed|^ERROR" gpurun_out/pytest_lin.log | cut -c1-260 | head -30
timeout 900 python bench.py --steps 200 --warmup 20 > gpurun_out/bench_r03a.json 2> gpurun_out/bench_r03a.err; tail -c 600 gpurun_out/bench_r03a.err; python - <<'PY'
import json
try:
    d = json.loads(open('gpurun_out/bench_r03a.json').read().strip().splitlines()[-1])
    for k in ('value', 'ms_per_step', 'roofline', 'by_seq', 'roofline_by_seq', 'eager_autograd', 'n3_fusions', 'cpu_baseline'):
        print(k, json.dumps(d.get(k))[:1800])
except Exception as e:
    print('bench parse failed', e)
PY
