cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_fused_linear_gpu.py tests/test_fwd64_gpu.py -q --timeout=600 > gpurun_out/pytest_lin.log 2>&1
grep -E "^E  |^FAILED|passed|failed|^ERROR" gpurun_out/pytest_lin.log | cut -c1-260 | head -50
python - <<'PY'
import torch, time, sys
sys.path.insert(0, '.')
from flasht5_amd import rmsnorm_linear, linear_residual, fast_rms_layernorm
def tm(fn, it=50):
    for _ in range(10): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(it): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / it * 1e3
for (M, N, K) in [(4096, 2304, 768), (4096, 4096, 768), (4096, 768, 768), (4096, 768, 2048), (16384, 2304, 768)]:
    x = torch.randn(M, K, device='cuda').bfloat16(); g = torch.ones(K, device='cuda').bfloat16()
    W = (torch.randn(N, K, device='cuda') / K ** .5).bfloat16(); r = torch.randn(M, N, device='cuda').bfloat16()
    with torch.no_grad():
        t_f = tm(lambda: rmsnorm_linear(x, g, W, 1e-6))
        t_s = tm(lambda: torch.nn.functional.linear(fast_rms_layernorm(x, g, 1e-6), W))
        t_g = tm(lambda: torch.nn.functional.linear(x, W))
        t_r = tm(lambda: linear_residual(x, W, r))
        t_rs = tm(lambda: r + torch.nn.functional.linear(x, W))
    fl = 2.0 * M * N * K
    print(f"M={M} N={N} K={K}: norm+linear fused {t_f:7.1f} us ({fl/t_f/1e6:6.1f} TF/s) | separate {t_s:7.1f} | bare library GEMM {t_g:7.1f} ({fl/t_g/1e6:6.1f} TF/s) || linear+residual fused {t_r:7.1f} | separate {t_rs:7.1f}", flush=True)
PY
