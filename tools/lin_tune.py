"""developer tool: fat5_linear_fused against torch's library GEMM per shape (kernel time by graph replay) and a correctness
check against the fp32 product.  (`--cfgs`: only meaningful in a tuning build whose API carries a configuration selector.)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from flasht5_amd import _lib
from flasht5_amd.fused_linear import fold_weights
from flasht5_amd.rms_norm import fast_rms_layernorm
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from bench import graph_time

cfgs = [int(c) for c in (sys.argv[sys.argv.index("--cfgs") + 1].split(",") if "--cfgs" in sys.argv else ["0"])]
lib = _lib.load()
dev = torch.device("cuda")
shapes = [("qkv  norm", 4096, 2304, 768, True), ("wi01 norm", 4096, 4096, 768, True), ("q    norm", 2048, 768, 768, True),
          ("o    res ", 4096, 768, 768, False), ("o    res ", 2048, 768, 768, False), ("wo   res ", 4096, 768, 2048, False), ("wo   res ", 2048, 768, 2048, False)]
for name, M, N, K, norm in shapes:
    g = torch.Generator().manual_seed(M + N)
    x = torch.randn(M, K, generator=g).to(dev).bfloat16()
    W = (torch.randn(N, K, generator=g) / K ** 0.5).to(dev).bfloat16()
    gw = (1 + 0.1 * torch.randn(K, generator=g)).to(dev).bfloat16()
    res = torch.randn(M, N, generator=g).to(dev).bfloat16()
    out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    rstd = torch.empty(M, device=dev, dtype=torch.float32)
    wg = fold_weights((W,), gw) if norm else W
    if norm:
        ref = (fast_rms_layernorm(x, gw, 1e-6).float() @ W.float().t())
        t_lib = graph_time(lambda: torch.nn.functional.linear(fast_rms_layernorm(x, gw, 1e-6), W))
    else:
        ref = res.float() + (x.float() @ W.float().t()).bfloat16().float()
        t_lib = graph_time(lambda: res + torch.nn.functional.linear(x, W))
    line = f"{name} {M:5d}x{N:5d}x{K:5d}  library {t_lib * 1e6:6.1f} us |"
    for c in cfgs:
        def run():
            _lib.check(lib.fat5_linear_fused(x.data_ptr(), wg.data_ptr(), None if norm else res.data_ptr(), out.data_ptr(), rstd.data_ptr() if norm else None,
                                             M, N, K, K, K, N, N, (1 if norm else 0), 1e-6, _lib.dtype_code(x.dtype), _lib.stream_ptr(dev)), "lin")
        run(); torch.cuda.synchronize()
        err = float((out.float() - ref).abs().max() / ref.abs().max())
        t = graph_time(run)
        line += f" cfg{c} {t * 1e6:6.1f} us ({2 * M * N * K / t / 1e12:4.0f} TF/s, err {err:.1e}) |"
    print(line, flush=True)
