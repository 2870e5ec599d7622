import sys, time
sys.path.insert(0, "/root/repo")
import torch
from flasht5_amd import fast_rms_layernorm, fused_add_rms_layernorm, FAT5Config, FAT5ForConditionalGeneration
def gv(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        fn()
        with torch.cuda.graph(g, stream=s):
            for _ in range(n): fn()
    torch.cuda.current_stream().wait_stream(s); torch.cuda.synchronize()
    g.replay(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); g.replay(); b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
for rows, n in ((4096, 768), (16384, 768), (65536, 1024)):
    x = torch.randn(rows, n, device="cuda").bfloat16(); r = torch.randn_like(x); w = torch.ones(n, device="cuda").bfloat16()
    gy = torch.randn_like(x); gh = torch.randn_like(x)
    from flasht5_amd.rms_norm import add_rmsnorm_fwd, add_rmsnorm_bwd, rmsnorm_fwd, rmsnorm_bwd
    h, y, rstd = add_rmsnorm_fwd(x, r, w, 1e-6)
    tf_f = gv(lambda: add_rmsnorm_fwd(x, r, w, 1e-6))
    tf_u = gv(lambda: rmsnorm_fwd(x + r, w, 1e-6))
    tb_f = gv(lambda: add_rmsnorm_bwd(gy, h, w, rstd, gh, True))
    def ub():
        dx, dw = rmsnorm_bwd(gy, h, w, rstd, 1e-6)
        return dx + gh
    tb_u = gv(ub)
    e = 2
    print(f"({rows},{n}): fwd fused {tf_f:6.1f} us ({4*rows*n*e/tf_f/1e3:6.0f} GB/s) vs add+norm {tf_u:6.1f} us | bwd fused {tb_f:6.1f} us ({4*rows*n*e/tb_f/1e3:6.0f} GB/s) vs norm+add {tb_u:6.1f} us", flush=True)
# cfg5 step
for fuse in (False, True):
    cfg = FAT5Config(); cfg.fuse_add_norm = fuse
    torch.manual_seed(0)
    m = FAT5ForConditionalGeneration(cfg).cuda().bfloat16()
    ids = torch.randint(0, cfg.vocab_size, (4, 1024)).cuda(); labels = torch.randint(0, cfg.vocab_size, (4, 512)).cuda()
    def step():
        m.zero_grad(set_to_none=True)
        m(ids, labels).backward()
    for _ in range(3): step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): step()
    torch.cuda.synchronize()
    print(f"cfg5 fwd+bwd (B=4, 1024/512) fuse_add_norm={fuse}: {(time.perf_counter()-t0)/10*1e3:.2f} ms", flush=True)
    del m
