import sys, torch
sys.path.insert(0, "/root/repo")
from flasht5_amd.flash_attention_v2_bias import AttentionPlan
def t(fn, n=20):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3
for S in (512, 1024):
    g = torch.Generator().manual_seed(S)
    q, k, v, do = (torch.randn(16, 12, S, 64, generator=g).bfloat16().cuda() for _ in range(4))
    bias = torch.randn(1, 12, S, S, generator=g).bfloat16().cuda()
    for name, kw in (("none", {}), ("dense no dbias", dict(bias=bias, need_dbias=False)), ("dense", dict(bias=bias))):
        plan = AttentionPlan(q, k, v, do, causal=True, sm_scale=1.3, **kw)
        plan.forward(); plan.backward()
        st = [t(lambda s=s: plan.backward(s)) for s in (1, 2)]
        print(f"S={S} {name:16s}: fwd {t(plan.forward):7.1f} bwd {t(plan.backward):7.1f}  dq {st[0]:.1f} dkdv {st[1]:.1f}", flush=True)
