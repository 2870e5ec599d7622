"""developer timing of the reference's published benchmark shape (B=16, H=12, causal, dense (1,H,S,S) bias + dbias): per stage."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from flasht5_amd.flash_attention_v2_bias import AttentionPlan

def t(fn, n=20):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3

Dh = int(sys.argv[1]) if len(sys.argv) > 1 else 64
for S in (512, 1024):
    g = torch.Generator().manual_seed(S + Dh)
    q, k, v, do = (torch.randn(16, 12, S, Dh, generator=g).bfloat16().cuda() for _ in range(4))
    bias = torch.randn(1, 12, S, S, generator=g).bfloat16().cuda()
    plan = AttentionPlan(q, k, v, do, bias=bias, causal=True, sm_scale=1.3)
    plan.forward(); plan.backward()
    f = 4.0 * 16 * 12 * S * S * Dh / 2
    tf, tb = t(plan.forward), t(plan.backward)
    st = [t(lambda s=s: plan.backward(s)) for s in (1, 2, 4)]
    print(f"S={S} D={Dh}: fwd {tf:7.1f} us ({f/tf/1e6:6.1f} TF/s)  bwd {tb:7.1f} us ({2.5*f/tb/1e6:6.1f} TF/s)  stages dq {st[0]:.1f} dkdv {st[1]:.1f} reduce {st[2]:.1f}  ws {plan.ws.numel()/1e6:.1f} MB", flush=True)
