"""where the RPE dK/dV time goes at cfg2 (developer tool)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from attn_helpers import make_inputs
from flasht5_amd.flash_attention_v2_bias import AttentionPlan
from flasht5_amd import positional_encoding as pe
S = int(sys.argv[1]) if len(sys.argv) > 1 else 512
def graph_time(fn, it=50):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(it): fn()
    g.replay(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record(); g.replay(); e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / it * 1e3
q, k, v, _, do = make_inputs(4, 12, S, S, 64, torch.bfloat16, None, seed=1, strided=True)
table = (torch.randn(32, 12) * 0.5).cuda()
for name, kw in (("none", {}), ("rpe nograd", dict(rpe1d=pe.rpe1d_from_table(table), radius=128, need_dbias=False)),
                 ("rpe grad", dict(rpe1d=pe.rpe1d_from_table(table), radius=128)),
                 ("rpe R=16", dict(rpe1d=pe.rpe1d_from_table(table)[:, 112:145].contiguous(), radius=16))):
    plan = AttentionPlan(q, k, v, do, sm_scale=0.125, **kw)
    plan.forward()
    print(f"S={S} {name:11s}: fwd {graph_time(plan.forward):6.1f} | dq {graph_time(lambda: plan.backward(1)):6.1f} dkdv {graph_time(lambda: plan.backward(2)):6.1f} fused {graph_time(lambda: plan.backward(3)):6.1f} us", flush=True)
