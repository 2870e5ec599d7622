"""dense-bias (drop-in API) timings, graph-replayed (developer tool): python tools/time_dense.py [S]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from attn_helpers import make_inputs
from flasht5_amd.flash_attention_v2_bias import AttentionPlan
from flasht5_amd import positional_encoding as pe
def graph_time(fn, it=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(it): fn()
    g.replay(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record(); g.replay(); e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / it * 1e3
for S in [int(a) for a in sys.argv[1:]] or [512, 2048]:
    q, k, v, _, do = make_inputs(4, 12, S, S, 64, torch.bfloat16, None, seed=1, strided=True)
    table = (torch.randn(32, 12) * 0.5).cuda()
    bias = pe.compute_bias(table, S, S).to(torch.bfloat16).contiguous()
    for name, kw in (("dense (1,H,M,N) +dbias", dict(bias=bias)), ("dense, no dbias", dict(bias=bias, need_dbias=False))):
        plan = AttentionPlan(q, k, v, do, sm_scale=0.125, **kw)
        plan.forward()
        f = 4.0 * 4 * 12 * S * S * 64
        tf, tb = graph_time(plan.forward), graph_time(plan.backward)
        print(f"S={S} {name:24s}: fwd {tf:8.1f} us ({f/tf/1e6:6.1f} TF/s) | bwd {tb:8.1f} us ({2.5*f/tb/1e6:6.1f} TF/s) | dq {graph_time(lambda: plan.backward(1)):7.1f} dkdv {graph_time(lambda: plan.backward(2)):7.1f} reduce {graph_time(lambda: plan.backward(4)):7.1f}", flush=True)
