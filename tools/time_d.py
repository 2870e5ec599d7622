"""head-dim sweep (developer tool): python tools/time_d.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from attn_helpers import make_inputs
from flasht5_amd.flash_attention_v2_bias import AttentionPlan
def graph_time(fn, it=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(it): fn()
    for _ in range(3): g.replay()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record(); g.replay(); e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / it * 1e3
for D, H in ((32, 24), (64, 12), (128, 6)):
    for S in (2048, 8192):
        q, k, v, _, do = make_inputs(4, H, S, S, D, torch.bfloat16, None, seed=1, strided=True)
        plan = AttentionPlan(q, k, v, do, sm_scale=D ** -0.5)
        plan.forward()
        f = 4.0 * 4 * H * S * S * D
        tf, tb = graph_time(plan.forward), graph_time(plan.backward)
        print(f"D={D:3d} H={H:2d} S={S}: fwd {tf:8.1f} us ({f/tf/1e6:6.1f} TF/s) | bwd {tb:8.1f} us ({2.5*f/tb/1e6:6.1f} TF/s)", flush=True)
