"""developer timing: forward at (4,12,S,64) in the T5-bias mode for several radii (how much a band tile costs over a pipelined one)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from attn_helpers import make_inputs
from flasht5_amd.flash_attention_v2_bias import AttentionPlan
from flasht5_amd import positional_encoding as pe
S = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
q, k, v, _, do = make_inputs(4, 12, S, S, 64, torch.bfloat16, None, seed=1, strided=True)
table = (torch.randn(32, 12) * 0.5).cuda()
def t(plan):
    for _ in range(5): plan.forward()
    torch.cuda.synchronize()
    best = 1e9
    for rep in range(3):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(10): plan.forward()
        e.record(); torch.cuda.synchronize()
        best = min(best, s.elapsed_time(e) / 10)
    return best * 1e3
print(f"S={S} none: {t(AttentionPlan(q, k, v, do, sm_scale=0.125)):8.1f} us", flush=True)
for R in (32, 64, 128, 256, 512):
    plan = AttentionPlan(q, k, v, do, sm_scale=0.125, rpe1d=pe.rpe1d_from_table(table, max_distance=R), radius=R, need_dbias=False)
    print(f"S={S} rpe R={R}: {t(plan):8.1f} us", flush=True)
