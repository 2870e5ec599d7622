"""developer timing of the dK/dV stage alone (stage-split backward) at (4,12,S,64)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from attn_helpers import make_inputs
from flasht5_amd.flash_attention_v2_bias import AttentionPlan
from flasht5_amd import positional_encoding as pe

S = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
modes = sys.argv[2].split(",") if len(sys.argv) > 2 else ["none", "rpe"]
stage = int(sys.argv[3]) if len(sys.argv) > 3 else 2
need = int(sys.argv[4]) if len(sys.argv) > 4 else 1      # 0: no table gradient
radius = int(sys.argv[5]) if len(sys.argv) > 5 else 128
for mode in modes:
    q, k, v, _, do = make_inputs(4, 12, S, S, 64, torch.bfloat16, None, seed=1, strided=True)
    table = (torch.randn(32, 12) * 0.5).cuda()
    kw = dict(rpe1d=pe.rpe1d_from_table(table, max_distance=radius), radius=radius, need_dbias=bool(need)) if mode == "rpe" else {}
    plan = AttentionPlan(q, k, v, do, sm_scale=0.125, **kw)
    plan.forward(); plan.backward()
    for _ in range(3): plan.backward(stage)
    torch.cuda.synchronize()
    best = 1e9
    for rep in range(3):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(10): plan.backward(stage)
        e.record(); torch.cuda.synchronize()
        best = min(best, s.elapsed_time(e) / 10)
    print(f"  stage {stage} S={S} {mode} need={need} R={radius}: {best*1e3:8.1f} us", flush=True)
