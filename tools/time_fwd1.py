"""single-shape fwd timing (developer tool): python tools/time_fwd1.py [S] [mode]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from attn_helpers import make_inputs
from flasht5_amd.flash_attention_v2_bias import AttentionPlan
from flasht5_amd import positional_encoding as pe
S = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
mode = sys.argv[2] if len(sys.argv) > 2 else "none"
q, k, v, _, do = make_inputs(4, 12, S, S, 64, torch.bfloat16, None, seed=1, strided=True)
kw = {}
if mode == "rpe":
    kw = dict(rpe1d=pe.rpe1d_from_table((torch.randn(32, 12) * 0.5).cuda()), radius=128)
plan = AttentionPlan(q, k, v, do, sm_scale=0.125, **kw)
for _ in range(3): plan.forward()
torch.cuda.synchronize()
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record()
for _ in range(20): plan.forward()
e.record(); torch.cuda.synchronize()
ms = s.elapsed_time(e) / 20
print(f"  fwd S={S} {mode}: {ms*1e3:8.1f} us {4.0*4*12*S*S*64/ms/1e9:7.1f} TF/s", flush=True)
