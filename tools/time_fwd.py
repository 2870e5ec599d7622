"""fwd-only timing sweep (developer tool)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from attn_helpers import make_inputs
from flasht5_amd.flash_attention_v2_bias import AttentionPlan
from flasht5_amd import positional_encoding as pe

def t(S, mode, what="fwd", D=64, iters=20):
    q, k, v, _, do = make_inputs(4, 12, S, S, D, torch.bfloat16, None, seed=1, strided=True)
    table = (torch.randn(32, 12) * 0.5).cuda()
    kw = {}
    if mode == "rpe": kw = dict(rpe1d=pe.rpe1d_from_table(table), radius=128)
    elif mode == "dense": kw = dict(bias=pe.compute_bias(table, S, S).to(torch.bfloat16).contiguous())
    plan = AttentionPlan(q, k, v, do, sm_scale=0.125, **kw)
    fn = plan.forward if what == "fwd" else plan.backward
    plan.forward()
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    ms = s.elapsed_time(e) / iters
    fl = 4.0 * 4 * 12 * S * S * D * (1.0 if what == "fwd" else 2.5)
    return ms * 1e3, fl / ms / 1e9

what = sys.argv[1] if len(sys.argv) > 1 else "fwd"
modes = sys.argv[2].split(",") if len(sys.argv) > 2 else ["none", "rpe"]
for S in (512, 2048, 8192):
    print(f"  {what} S={S}: " + " | ".join(f"{m} {t(S, m, what)[0]:8.1f} us {t(S, m, what)[1]:7.1f} TF/s" for m in modes), flush=True)
