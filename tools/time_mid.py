import os, sys
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import torch
from attn_helpers import make_inputs
from flasht5_amd.flash_attention_v2_bias import AttentionPlan
from flasht5_amd import positional_encoding as pe
def t(S, mode, what, iters=30):
    q, k, v, _, do = make_inputs(4, 12, S, S, 64, torch.bfloat16, None, seed=1, strided=True)
    table = (torch.randn(32, 12) * 0.5).cuda()
    kw = dict(rpe1d=pe.rpe1d_from_table(table), radius=128) if mode == "rpe" else {}
    plan = AttentionPlan(q, k, v, do, sm_scale=0.125, **kw)
    fn = plan.forward if what == "fwd" else plan.backward
    plan.forward()
    for _ in range(5): fn()
    torch.cuda.synchronize()
    best = 1e9
    for rep in range(3):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(iters): fn()
        e.record(); torch.cuda.synchronize()
        best = min(best, s.elapsed_time(e) / iters * 1e3)
    return best
SIZES = tuple(int(x) for x in sys.argv[1].split(',')) if len(sys.argv) > 1 else (1024, 2048, 4096)
for S in SIZES:
    print(f"S={S}: " + " | ".join(f"{w} {m} {t(S, m, w):7.1f}" for w in ("fwd", "bwd") for m in ("none", "rpe")), flush=True)
