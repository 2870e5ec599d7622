"""Run one attention kernel configuration a few times (for rocprofv3 --pmc / --kernel-trace runs).
usage: python tools/run_one.py --S 8192 --mode none|rpe|dense --what fwd|bwd|both --iters 3 [--D 64] [--causal]"""
import argparse, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from attn_helpers import make_inputs
from flasht5_amd.flash_attention_v2_bias import AttentionPlan
from flasht5_amd import positional_encoding as pe

ap = argparse.ArgumentParser()
ap.add_argument("--S", type=int, default=8192); ap.add_argument("--mode", default="none"); ap.add_argument("--what", default="fwd")
ap.add_argument("--iters", type=int, default=3); ap.add_argument("--D", type=int, default=64); ap.add_argument("--causal", action="store_true")
ap.add_argument("--B", type=int, default=4); ap.add_argument("--H", type=int, default=12)
ap.add_argument("--variant", type=int, default=0)  # fat5_variant bits (include/fat5.h)
ap.add_argument("--dtype", default="bf16")
ap.add_argument("--seconds", type=float, default=0.0)  # wall-clock pre-warm with the same launches before the timed ones (bench.py's: the first
                                                       # milliseconds after an idle gap run at the idle clock; rocprofv3 averages then cover warm launches)
a = ap.parse_args()
DT = torch.bfloat16 if a.dtype == "bf16" else torch.float16
q, k, v, _, do = make_inputs(a.B, a.H, a.S, a.S, a.D, DT, None, seed=1, strided=True)
table = (torch.randn(32, a.H) * 0.5).cuda()
kw = {}
if a.mode == "rpe": kw = dict(rpe1d=pe.rpe1d_from_table(table), radius=128)
elif a.mode == "dense": kw = dict(bias=pe.compute_bias(table, a.S, a.S).to(DT).contiguous())
plan = AttentionPlan(q, k, v, do, causal=a.causal, sm_scale=0.125, variant=a.variant or None, **kw)
plan.forward(); torch.cuda.synchronize()
import time
t_pre = time.perf_counter()
while time.perf_counter() - t_pre < a.seconds:
    for _ in range(3):
        if a.what in ("fwd", "both"): plan.forward()
        if a.what in ("bwd", "both"): plan.backward()
    torch.cuda.synchronize()
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record()
for _ in range(a.iters):
    if a.what in ("fwd", "both"): plan.forward()
    if a.what in ("bwd", "both"): plan.backward()
e.record(); torch.cuda.synchronize()
print(f"{a.what} S={a.S} mode={a.mode}: {s.elapsed_time(e)/a.iters*1e3:.1f} us/iter")
