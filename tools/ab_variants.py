"""developer tool: two (or more) forced kernel variants of one attention stage, timed INTERLEAVED (A, B, A, B, ...) by graph replay --
the confirmation step behind a dispatch rule (tools/dispatch_audit.py measures variants one after the other).
    python tools/ab_variants.py --B 8 --H 12 --M 2048 --N 2048 --causal --mode none --what fwd --variants 0,2049,1025"""
import argparse, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from attn_helpers import make_inputs
from flasht5_amd.flash_attention_v2_bias import AttentionPlan
from flasht5_amd import positional_encoding as pe, _lib as L
ap = argparse.ArgumentParser()
ap.add_argument("--B", type=int, default=4); ap.add_argument("--H", type=int, default=12); ap.add_argument("--M", type=int, default=2048)
ap.add_argument("--N", type=int, default=2048); ap.add_argument("--causal", action="store_true"); ap.add_argument("--mode", default="none")
ap.add_argument("--what", default="fwd"); ap.add_argument("--variants", default="0"); ap.add_argument("--rounds", type=int, default=5)
a = ap.parse_args()
q, k, v, _, do = make_inputs(a.B, a.H, a.M, a.N, 64, torch.bfloat16, None, seed=1, strided=True)
table = (torch.randn(32, a.H, generator=torch.Generator().manual_seed(1)) * 0.5).cuda()
kw = dict(rpe1d=pe.rpe1d_from_table(table), radius=128) if a.mode == "rpe" else {}
stage = {"fwd": None, "dq": 1, "dkdv": 2, "bwd": 7}[a.what]
graphs = []
for bits in (int(x) for x in a.variants.split(",")):
    plan = AttentionPlan(q, k, v, do, causal=a.causal, sm_scale=0.125, variant=bits or None, **kw)
    plan.forward(); plan.backward(); torch.cuda.synchronize()
    fn = plan.forward if stage is None else (lambda p=plan: p.backward(stage))
    it = max(2, min(20, int(2e10 / (a.B * a.H * float(a.M) * a.N))))
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(it):
            fn()
    graphs.append((bits, plan, g, it, []))
t0 = time.perf_counter()
while time.perf_counter() - t0 < 0.2:
    for _, _, g, _, _ in graphs:
        g.replay()
torch.cuda.synchronize()
for _ in range(a.rounds):
    for bits, _, g, it, ts in graphs:
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(); g.replay(); e.record(); torch.cuda.synchronize()
        ts.append(s.elapsed_time(e) / it * 1e3)
for bits, plan, _, _, ts in graphs:
    d = L.describe(a.B, a.H, a.M, a.N, causal=a.causal, bias_mode=L.BIAS_RPE1D if a.mode == "rpe" else 0, radius=128 if a.mode == "rpe" else 0, variant=bits)
    print(f"variant {bits:6d}: min {min(ts):8.1f}  median {sorted(ts)[len(ts) // 2]:8.1f} us   {d}", flush=True)
