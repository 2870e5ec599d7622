"""Per-workgroup timeline of the pipelined forward (developer tool; needs the FAT5_TRACE=1 variant of the forward unit:
  python tools/build_variant.py ftrace attn_fwd64_d64.o "-DFAT5_TRACE=1";  FAT5_LIB_VARIANT=ftrace python tools/trace_fwd64.py [--S 512] [--mode rpe]
Thread 0 of every workgroup stamps s_memtime at: 0 entry, 1 arguments decoded / Q on its way, 2 first K / V tiles and the bias table requested,
3 landed (barrier), 4 loop entered, 5 sweep done, 6 halves merged, 7 rows stored -- and leaves the eight stamps in the first O row of the workgroup."""
import argparse, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from attn_helpers import make_inputs
from flasht5_amd.flash_attention_v2_bias import AttentionPlan
from flasht5_amd import positional_encoding as pe
ap = argparse.ArgumentParser()
ap.add_argument("--S", type=int, default=512); ap.add_argument("--mode", default="rpe"); ap.add_argument("--variant", type=int, default=0)
a = ap.parse_args()
B, H = 4, 12
q, k, v, _, do = make_inputs(B, H, a.S, a.S, 64, torch.bfloat16, None, seed=1, strided=True)
kw = dict(rpe1d=pe.rpe1d_from_table((torch.randn(32, H) * 0.5).cuda()), radius=128) if a.mode == "rpe" else {}
plan = AttentionPlan(q, k, v, do, sm_scale=0.125, variant=a.variant, **kw)
for _ in range(5):
    plan.forward()
torch.cuda.synchronize()
desc = plan.describe() if hasattr(plan, "describe") else {}
rows_wg = 128 if "ksplit" in str(desc) else 256
o = plan.o  # (B, H, S, D) view of (B, S, H, D) storage
rows = []
for b in range(B):
    for h in range(H):
        for m0 in range(0, a.S, rows_wg):
            rows.append(o[b, h, m0].contiguous().view(torch.int64).cpu().numpy()[:8])
r = np.stack(rows).astype(np.float64)
d = np.diff(r, axis=1)
names = ["arguments", "requests issued", "landed + barrier", "addresses / setup", "sweep", "merge", "store"]
print(f"S={a.S} {a.mode} {desc}: {len(rows)} workgroups of {rows_wg} rows; phase lengths in shader-clock ticks, median (max)")
for n, col in zip(names, d.T):
    print(f"   {n:20s} {np.median(col):8.0f} ({col.max():8.0f})")
tot = r[:, 7] - r[:, 0]
print(f"   workgroup duration   {np.median(tot):8.0f} ({tot.max():8.0f})")
