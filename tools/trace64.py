"""Per-workgroup timeline of the 64-wide backward bodies (developer tool; needs the FAT5_TRACE=1 library variant `trace`:
  cd flasht5_amd/lib && cp -r obj obj_trace && rm obj_trace/attn_bwd64_d64.o; FAT5_VARIANT=trace FAT5_EXTRA_FLAGS=-DFAT5_TRACE=1 python flasht5_amd/build.py
usage (GPU box): FAT5_LIB_VARIANT=trace python tools/trace64.py [--S 512] [--mode rpe] [--variant BITS] [--stage 1|2]
Thread 0 of every workgroup stamps s_memtime at: 0 entry, 1 prologue done (operands loaded, first steps in LDS), 2 first scores, 3 main loop done,
4 drain done, 5 partial sums out (dK/dV), 6 outputs stored.  Printed: medians relative to the first workgroup's entry, in shader-clock ticks."""
import argparse, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from attn_helpers import make_inputs
from flasht5_amd.flash_attention_v2_bias import AttentionPlan
from flasht5_amd import positional_encoding as pe
ap = argparse.ArgumentParser()
ap.add_argument("--S", type=int, default=512); ap.add_argument("--mode", default="rpe"); ap.add_argument("--variant", type=int, default=40980)
ap.add_argument("--stage", type=int, default=2)
a = ap.parse_args()
q, k, v, _, do = make_inputs(4, 12, a.S, a.S, 64, torch.bfloat16, None, seed=1, strided=True)
kw = {}
if a.mode == "rpe":
    kw = dict(rpe1d=pe.rpe1d_from_table((torch.randn(32, 12) * 0.5).cuda()), radius=128)
plan = AttentionPlan(q, k, v, do, sm_scale=0.125, variant=a.variant, **kw)
plan.forward()
for _ in range(5):
    plan.backward(a.stage)
torch.cuda.synchronize()
plan.ws.zero_(); torch.cuda.synchronize()
plan.backward(1); torch.cuda.synchronize()   # (the dQ stage leaves the row statistics; its stamps are overwritten unless it is the traced stage)
if a.stage != 1:
    plan.ws[: 4096 * 16 * 8].zero_(); torch.cuda.synchronize()
    plan.backward(a.stage); torch.cuda.synchronize()
nwg = plan.delta_bytes // 128 if hasattr(plan, "delta_bytes") else (4 * 12 * a.S * 4) // 128   # the delta scratch holds 16 int64 per workgroup
raw = plan.ws[: nwg * 16 * 8].view(torch.int64).cpu().numpy().reshape(-1, 16)
def show(rows, title):
    rows = rows[(rows[:, 0] > 0) & (rows[:, 6] > rows[:, 0]) & (rows[:, 6] - rows[:, 0] < 10 ** 9)]
    if len(rows) == 0:
        return
    names = ["entry", "prologue", "first scores", "loop", "drain", "partials", "stored"]
    cols = [0, 1, 2, 3, 4, 5, 6] if rows[:, 5].min() > 0 else [0, 1, 2, 3, 4, 6]
    d = np.diff(rows[:, cols].astype(np.float64), axis=1)   # (s_memtime counters of different XCDs are not synchronised: per-workgroup differences only)
    print(f"{title}: {len(rows)} workgroups; per-workgroup phase lengths in shader-clock ticks, median (max):")
    for c_, col in zip(cols[1:], d.T):
        print(f"   -> {names[c_]:13s} {np.median(col):8.0f} ({col.max():8.0f})")
    if rows[:, 9].min() > 0:
        for n_, k_ in (("arguments decoded", 10), ("staging DMAs issued", 11), ("DMAs issued", 7), ("table filled", 8), ("landed + barrier", 9)):
            if rows[:, k_].min() <= 0:
                continue
            print(f"   prologue detail: entry -> {n_:18s} {np.median(rows[:, k_] - rows[:, 0]):8.0f}")
    for n_, k_ in (("dQ rows on their way", 12), ("diagonal arrays complete (barrier)", 13)):
        if rows[:, k_].min() > 0:
            print(f"   epilogue detail: drain -> {n_:34s} {np.median(rows[:, k_] - rows[:, 4]):8.0f}")
    tot = (rows[:, 6] - rows[:, 0]).astype(np.float64)
    print(f"   workgroup duration: median {np.median(tot):.0f}  max {tot.max():.0f}")


head = f"S={a.S} {a.mode} variant={a.variant} stage={a.stage}"
if a.stage == 3 and plan.bwd_launches() == 1:
    nkv = 4 * 12 * ((a.S + 255) // 256)
    show(raw[:nkv], head + " fused launch, dK/dV workgroups")
    show(raw[nkv:], head + " fused launch, dQ workgroups")
else:
    show(raw, head)
