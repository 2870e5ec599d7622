"""per-workgroup timeline of the fused backward at cfg2 (needs the FAT5_TRACE=1 library variant `trace`).
usage (GPU box): FAT5_LIB_VARIANT=trace python tools/trace_bwd.py [mode]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch, numpy as np
from attn_helpers import make_inputs
from flasht5_amd.flash_attention_v2_bias import AttentionPlan
from flasht5_amd import positional_encoding as pe
mode = sys.argv[1] if len(sys.argv) > 1 else "rpe"
S = 512
q, k, v, _, do = make_inputs(4, 12, S, S, 64, torch.bfloat16, None, seed=1, strided=True)
kw = {}
if mode == "rpe":
    kw = dict(rpe1d=pe.rpe1d_from_table((torch.randn(32, 12) * 0.5).cuda()), radius=128)
plan = AttentionPlan(q, k, v, do, sm_scale=0.125, **kw)
plan.forward()
for _ in range(5): plan.backward(3)
torch.cuda.synchronize()
plan.ws.zero_(); torch.cuda.synchronize()
plan.backward(3); torch.cuda.synchronize()
# delta scratch is the first region of the workspace
n = plan.bwd_launches()
raw = plan.ws[: 384 * 16 * 8].view(torch.int64).cpu().numpy().reshape(384, 16)  # delta scratch = first region of the workspace
t0 = raw[raw[:, 0] > 0, 0].min()
def show(name, rows):
    r = rows.astype(np.float64)
    ok = r[:, 0] > 0
    r = r[ok]
    rel = (r - t0)
    rel[r == 0] = np.nan
    cols = [0, 1] + list(range(2, 10)) + [14, 15]
    med = np.nanmedian(rel[:, cols], axis=0)
    print(f"{name}: n={len(r)}  (median cycles since first workgroup start; s_memtime ticks)")
    print("   start %7.0f | prologue done %7.0f | tiles " % (med[0], med[1]) + " ".join("%7.0f" % x for x in med[2:10]) + " | out %7.0f | end %7.0f" % (med[10], med[11]))
    dur = rel[:, 15] - rel[:, 0]
    print("   duration: median %.0f  max %.0f ;  prologue %.0f ; per tile %.0f ; epilogue %.0f" % (
        np.nanmedian(dur), np.nanmax(dur), np.nanmedian(rel[:, 1] - rel[:, 0]),
        np.nanmedian((rel[:, 9] - rel[:, 1]) / 8), np.nanmedian(rel[:, 15] - rel[:, 9])))
    print("   last end over all: %.0f" % np.nanmax(rel[:, 15]))
show("dK/dV workgroups", raw[:192])
show("dQ workgroups   ", raw[192:])
