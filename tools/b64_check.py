"""developer check: the two dK/dV bodies side by side (per-diagonal sums, dk, dv) on one problem."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from flasht5_amd.flash_attention_v2_bias import AttentionPlan
from flasht5_amd import positional_encoding as pe
from attn_helpers import make_inputs

B, H, M, N, md, causal = [int(x) for x in sys.argv[1:7]]
q, k, v, _, do = make_inputs(B, H, M, N, 64, torch.bfloat16, None, seed=M + 5 * N)
table = (torch.randn(32, H, generator=torch.Generator().manual_seed(1)) * 0.5).cuda()
R = md
outs = []
for f in ("0", "1"):
    os.environ["FAT5_BWD64"] = f
    plan = AttentionPlan(q, k, v, do, sm_scale=0.125, causal=bool(causal), need_dbias=True, rpe1d=pe.rpe1d_from_table(table, True, 32, md), radius=R)
    plan.forward(); plan.backward(); torch.cuda.synchronize()
    outs.append((plan.dk.float().clone(), plan.dv.float().clone(), plan.dbias.clone()))
for i, n in enumerate(("dk", "dv", "drpe1d")):
    d = (outs[0][i] - outs[1][i]).abs()
    print(n, "maxdiff", d.max().item(), "max", outs[0][i].abs().max().item())
d = (outs[0][2] - outs[1][2])
print("drpe1d shape", tuple(d.shape))
for h in range(H):
    idx = d[h].abs().argsort(descending=True)[:8]
    print("h", h, [(int(i) - R, round(float(d[h, i]), 4), round(float(outs[0][2][h, i]), 3)) for i in idx])
