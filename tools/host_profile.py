import os, sys, cProfile, pstats
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from attn_helpers import make_inputs
from flasht5_amd import flash_attention_v2_rpe
q, k, v, b, do = make_inputs(4, 12, 512, 512, 64, torch.bfloat16, None, seed=1)
table = (torch.randn(32, 12) * 0.5).cuda().requires_grad_()
ql, kl, vl = (t.clone().requires_grad_() for t in (q, k, v))
def fb():
    o = flash_attention_v2_rpe(ql, kl, vl, table, True, 32, 128, False, 0.125)
    o.backward(do)
for _ in range(20): fb()
torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable()
for _ in range(200): fb()
pr.disable(); torch.cuda.synchronize()
st = pstats.Stats(pr); st.sort_stats("cumulative").print_stats(28)
