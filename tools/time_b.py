import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from attn_helpers import make_inputs
from flasht5_amd.flash_attention_v2_bias import AttentionPlan
def ev(fn, it=30):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s,e=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(it): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e)/it*1e3
S=512
for B in (1,2,4,8,16,32):
    q,k,v,_,do = make_inputs(B,12,S,S,64,torch.bfloat16,None,seed=1,strided=True)
    plan=AttentionPlan(q,k,v,do,sm_scale=0.125)
    plan.forward()
    print(f"B={B:2d} S={S} none: fwd {ev(plan.forward):7.1f} dq {ev(lambda: plan.backward(1)):7.1f} dkdv {ev(lambda: plan.backward(2)):7.1f} us", flush=True)
