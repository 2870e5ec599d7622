import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from attn_helpers import make_inputs
from flasht5_amd.flash_attention_v2_bias import AttentionPlan
from flasht5_amd import positional_encoding as pe
def ev(fn, it=30):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s,e=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(it): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e)/it*1e3
for S in (512, 2048):
    q,k,v,_,do = make_inputs(4,12,S,S,64,torch.bfloat16,None,seed=1,strided=True)
    table=(torch.randn(32,12)*0.5).cuda()
    r1=pe.rpe1d_from_table(table)
    for name,kw in (("none",{}),("rpe+grad",dict(rpe1d=r1,radius=128)),("rpe nograd",dict(rpe1d=r1,radius=128,need_dbias=False)),("rpe R=16 +grad",dict(rpe1d=pe.rpe1d_from_table(table,True,32,16),radius=16))):
        plan=AttentionPlan(q,k,v,do,sm_scale=0.125,**kw)
        plan.forward()
        print(f"S={S} {name:16s}: fwd {ev(plan.forward):7.1f} dq {ev(lambda: plan.backward(1)):7.1f} dkdv {ev(lambda: plan.backward(2)):7.1f} red {ev(lambda: plan.backward(4)):6.1f} us", flush=True)
