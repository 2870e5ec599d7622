import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch, oracle
from attn_helpers import make_inputs, oracle_all, maxdiff
from flasht5_amd.flash_attention_v2_bias import _attn_fwd, _attn_bwd
def run(B,H,M,N,causal,scale=1.0):
    q,k,v,_,do = make_inputs(B,H,M,N,64,torch.bfloat16,None,seed=M*7+N)
    ref = oracle_all(q,k,v,None,do,scale,causal)
    o,L = _attn_fwd(q,k,v,None,None,0,causal,scale)
    dq,dk,dv,_ = _attn_bwd(o,do,q,k,v,None,None,0,L,causal,scale,False)
    torch.cuda.synchronize()
    ek=(dk.float()-ref["dk"]).abs().amax(-1)  # (B,H,N)
    ev=(dv.float()-ref["dv"]).abs().amax(-1)
    print(f"M{M} N{N} causal={causal}: dk err per key (b0,h0):", [round(x,2) for x in ek[0,0].tolist()])
    print(f"   dv err per key (b0,h0):", [round(x,2) for x in ev[0,0].tolist()])
run(1,2,80,80,True,0.125)
run(1,2,64,64,True,0.125)
run(1,2,128,128,True,0.125)
