import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch, oracle
from attn_helpers import make_inputs, oracle_all, maxdiff
from flasht5_amd.flash_attention_v2_bias import _attn_fwd, _attn_bwd
def run(B,H,M,N,causal,scale=0.125, bias=False):
    q,k,v,b,do = make_inputs(B,H,M,N,64,torch.bfloat16,"1h" if bias else None,seed=M*7+N)
    if bias: b = b*0
    ref = oracle_all(q,k,v,b,do,scale,causal)
    o,L = _attn_fwd(q,k,v,b,None,0,causal,scale)
    dq,dk,dv,_ = _attn_bwd(o,do,q,k,v,b,None,0,L,causal,scale,False)
    torch.cuda.synchronize()
    print(f"M{M} N{N} causal={int(causal)} bias={int(bias)}: dq {maxdiff(dq,ref['dq']):.3f} dk {maxdiff(dk,ref['dk']):.3f} dv {maxdiff(dv,ref['dv']):.3f}")
for c in (False, True):
    for bias in (False, True):
        run(1,1,64,64,c,bias=bias)
        run(1,2,128,128,c,bias=bias)
        run(2,2,200,136,c,bias=bias)
