import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch, oracle
from attn_helpers import make_inputs, oracle_all, maxdiff
from flasht5_amd import flash_attention_v2_rpe
from flasht5_amd import positional_encoding as pe
from flasht5_amd.flash_attention_v2_bias import _attn_fwd, _attn_bwd
B,H,M,N,D=2,2,300,200,64
dtype=torch.float16
q,k,v,_,do = make_inputs(B,H,M,N,D,dtype,None,seed=M+3*N)
g = torch.Generator().manual_seed(5)
table = (torch.randn(32,H,generator=g)*0.5).cuda()
bias = pe.compute_bias(table,M,N,True,32,64).contiguous()
ref = oracle_all(q,k,v,bias,do,1.0,False)
rpe1d = pe.rpe1d_from_table(table,True,32,64)
for it in range(3):
    o,L = _attn_fwd(q,k,v,None,rpe1d,64,False,1.0)
    dq,dk,dv,d1 = _attn_bwd(o,do,q,k,v,None,rpe1d,64,L,False,1.0,True)
    torch.cuda.synchronize()
    print(it, "o",maxdiff(o,ref["o"]),"L",maxdiff(L,ref["L"]),"dq",maxdiff(dq,ref["dq"]),"dk",maxdiff(dk,ref["dk"]),"dv",maxdiff(dv,ref["dv"]))
    e=(dv.float()-ref["dv"]).abs()
    idx=torch.nonzero(e>0.1)
    print(" bad dv count",idx.shape[0], idx[:8].tolist())
    eL=(L-ref["L"]).abs(); print(" bad L", torch.nonzero(eL>1e-2)[:8].tolist(), L[0,0,:4].tolist(), ref["L"][0,0,:4].tolist())
# with oracle L,o
dq,dk,dv,d1 = _attn_bwd(ref["o"].to(dtype),do,q,k,v,None,rpe1d,64,ref["L"].contiguous(),False,1.0,True)
print("oracle o/L -> dv", maxdiff(dv,ref["dv"]))
