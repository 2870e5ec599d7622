import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch, oracle
from attn_helpers import make_inputs, oracle_all, maxdiff
from flasht5_amd import positional_encoding as pe
from flasht5_amd.flash_attention_v2_bias import _attn_fwd, _attn_bwd
def run(B,H,M,N,causal,need,scale=1.0,zero_table=False):
    q,k,v,_,do = make_inputs(B,H,M,N,64,torch.bfloat16,None,seed=M+3*N)
    g = torch.Generator().manual_seed(5)
    table = (torch.randn(32,H,generator=g)*0.5).cuda()
    if zero_table: table = table*0
    bias = pe.compute_bias(table,M,N,True,32,128).contiguous()
    ref = oracle_all(q,k,v,bias,do,scale,causal)
    rpe1d = pe.rpe1d_from_table(table,True,32,128)
    o,L = _attn_fwd(q,k,v,None,rpe1d,128,causal,scale)
    dq,dk,dv,d1 = _attn_bwd(o,do,q,k,v,None,rpe1d,128,L,causal,scale,need)
    torch.cuda.synchronize()
    ek=(dk.float()-ref["dk"]).abs().amax(-1)
    print(f"M{M} N{N} causal={int(causal)} need_dbias={int(need)} zero={int(zero_table)}: dq {maxdiff(dq,ref['dq']):.3f} dk {maxdiff(dk,ref['dk']):.3f} dv {maxdiff(dv,ref['dv']):.3f}")
    print("   dk err/key:", [round(x,1) for x in ek[0,0].tolist()][:128])
run(1,1,128,128,False,True)
run(1,1,128,128,False,False)
run(1,1,128,128,False,False,zero_table=True)
run(1,1,128,64,False,False)
run(1,1,64,128,False,False)
