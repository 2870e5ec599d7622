import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from flasht5_amd.rms_norm import rmsnorm_fwd, rmsnorm_bwd
for rows, n in ((4096, 768), (65536, 1024)):
    x = torch.randn(rows, n, device="cuda").bfloat16(); w = torch.ones(n, device="cuda").bfloat16(); dy = torch.randn_like(x)
    y, rstd = rmsnorm_fwd(x, w, 1e-6)
    for _ in range(20): rmsnorm_bwd(dy, x, w, rstd, 1e-6)
    torch.cuda.synchronize()
