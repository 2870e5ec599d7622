"""Run fat5_linear_fused a few times on one shape (for rocprofv3 --pmc / --kernel-trace runs).
usage: python tools/run_lin.py --M 4096 --N 4096 --K 768 [--res] [--iters 5]"""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from flasht5_amd import _lib
ap = argparse.ArgumentParser()
ap.add_argument("--M", type=int, default=4096); ap.add_argument("--N", type=int, default=4096); ap.add_argument("--K", type=int, default=768)
ap.add_argument("--res", action="store_true"); ap.add_argument("--iters", type=int, default=5)
a = ap.parse_args()
lib = _lib.load(); dev = torch.device("cuda")
x = torch.randn(a.M, a.K, device=dev).bfloat16(); W = (torch.randn(a.N, a.K, device=dev) / a.K ** 0.5).bfloat16()
res = torch.randn(a.M, a.N, device=dev).bfloat16(); out = torch.empty(a.M, a.N, device=dev, dtype=torch.bfloat16); rstd = torch.empty(a.M, device=dev)
for _ in range(a.iters):
    _lib.check(lib.fat5_linear_fused(x.data_ptr(), W.data_ptr(), res.data_ptr() if a.res else None, out.data_ptr(), None if a.res else rstd.data_ptr(), a.M, a.N, a.K,
                                     a.K, a.K, a.N, a.N, 0 if a.res else 1, 1e-6, _lib.dtype_code(x.dtype), _lib.stream_ptr(dev)), "lin")
torch.cuda.synchronize()
print("done")
