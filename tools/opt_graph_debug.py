"""developer probe: AdamWScale.step() alone in a graph"""
import os, sys, faulthandler
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from flasht5_amd import AdamWScale
torch.manual_seed(0)
ps = [torch.nn.Parameter(torch.randn(n, device="cuda").bfloat16()) for n in (1000, 70000, 8192, 33)]
opt = AdamWScale(ps, lr=1e-3, kahan_sum=True, max_grad_norm=1.0 if "--noclip" not in sys.argv else None)
for p in ps:
    p.grad = torch.randn_like(p)
opt.step()
torch.cuda.synchronize()
print("eager ok", float(ps[0][0]), flush=True)
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    opt.step()
print("captured", flush=True)
torch.cuda.synchronize()
opt.graph_advance()
torch.cuda.synchronize()
print("advanced", flush=True)
for i in range(3):
    g.replay()
    torch.cuda.synchronize()
    print("replay", i, float(ps[0][0]), flush=True)
    opt.graph_advance()
