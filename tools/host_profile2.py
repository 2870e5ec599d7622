"""cProfile of the eager drop-in path at cfg2 (developer tool): where the host time of one fwd+bwd step goes."""
import cProfile, pstats, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from flasht5_amd import flash_attention_v2_rpe
B, H, S, D = 4, 12, 512, 64
g = torch.Generator().manual_seed(0)
mk = lambda: torch.randn(B, S, H, D, generator=g).bfloat16().cuda().permute(0, 2, 1, 3).requires_grad_()
q, k, v = mk(), mk(), mk()
do = torch.randn(B, S, H, D, generator=g).bfloat16().cuda().permute(0, 2, 1, 3)
table = (torch.randn(32, H, generator=g) * 0.5).cuda().requires_grad_()
def step():
    o = flash_attention_v2_rpe(q, k, v, table, True, 32, 128, False, 0.125)
    return torch.autograd.grad(o, (q, k, v, table), do)
for _ in range(100): step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(500): step()
print("host us/step: %.1f" % ((time.perf_counter() - t0) / 500 * 1e6))
torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable()
for _ in range(500): step()
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("tottime").print_stats(22)
