"""host-side cost of the eager drop-in path at cfg2 (developer tool): wall clock of each part of
flash_attention_v2_rpe(...) + autograd.grad, and the torch profiler's CPU-op table for one step."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from flasht5_amd import flash_attention_v2_rpe, _lib
from flasht5_amd import positional_encoding as pe
B, H, S, D = 4, 12, 512, 64
dev = "cuda"
g = torch.Generator().manual_seed(0)
mk = lambda: torch.randn(B, S, H, D, generator=g).to(torch.bfloat16).to(dev).permute(0, 2, 1, 3).requires_grad_()
q, k, v = mk(), mk(), mk()
do = torch.randn(B, S, H, D, generator=g).to(torch.bfloat16).to(dev).permute(0, 2, 1, 3)
table = (torch.randn(32, H, generator=g) * 0.5).to(dev).requires_grad_()
nat = _lib.native()
idx = pe.bucket_index32(128, True, 32, 128, q.device)


def t(fn, n=2000):
    for _ in range(100):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    th = time.perf_counter() - t0
    torch.cuda.synchronize()
    return th / n * 1e6, (time.perf_counter() - t0) / n * 1e6


def full():
    o = flash_attention_v2_rpe(q, k, v, table, True, 32, 128, False, 0.125)
    return torch.autograd.grad(o, (q, k, v, table), do)


def fwd_only():
    return flash_attention_v2_rpe(q, k, v, table, True, 32, 128, False, 0.125)


def fwd_nograd():
    with torch.no_grad():
        return flash_attention_v2_rpe(q, k, v, table, True, 32, 128, False, 0.125)


def raw_fwd():
    r1 = nat.rpe1d_of(table.detach(), idx, 128, 32)
    return nat.attn_fwd(q.detach(), k.detach(), v.detach(), None, r1, 128, False, 0.125)


r1 = nat.rpe1d_of(table.detach(), idx, 128, 32)
o, L = nat.attn_fwd(q.detach(), k.detach(), v.detach(), None, r1, 128, False, 0.125)
qd, kd, vd = q.detach(), k.detach(), v.detach()


def raw_bwd():
    return nat.attn_bwd(o, do, qd, kd, vd, None, r1, 128, L, False, 0.125, True, idx, 32)


def empty8():
    return [torch.empty_like(qd) for _ in range(8)]


for name, fn in (("full step", full), ("forward (grad mode)", fwd_only), ("forward (no_grad)", fwd_nograd), ("raw rpe1d_of + attn_fwd", raw_fwd),
                 ("raw attn_bwd", raw_bwd), ("8 x empty_like", empty8)):
    h, w = t(fn)
    print(f"{name:28s} host {h:7.1f} us   wall {w:7.1f} us", flush=True)

from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    for _ in range(20):
        full()
    torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="self_cpu_time_total", row_limit=30, max_name_column_width=60))
