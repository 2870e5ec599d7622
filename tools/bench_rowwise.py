"""RMSNorm / cross-entropy bandwidth benchmark (bandwidth-bound siblings of the attention path).
Algorithmic bytes per SURVEY 8(d): RMSNorm fwd 2*R*N*e, bwd 3*R*N*e; CE fwd R*V*e, bwd 2*R*V*e."""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from flasht5_amd.rms_norm import rmsnorm_fwd, rmsnorm_bwd
from flasht5_amd.cross_entropy_loss import cross_entropy_fwd, cross_entropy_bwd

def ev(fn, it=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    s,e=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(it): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e)/it*1e-3  # seconds

def gv(fn, it=50):
    """kernel-side time: the same calls captured in a HIP graph (no Python / ctypes / allocator time)"""
    for _ in range(3): fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(it): fn()
    g.replay(); torch.cuda.synchronize()
    s,e=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    s.record(); g.replay(); e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e)/it*1e-3

out = {}
for rows, n in ((4096, 768), (16384, 768), (16384, 1024), (65536, 1024)):
    x = torch.randn(rows, n, device="cuda").bfloat16(); w = torch.ones(n, device="cuda").bfloat16(); dy = torch.randn_like(x)
    y, rstd = rmsnorm_fwd(x, w, 1e-6)
    tf = ev(lambda: rmsnorm_fwd(x, w, 1e-6)); tb = ev(lambda: rmsnorm_bwd(dy, x, w, rstd, 1e-6))
    out[f"rmsnorm_{rows}x{n}"] = {"fwd_us": round(tf*1e6,1), "fwd_GBs": round(2*rows*n*2/tf/1e9,1), "bwd_us": round(tb*1e6,1), "bwd_GBs": round(3*rows*n*2/tb/1e9,1)}
    gf = gv(lambda: rmsnorm_fwd(x, w, 1e-6)); gb = gv(lambda: rmsnorm_bwd(dy, x, w, rstd, 1e-6))
    out[f"rmsnorm_{rows}x{n}"].update({"fwd_graph_us": round(gf*1e6,1), "bwd_graph_us": round(gb*1e6,1)})
    print(f"rmsnorm ({rows},{n}) bf16: fwd {tf*1e6:7.1f} us {2*rows*n*2/tf/1e9:7.1f} GB/s (graph {gf*1e6:6.1f} us {2*rows*n*2/gf/1e9:7.1f} GB/s) | bwd {tb*1e6:7.1f} us {3*rows*n*2/tb/1e9:7.1f} GB/s (graph {gb*1e6:6.1f} us {3*rows*n*2/gb/1e9:7.1f} GB/s)", flush=True)
for rows, V in ((4096, 32768), (16384, 32768), (16384, 32128)):
    lg = torch.randn(rows, V, device="cuda").bfloat16(); lab = torch.randint(0, V, (rows,), device="cuda"); dl = torch.randn(rows, device="cuda")
    l, z, lse = cross_entropy_fwd(lg, lab, None, 0.0, 1.0, 1e-4, -100)
    tf = ev(lambda: cross_entropy_fwd(lg, lab, None, 0.0, 1.0, 1e-4, -100), 20)
    tb = ev(lambda: cross_entropy_bwd(dl, lg, lse, lab, False, 0.0, 1.0, 1e-4, -100), 20)
    tbi = ev(lambda: cross_entropy_bwd(dl, lg, lse, lab, True, 0.0, 1.0, 1e-4, -100), 20)
    out[f"ce_{rows}x{V}"] = {"fwd_us": round(tf*1e6,1), "fwd_GBs": round(rows*V*2/tf/1e9,1), "bwd_us": round(tb*1e6,1), "bwd_GBs": round(2*rows*V*2/tb/1e9,1), "bwd_inplace_us": round(tbi*1e6,1)}
    print(f"ce ({rows},{V}) bf16: fwd {tf*1e6:7.1f} us {rows*V*2/tf/1e9:7.1f} GB/s | bwd {tb*1e6:7.1f} us {2*rows*V*2/tb/1e9:7.1f} GB/s | bwd inplace {tbi*1e6:7.1f} us {2*rows*V*2/tbi/1e9:7.1f} GB/s", flush=True)
print(json.dumps(out))
