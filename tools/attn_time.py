"""One timing tool for the attention path (developer tool; replaces the per-question time_*.py scripts of rounds 1-2).

    python tools/attn_time.py [--S 512,2048,8192] [--modes none,rpe,dense] [--what fwd,bwd,dq,dkdv,red] [--variant BITS]
                           [--B 4 --H 12 --D 64] [--dtype bf16|fp16] [--causal] [--radius 128] [--no-dbias] [--iters 20] [--reps 5]

Per (S, mode, stage): min / median / max over --reps event-timed batches of --iters launches (us; the batch is captured in a HIP graph
and replayed -- kernel-side time, launch gaps included, no Python / ctypes / hipFuncSetAttribute time: a Python loop is host-bound
below ~25 us per launch; --eager times the Python loop instead) and TFLOP/s of the median by the
reference's FLOP model (benchmarks/bench_fa2_bias.py:10-13: fwd 4BHMND, bwd 2.5x; dq 0.5x, dkdv 2x of the forward).
--variant: fat5_variant bits (include/fat5.h), e.g. 1 = FWD64_ON, 2 = FWD64_OFF, 4 / 8 = KV64 on / off, 16 / 32 = Q64 on / off."""
import argparse, os, statistics, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from attn_helpers import make_inputs
from flasht5_amd.flash_attention_v2_bias import AttentionPlan
from flasht5_amd import positional_encoding as pe

ap = argparse.ArgumentParser()
ap.add_argument("--S", default="512,2048,8192"); ap.add_argument("--modes", default="none,rpe"); ap.add_argument("--what", default="fwd,bwd")
ap.add_argument("--variant", type=int, default=0); ap.add_argument("--B", type=int, default=4); ap.add_argument("--H", type=int, default=12)
ap.add_argument("--D", type=int, default=64); ap.add_argument("--dtype", default="bf16"); ap.add_argument("--causal", action="store_true")
ap.add_argument("--radius", type=int, default=128); ap.add_argument("--no-dbias", action="store_true")
ap.add_argument("--contig", action="store_true"); ap.add_argument("--randbias", action="store_true"); ap.add_argument("--scale", type=float, default=0.125); ap.add_argument("--iters", type=int, default=20); ap.add_argument("--reps", type=int, default=5); ap.add_argument("--eager", action="store_true")
a = ap.parse_args()
dtype = torch.bfloat16 if a.dtype == "bf16" else torch.float16
STAGE = {"bwd": 7, "dq": 1, "dkdv": 2, "red": 4, "dq+dkdv": 3}
FRAC = {"fwd": 1.0, "bwd": 2.5, "dq": 0.5, "dkdv": 2.0, "dq+dkdv": 2.5, "red": 0.0}


def prewarm(fn, seconds=0.15):  # leave the idle clock
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < seconds:
        for _ in range(3):
            fn()
        torch.cuda.synchronize()


for S in (int(x) for x in a.S.split(",")):
    for mode in a.modes.split(","):
        q, k, v, _, do = make_inputs(a.B, a.H, S, S, a.D, dtype, None, seed=1, strided=not a.contig)
        table = (torch.randn(32, a.H, generator=torch.Generator().manual_seed(1)) * 0.5).cuda()
        kw = {}
        if mode == "rpe":
            kw = dict(rpe1d=pe.rpe1d_from_table(table, max_distance=a.radius), radius=a.radius)
        elif mode == "dense":
            kw = dict(bias=(torch.randn(1, a.H, S, S, generator=torch.Generator().manual_seed(2)).to(dtype).cuda() if a.randbias else pe.compute_bias(table, S, S).to(dtype).contiguous()))
        plan = AttentionPlan(q, k, v, do, causal=a.causal, sm_scale=a.scale, need_dbias=not a.no_dbias, variant=a.variant or None, **kw)
        plan.forward(); plan.backward(); torch.cuda.synchronize()
        f = 4.0 * a.B * a.H * S * S * a.D / (2 if a.causal else 1)
        cells = []
        for what in a.what.split(","):
            fn = plan.forward if what == "fwd" else (lambda st=STAGE[what]: plan.backward(st))
            batch = fn
            if not a.eager:
                g = torch.cuda.CUDAGraph()
                side = torch.cuda.Stream()
                side.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(side), torch.cuda.graph(g, stream=side):
                    for _ in range(a.iters):
                        fn()
                torch.cuda.current_stream().wait_stream(side)
                batch = g.replay
            prewarm(batch if not a.eager else fn)
            ts = []
            for _ in range(a.reps):
                s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                s.record()
                if a.eager:
                    for _ in range(a.iters):
                        fn()
                else:
                    batch()
                e.record(); torch.cuda.synchronize()
                ts.append(s.elapsed_time(e) / a.iters * 1e3)
            med = statistics.median(ts)
            tf = f * FRAC[what] / med / 1e6
            cells.append(f"{what} {min(ts):8.1f}/{med:8.1f}/{max(ts):8.1f} us" + (f" {tf:7.1f} TF/s" if tf > 0 else ""))
        print(f"S={S:5d} {mode:5s} v={a.variant}: " + " | ".join(cells), flush=True)
        del plan, q, k, v, do
