"""Dispatch audit (developer tool): for a grid of shapes, every stage of the attention path timed with the library's own choice of
kernel body and with each body forced (fat5_attn_params.variant) -- graph replay, i.e. GPU time without host time -- and a flag
wherever the default is more than 5 % behind the best forced variant.  How the causal mis-dispatch of round 3 was found
((16,12,1024) causal: the 64-wide backward bodies 20 % behind the 32-wide ones).

    python tools/dispatch_audit.py [--bh 4x12,16x12] [--S 512,1024,2048,4096,8192] [--modes none,rpe,dense] [--D 64|128] [--stages fwd] [--scale 0.125]"""
import argparse, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from attn_helpers import make_inputs
from flasht5_amd.flash_attention_v2_bias import AttentionPlan
from flasht5_amd import positional_encoding as pe, _lib as L

ap = argparse.ArgumentParser()
ap.add_argument("--bh", default="4x12,8x12,16x12"); ap.add_argument("--S", default="512,1024,2048,4096,8192"); ap.add_argument("--modes", default="none,rpe")
ap.add_argument("--max-work", type=float, default=48 * 8192.0 ** 2 * 1.01)
ap.add_argument("--D", type=int, default=64); ap.add_argument("--scale", type=float, default=0.125)
ap.add_argument("--stages", default="fwd,dq,dkdv,bwd")  # (head_dim 128: --stages fwd -- only the forward has more than one body there)
ap.add_argument("--all", action="store_true")  # print every forced body's time, not only the best
ap.add_argument("--MN", default="")  # rectangular problems instead of --S: "512x1024,2048x512" (M x N)
a = ap.parse_args()

FWD = {"default": 0, "32row": L.V_FWD64_OFF, "64row": L.V_FWD64_ON | L.V_FWD64_KSPLIT_OFF, "64row-ksplit": L.V_FWD64_ON | L.V_FWD64_KSPLIT_ON | L.V_FWD64_MIX_OFF,
       "64row-mixed": L.V_FWD64_ON | L.V_FWD64_MIX_ON}
DQ = {"default": 0, "32row": L.V_Q64_OFF, "64row": L.V_Q64_ON}
DQ_DENSE = {"default": 0, "32row+staged/inkernel": L.V_QDB64_OFF, "64row-batch4": L.V_QDB64_ON}  # dense (1,H,M,N) bias: dQ + dbias (+ reduction) as one stage
BWD = {"default": 0, "separate": L.V_FUSED64_OFF, "fused64": L.V_FUSED64_ON}  # (round 4: dQ + dK/dV of one call, the library's choice against the one-launch 64-wide form forced / forbidden)
KV = {"default": 0, "32key": L.V_KV64_OFF, "64key": L.V_KV64_ON | L.V_KV64_HALF_OFF | L.V_KV64_MIX_OFF, "64key-half": L.V_KV64_ON | L.V_KV64_HALF_ON,
      "64key-mixed": L.V_KV64_ON | L.V_KV64_MIX_ON}


def gpu_time(fn, it):
    fn(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(it):
            fn()
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.05:  # leave the idle clock
        g.replay()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(); g.replay(); e.record(); torch.cuda.synchronize()
        best = min(best, s.elapsed_time(e) / it * 1e3)
    return best


flags = []
for bh in a.bh.split(","):
    B, H = (int(x) for x in bh.split("x"))
    shapes = [tuple(int(v) for v in x.split("x")) for x in a.MN.split(",")] if a.MN else [(int(x), int(x)) for x in a.S.split(",")]
    for M, S in shapes:  # (S: the number of keys)
        if B * H * float(M) * S > a.max_work:
            continue
        for causal in (False, True):
            for mode in a.modes.split(","):
                q, k, v, _, do = make_inputs(B, H, M, S, a.D, torch.bfloat16, None, seed=1, strided=True)
                table = (torch.randn(32, H, generator=torch.Generator().manual_seed(1)) * 0.5).cuda()
                kw = dict(rpe1d=pe.rpe1d_from_table(table), radius=128) if mode == "rpe" else {}
                if mode == "dense":
                    kw = dict(bias=pe.compute_bias(table, M, S).to(torch.bfloat16).contiguous())
                plan = AttentionPlan(q, k, v, do, causal=causal, sm_scale=a.scale, **kw)
                stages = a.stages.split(",")
                plan.forward(); torch.cuda.synchronize()
                if stages != ["fwd"]:
                    plan.backward(); torch.cuda.synchronize()
                it = max(2, min(20, int(2e10 / (B * H * float(M) * S))))
                line = f"({B:2d},{H},{M:5d}x{S:5d}) {'causal' if causal else 'full  '} {mode:4s}"
                fused = plan.bwd_launches() == 1  # (short problems: dQ and dK/dV share ONE launch -- the stage timings below are not the real path)
                dense = mode == "dense"
                kvt = {k_: v_ for k_, v_ in KV.items() if not (dense and k_ in ("64key-half", "64key-mixed"))}
                for stage, table_, fn in (("fwd", FWD if not (dense or a.D != 64) else {k_: v_ for k_, v_ in FWD.items() if k_ in ("default", "32row", "64row")}, plan.forward),
                                          ("dq", DQ_DENSE if dense else DQ, (lambda: plan.backward(5)) if dense else (lambda: plan.backward(1))), ("dkdv", kvt, lambda: plan.backward(2))):
                    if stage not in stages:
                        continue
                    if fused and stage != "fwd":
                        line += f" | {stage}: (fused launch)"
                        continue
                    res = {}
                    for name, bits in table_.items():
                        plan.set_variant(bits)
                        res[name] = gpu_time(fn, it)
                    plan.set_variant(0)
                    best = min((t, n) for n, t in res.items() if n != "default")
                    bad = res["default"] > 1.05 * best[0]
                    line += f" | {stage}: {res['default']:8.1f} us, best {best[1]} {best[0]:8.1f}" + (" <-- MISS" if bad else "")
                    if a.all:
                        line += " [" + " ".join(f"{n_}={t_:.1f}" for n_, t_ in res.items() if n_ != "default") + "]"
                    if bad:
                        flags.append((B, H, M, S, causal, mode, stage, round(res["default"], 1), best[1], round(best[0], 1)))
                res = {}
                for name, bits in () if "bwd" not in stages else (BWD.items() if not dense else {"default": 0, "round4": L.V_QDB64_OFF | L.V_KV64_OFF, "64wide": L.V_QDB64_ON | L.V_KV64_ON | L.V_FUSED64_OFF, "64wide-one-launch": L.V_QDB64_ON | L.V_KV64_ON | L.V_FUSED64_ON}.items()):
                    plan.set_variant(bits)
                    res[name] = gpu_time(lambda: plan.backward(7 if dense else 3), it)
                plan.set_variant(0)
                if not res:
                    print(line, flush=True)
                    del plan, q, k, v, do
                    continue
                best = min((t, n) for n, t in res.items() if n != "default")
                bad = res["default"] > 1.05 * best[0]
                line += f" | bwd: {res['default']:8.1f} us, best {best[1]} {best[0]:8.1f}" + (" <-- MISS" if bad else "")
                if a.all:
                    line += " [" + " ".join(f"{n_}={t_:.1f}" for n_, t_ in res.items() if n_ != "default") + "]"
                if bad:
                    flags.append((B, H, M, S, causal, mode, "bwd", round(res["default"], 1), best[1], round(best[0], 1)))
                print(line, flush=True)
                del plan, q, k, v, do
print("\nmis-dispatches (> 5 % behind the best forced body):", len(flags))
for f in flags:
    print("  ", f)
