"""Forward-only check + timing of the 64-rows-per-wave pipelined body (attn_fwd64.h) against the 32-row body and a
GPU fp32 reference (developer tool; runs on the GPU box).

    python tools/f64_check.py [--no-check] [--seqs 2048,8192]
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import torch  # noqa: E402

from flasht5_amd.flash_attention_v2_bias import AttentionPlan  # noqa: E402
from flasht5_amd import positional_encoding as pe  # noqa: E402


def ref_fwd(q, k, v, bias, scale, causal):
    """fp32 eager reference on the GPU, one (b, h) at a time"""
    B, H, M, D = q.shape
    N = k.shape[2]
    o = torch.empty(B, H, M, D, dtype=torch.float32, device=q.device)
    L = torch.empty(B, H, M, dtype=torch.float32, device=q.device)
    for b in range(B):
        for h in range(H):
            s = (q[b, h].float() @ k[b, h].float().T) * scale
            if bias is not None:
                s = s + bias[0, h].float()
            if causal:
                m = torch.arange(M, device=q.device)[:, None] + (N - M) >= torch.arange(N, device=q.device)[None, :]
                s = s.masked_fill(~m, float("-inf"))
            L[b, h] = torch.logsumexp(s, -1)
            o[b, h] = torch.softmax(s, -1) @ v[b, h].float()
    return o, L


def mk(B, H, M, N, D, dtype, seed, strided=True, amp=1.0):
    g = torch.Generator().manual_seed(seed)
    def t(S):
        x = (torch.randn(B, S, H, D, generator=g) * amp).to(dtype).cuda()
        return x.permute(0, 2, 1, 3) if strided else x.permute(0, 2, 1, 3).contiguous()
    return t(M), t(N), t(N), t(M)


def run(plan, f64):
    os.environ["FAT5_FWD64"] = "1" if f64 else "0"
    plan.forward()
    torch.cuda.synchronize()
    return plan.o.float().clone(), plan.lse.clone()


def check(B, H, M, N, D, mode, causal, dtype=torch.bfloat16, scale=0.125, amp=1.0, R=128):
    q, k, v, do = mk(B, H, M, N, D, dtype, seed=M + N)
    table = (torch.randn(32, H, generator=torch.Generator().manual_seed(3)) * 0.5).cuda()
    kw, bias = {}, None
    if mode == "rpe":
        kw = dict(rpe1d=pe.rpe1d_from_table(table, True, 32, R), radius=R)
        bias = pe.compute_bias(table, M, N, True, 32, R)
    plan = AttentionPlan(q, k, v, do, causal=causal, sm_scale=scale, need_dbias=False, **kw)
    o_ref, L_ref = ref_fwd(q, k, v, bias, scale, causal)
    res = []
    for f64 in (0, 1):
        o, L = run(plan, f64)
        fin = torch.isfinite(o).all().item()
        eo = (o - o_ref).abs().max().item()
        ok_rows = torch.isfinite(L_ref)
        eL = (L[ok_rows] - L_ref[ok_rows]).abs().max().item() if ok_rows.any() else 0.0
        res.append((eo, eL, fin))
    flag = "OK " if res[1][2] and res[1][0] < max(2.5 * res[0][0], 2e-3) + 1e-3 and res[1][1] < 2e-3 + 2 * res[0][1] else "BAD"
    print(f"  {flag} B{B} H{H} M{M} N{N} {mode:4s} c={int(causal)} {str(dtype)[6:]} amp={amp}: 32-row o {res[0][0]:.2e} L {res[0][1]:.2e} | "
          f"64-row o {res[1][0]:.2e} L {res[1][1]:.2e}{'' if res[1][2] else ' NONFINITE'}", flush=True)


def timeit(S, mode, f64, iters=20, B=4, H=12, D=64):
    q, k, v, do = mk(B, H, S, S, D, torch.bfloat16, seed=1)
    table = (torch.randn(32, H) * 0.5).cuda()
    kw = dict(rpe1d=pe.rpe1d_from_table(table), radius=128) if mode == "rpe" else {}
    plan = AttentionPlan(q, k, v, do, sm_scale=0.125, need_dbias=False, **kw)
    os.environ["FAT5_FWD64"] = "1" if f64 else "0"
    import time
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.3:
        for _ in range(5):
            plan.forward()
        torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        plan.forward()
    e.record()
    torch.cuda.synchronize()
    ms = s.elapsed_time(e) / iters
    return ms * 1e3, 4.0 * B * H * S * S * D / ms / 1e9


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--no-check", action="store_true")
    ap.add_argument("--seqs", default="1024,2048,4096,8192")
    a = ap.parse_args()
    if not a.no_check:
        print("correctness (max-abs error vs GPU fp32 eager):")
        for (B, H, M, N, mode, causal) in [
                (1, 2, 256, 256, "none", False), (1, 2, 512, 512, "rpe", False), (2, 3, 1024, 1024, "none", False),
                (2, 3, 1024, 1024, "rpe", False), (1, 2, 2048, 2048, "rpe", False), (1, 2, 2048, 2048, "rpe", True),
                (1, 2, 2048, 2048, "none", True), (1, 2, 1000, 1100, "rpe", False), (1, 2, 300, 2500, "rpe", True),
                (1, 2, 2500, 300, "none", True), (1, 1, 8192, 8192, "rpe", False), (1, 2, 4096, 4096, "none", False)]:
            check(B, H, M, N, 64, mode, causal)
        check(1, 2, 2048, 2048, 64, "rpe", False, amp=4.0, scale=1.0)   # large scores: growth / renormalisation / second pass
        check(1, 2, 2048, 2048, 64, "none", False, amp=6.0, scale=1.0)
        check(1, 2, 1024, 1024, 64, "rpe", False, dtype=torch.float16)
    print("timing (fwd, (4,12,S,64) bf16 strided):")
    for S in [int(x) for x in a.seqs.split(",")]:
        for mode in ("none", "rpe"):
            a32, b32 = timeit(S, mode, 0)
            a64, b64 = timeit(S, mode, 1)
            print(f"  S={S:5d} {mode:4s}: 32-row {a32:8.1f} us {b32:7.1f} TF/s | 64-row {a64:8.1f} us {b64:7.1f} TF/s ({b64 / 25:.1f} % of peak)", flush=True)
