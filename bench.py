#!/usr/bin/env python
"""Headline benchmark: FA2 + T5-bias forward+backward TFLOP/s (bf16, d_head = 64).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--seq 512] [--mode rpe|dense|none]

A step = one forward + one backward of the attention hot path over one synthetic batch
(B,H,S,D) = (4,12,S,64), bias = 32-bucket T5 relative-position bias, inputs resident in HBM, every kernel
launched through the C ABI of libfat5.so and replayed from a HIP graph.  At N > 1:
  --scaling weak   (default) every rank owns its own (4,12,S,64) batch (data parallel);
  --scaling strong the ONE (4,12,S,64) batch is split over the ranks by (batch, head) units (head-major chunks):
                   a rank runs its unit range in one forward and one backward call (fat5_attn_params.unit_begin/unit_count);
either way the step ends with ONE all-reduce (RCCL over xGMI) of the (32,12) bias-table gradient.
Prints ONE JSON line on rank 0 (see the repo task contract); extra keys: by_seq, kernels, roofline, cpu_baseline,
eager_autograd (the drop-in autograd.Function path, no plan / graph), rowwise (RMSNorm, CE bandwidth), reference_shape
(the reference's own published benchmark shape).
"""
import argparse
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

PEAK_BF16_TFLOPS = 2500.0  # dense MFMA peak, /opt/skills/guides/MI355X_MICROARCH.md
PEAK_HBM_GBS = 8000.0
B, H, D = 4, 12, 64
NUM_BUCKETS, MAX_DISTANCE = 32, 128


def fwd_flops(S, causal=False):
    return 4.0 * B * H * S * S * D / (2 if causal else 1)  # reference benchmarks/bench_fa2_bias.py:10-13


def make_plan(S, mode, device, seed, units=None):
    from flasht5_amd.flash_attention_v2_bias import AttentionPlan
    from flasht5_amd import positional_encoding as pe
    g = torch.Generator().manual_seed(seed)
    # the model's real layout: (B,S,H,D) storage viewed as (B,H,S,D)
    mk = lambda: torch.randn(B, S, H, D, generator=g).to(torch.bfloat16).to(device).permute(0, 2, 1, 3)  # noqa: E731
    q, k, v, do = mk(), mk(), mk(), mk()
    table = (torch.randn(NUM_BUCKETS, H, generator=g) * 0.5).to(device)
    kw = {}
    if mode == "rpe":
        kw = dict(rpe1d=pe.rpe1d_from_table(table, True, NUM_BUCKETS, MAX_DISTANCE), radius=MAX_DISTANCE,
                  rpe_bucket=pe.bucket_index32(MAX_DISTANCE, True, NUM_BUCKETS, MAX_DISTANCE, device), num_buckets=NUM_BUCKETS)
    elif mode == "dense":
        kw = dict(bias=pe.compute_bias(table, S, S, True, NUM_BUCKETS, MAX_DISTANCE).to(torch.bfloat16).contiguous())
    plan = AttentionPlan(q, k, v, do, causal=False, sm_scale=0.125, units=units, **kw)
    idx = pe.bucket_index(MAX_DISTANCE, True, NUM_BUCKETS, MAX_DISTANCE, device)
    return plan, table, idx


def eager_autograd(S, device, iters=200):
    """The drop-in path a FAT5 model calls: flash_attention_v2_rpe(...) (autograd.Function) + autograd.grad on fresh leaves,
    eager, no AttentionPlan, no graph -- wall clock per step (host-bound at S = 512)."""
    from flasht5_amd import flash_attention_v2_rpe
    g = torch.Generator().manual_seed(0)
    mk = lambda: torch.randn(B, S, H, D, generator=g).to(torch.bfloat16).to(device).permute(0, 2, 1, 3).requires_grad_()  # noqa: E731
    q, k, v = mk(), mk(), mk()
    do = torch.randn(B, S, H, D, generator=g).to(torch.bfloat16).to(device).permute(0, 2, 1, 3)
    table = (torch.randn(NUM_BUCKETS, H, generator=g) * 0.5).to(device).requires_grad_()

    def step():
        o = flash_attention_v2_rpe(q, k, v, table, True, NUM_BUCKETS, MAX_DISTANCE, False, 0.125)
        return torch.autograd.grad(o, (q, k, v, table), do)
    for _ in range(50):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        step()
    t_host = time.perf_counter() - t0
    torch.cuda.synchronize()
    t = time.perf_counter() - t0
    out = {"ms_per_step": round(t / iters * 1e3, 5), "host_enqueue_ms_per_step": round(t_host / iters * 1e3, 5),
           "tflops": round(3.5 * fwd_flops(S) / (t / iters) / 1e12, 2),
           "what": "flash_attention_v2_rpe + torch.autograd.grad (q, k, v, table), eager, no plan, no graph"}
    # where the host time goes: the forward call, the backward node's own work (allocations + C-ABI call), and the rest =
    # torch.autograd.grad itself (graph task, hand-over to the device thread and back: paid once per backward pass of a model,
    # not once per attention layer)
    from flasht5_amd import _lib, positional_encoding as pe
    nat = _lib.native()
    if nat is not None:
        def host_us(fn, n=500):
            for _ in range(20):
                fn()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(n):
                fn()
            dt = time.perf_counter() - t0
            torch.cuda.synchronize()
            return round(dt / n * 1e6, 1)
        idx = pe.bucket_index32(pe.rpe_radius(MAX_DISTANCE), True, NUM_BUCKETS, MAX_DISTANCE, q.device)
        R = pe.rpe_radius(MAX_DISTANCE)
        qd, kd, vd, td = q.detach(), k.detach(), v.detach(), table.detach()
        r1 = nat.rpe1d_of(td, idx, R, NUM_BUCKETS)
        o, L = nat.attn_fwd(qd, kd, vd, None, r1, R, False, 0.125)
        out["host_us"] = {"forward_call": host_us(lambda: flash_attention_v2_rpe(q, k, v, table, True, NUM_BUCKETS, MAX_DISTANCE, False, 0.125)),
                          "backward_node": host_us(lambda: nat.attn_bwd(o, do, qd, kd, vd, None, r1, R, L, False, 0.125, True, idx, NUM_BUCKETS))}
        out["host_us"]["autograd_engine"] = round(out["host_enqueue_ms_per_step"] * 1e3 - out["host_us"]["forward_call"] - out["host_us"]["backward_node"], 1)
    return out


def graph_time(fn, it=20):
    """average seconds per call with the calls captured in a HIP graph (kernel-side time: no Python / ctypes time)"""
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        with torch.cuda.graph(g, stream=side):
            for _ in range(it):
                fn()
    torch.cuda.current_stream().wait_stream(side)
    # wall-clock pre-warm like event_stats (the first milliseconds after an idle gap run at the idle clock: three replays of a 30 us kernel are not
    # enough to leave it), then the median of five timed replays
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.1:
        for _ in range(3):
            g.replay()
        torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        g.replay()
        e.record()
        torch.cuda.synchronize()
        ts.append(s.elapsed_time(e) / it * 1e-3)
    return sorted(ts)[2]


def rowwise_bench(device):
    """RMSNorm / cross-entropy + z-loss bandwidth (bf16): algorithmic bytes of SURVEY 8(d) / kernel time, vs 8 TB/s"""
    from flasht5_amd.rms_norm import rmsnorm_fwd, rmsnorm_bwd
    from flasht5_amd.cross_entropy_loss import cross_entropy_fwd, cross_entropy_bwd
    out = {}
    for rows, n in ((4096, 768), (65536, 1024)):
        x = torch.randn(rows, n, device=device).bfloat16()
        w = torch.ones(n, device=device).bfloat16()
        dy = torch.randn_like(x)
        _, rstd = rmsnorm_fwd(x, w, 1e-6)
        tf = graph_time(lambda: rmsnorm_fwd(x, w, 1e-6))
        tb = graph_time(lambda: rmsnorm_bwd(dy, x, w, rstd, 1e-6))
        bf, bb = 2 * rows * n * 2, 3 * rows * n * 2
        out[f"rmsnorm_{rows}x{n}"] = {"fwd_us": round(tf * 1e6, 2), "fwd_GBs": round(bf / tf / 1e9, 1), "fwd_frac": round(bf / tf / 1e9 / PEAK_HBM_GBS, 3),
                                      "bwd_us": round(tb * 1e6, 2), "bwd_GBs": round(bb / tb / 1e9, 1), "bwd_frac": round(bb / tb / 1e9 / PEAK_HBM_GBS, 3)}
        # residual add fused into the pre-norm: x, r read; h, y written / dy, h, dres read; dx written
        from flasht5_amd.rms_norm import add_rmsnorm_fwd, add_rmsnorm_bwd
        r_ = torch.randn_like(x)
        h_, _, rstd2 = add_rmsnorm_fwd(x, r_, w, 1e-6)
        tf2 = graph_time(lambda: add_rmsnorm_fwd(x, r_, w, 1e-6))
        tb2 = graph_time(lambda: add_rmsnorm_bwd(dy, h_, w, rstd2, r_, True))
        b4 = 4 * rows * n * 2
        out[f"add_rmsnorm_{rows}x{n}"] = {"fwd_us": round(tf2 * 1e6, 2), "fwd_GBs": round(b4 / tf2 / 1e9, 1), "fwd_frac": round(b4 / tf2 / 1e9 / PEAK_HBM_GBS, 3),
                                          "bwd_us": round(tb2 * 1e6, 2), "bwd_GBs": round(b4 / tb2 / 1e9, 1), "bwd_frac": round(b4 / tb2 / 1e9 / PEAK_HBM_GBS, 3)}
        del x, dy, r_, h_
    for rows, V in ((4096, 32768), (16384, 32768)):
        lg = torch.randn(rows, V, device=device).bfloat16()
        lab = torch.randint(0, V, (rows,), device=device)
        dl = torch.randn(rows, device=device)
        _, _, lse = cross_entropy_fwd(lg, lab, None, 0.1, 1.0, 1e-4, -100)
        tf = graph_time(lambda: cross_entropy_fwd(lg, lab, None, 0.1, 1.0, 1e-4, -100), 10)
        dst = torch.empty_like(lg)
        tb = graph_time(lambda: cross_entropy_bwd(dl, lg, lse, lab, False, 0.1, 1.0, 1e-4, -100), 10)
        bf, bb = rows * V * 2, 2 * rows * V * 2
        out[f"ce_{rows}x{V}"] = {"fwd_us": round(tf * 1e6, 2), "fwd_GBs": round(bf / tf / 1e9, 1), "fwd_frac": round(bf / tf / 1e9 / PEAK_HBM_GBS, 3),
                                 "bwd_us": round(tb * 1e6, 2), "bwd_GBs": round(bb / tb / 1e9, 1), "bwd_frac": round(bb / tb / 1e9 / PEAK_HBM_GBS, 3)}
        del lg, dst
    out["bytes_model"] = "rmsnorm fwd 2RNe, bwd 3RNe; add_rmsnorm fwd 4RNe, bwd 4RNe; ce fwd RVe, bwd 2RVe (e = 2); label smoothing 0.1, z-loss 1e-4"
    # fused AdamWScale step over a FAT5-base sized parameter set (bf16 parameters + Kahan compensation): two launches
    from flasht5_amd import AdamWScale
    shapes = [(32768, 768)] * 2 + [(768, 768)] * (4 * 36) + [(2048, 768)] * (3 * 24) + [(768,)] * 62 + [(32, 12)] * 2
    params = [torch.nn.Parameter((torch.randn(*sh, device=device) * 0.02).bfloat16()) for sh in shapes]
    for p_ in params:
        p_.grad = (torch.randn_like(p_) * 0.01)
    opt = AdamWScale(params, lr=1e-3, weight_decay=0.01, kahan_sum=True)
    # (one un-warmed sample of five steps read 1.82 ms where every other run reads 1.17) -- wall-clock pre-warm, then the
    #  median of seven separately event-timed steps, min / max beside it
    t_pre = time.perf_counter()
    while time.perf_counter() - t_pre < 0.2:
        for _ in range(3):
            opt.step()
        torch.cuda.synchronize()
    dev_t, wall_t = [], []
    for _ in range(7):
        s_, e_ = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        s_.record()
        opt.step()
        e_.record()
        torch.cuda.synchronize()
        wall_t.append(time.perf_counter() - t0)
        dev_t.append(s_.elapsed_time(e_) * 1e-3)
    wall = sorted(wall_t)[3]
    n_el = sum(p_.numel() for p_ in params)
    byt = n_el * 2 * 10  # sumsq pass reads p; update reads p, g, m, v, k and writes p, m, v, k
    dev_s = sorted(dev_t)[3]
    out["adamw_scale_fat5_base"] = {"params_M": round(n_el / 1e6, 1), "tensors": len(params), "ms_per_step_wall": round(wall * 1e3, 3),
                                    "ms_per_step_device": round(dev_s * 1e3, 3), "ms_per_step_device_min_max": [round(min(dev_t) * 1e3, 3), round(max(dev_t) * 1e3, 3)],
                                    "timing": "median of 7 event-timed steps after a 0.2 s pre-warm", "GBs": round(byt / dev_s / 1e9, 1),
                                    "frac": round(byt / dev_s / 1e9 / PEAK_HBM_GBS, 3),
                                    "bytes_model": "10 x 2 B per element (bf16 + Kahan): p twice, g, m, v, k read; p, m, v, k written"}
    del params, opt
    return out


def reference_shape_bench(device):
    """The shape the reference publishes (benchmarks/bench_fa2_bias.py:10-40: B=16, H=12, causal, dense (1,H,S,S) bias,
    sm_scale 1.3; FLOPs = 4*B*S^2*H*D/2 fwd, x2.5 bwd) -- a same-shape row for BASELINE.md's A100 table."""
    from flasht5_amd.flash_attention_v2_bias import AttentionPlan
    out = {}
    for dtype, dn in ((torch.bfloat16, "bf16"), (torch.float16, "fp16")):
        for Dh in (64, 128):
            for S in (512, 1024):
                g = torch.Generator().manual_seed(S + Dh)
                q, k, v, do = (torch.randn(16, 12, S, Dh, generator=g).to(dtype).to(device) for _ in range(4))
                bias = torch.randn(1, 12, S, S, generator=g).to(dtype).to(device)
                plan = AttentionPlan(q, k, v, do, bias=bias, causal=True, sm_scale=1.3)
                plan.forward()
                tf = graph_time(plan.forward, 10)
                tb = graph_time(plan.backward, 10)
                f = 4.0 * 16 * 12 * S * S * Dh / 2
                out[f"{dn}_d{Dh}_s{S}"] = {"fwd_tflops": round(f / tf / 1e12, 1), "bwd_tflops": round(2.5 * f / tb / 1e12, 1),
                                           "fwd_us": round(tf * 1e6, 1), "bwd_us": round(tb * 1e6, 1)}
                del plan, q, k, v, do, bias
    out["what"] = "B=16, H=12, causal, dense (1,12,S,S) bias + dbias, sm_scale 1.3 (reference benchmarks/bench_fa2_bias.py)"
    return out


def d128_forward_bench(device):
    """head_dim 128 (the reference benchmarks d_head 64 and 128: benchmarks/bench_fa2_bias.py): forward of (4,12,S,128) bf16, non-causal, with the T5 bias generated
    in-kernel and with the reference's dense (1,12,S,S) bias -- the pipelined 64-row body at one wave per SIMD (round 5)."""
    from flasht5_amd.flash_attention_v2_bias import AttentionPlan
    from flasht5_amd import positional_encoding as pe
    out = {}
    for S in (1024, 2048, 8192):
        g = torch.Generator().manual_seed(S)
        q, k, v, do = (torch.randn(4, S, 12, 128, generator=g).to(torch.bfloat16).to(device).permute(0, 2, 1, 3) for _ in range(4))
        table = (torch.randn(NUM_BUCKETS, 12, generator=g) * 0.5).to(device)
        bias = torch.randn(1, 12, S, S, generator=g).to(torch.bfloat16).to(device)
        f = 4.0 * 4 * 12 * S * S * 128
        row = {}
        for mode, kw in (("rpe", dict(rpe1d=pe.rpe1d_from_table(table, True, NUM_BUCKETS, MAX_DISTANCE), radius=MAX_DISTANCE, need_dbias=False)),
                         ("dense", dict(bias=bias, need_dbias=False))):
            plan = AttentionPlan(q, k, v, do, sm_scale=128 ** -0.5, **kw)
            plan.forward()
            st = event_stats(plan.forward, 20 if S <= 2048 else 5, reps=3)
            row[mode] = {"fwd_ms": round(st["median"], 4), "fwd_tflops": round(f / st["median"] / 1e9, 1), "fwd_frac_of_peak": round(f / st["median"] / 1e9 / PEAK_BF16_TFLOPS, 4),
                         "kernel": plan.describe()["fwd"]}
            del plan
        out[str(S)] = row
        del q, k, v, do, bias
        torch.cuda.empty_cache()
    out["what"] = "(4,12,S,128) bf16 forward, non-causal, (B,S,H,D)-strided inputs; T5 bias in-kernel / dense (1,12,S,S) bias; median of 3 event-timed batches"
    return out


def dense_by_seq_bench(device):
    """The reference's own operator at the metric's sizes: (4,12,S,64) bf16, non-causal, a dense (1,12,S,S) bias shared by the batch + its gradient
    (flash_attention_v2_bias(q, k, v, bias); modeling_flash_t5.py:280-285).  FLOPs as everywhere (benchmarks/bench_fa2_bias.py:10-13)."""
    out = {}
    for S in (512, 2048, 8192):
        plan = make_plan(S, "dense", device, seed=0)[0]
        it = 50 if S <= 2048 else 10
        sf = event_stats(plan.forward, it, reps=3)
        plan.forward()
        sb = event_stats(plan.backward, it, reps=3)
        stages = {}
        for name, st in (("dq_dbias", 1), ("dkdv", 2), ("reduce", 4)):
            stages[name + "_ms"] = round(event_time(lambda st=st: plan.backward(st), max(5, it // 2), warmup=1, prewarm_s=0.05), 4)
        f = fwd_flops(S)
        tf, tb = sf["median"], sb["median"]
        bias_b = H * S * S * 2
        alg_bwd = 2 * D * 2 * B * H * (S + S) * 2 + 8 * B * H * S + 2 * bias_b  # q, k, v, o, do read, dq, dk, dv written, L / delta, bias read + dbias written
        out[str(S)] = {"fwd_ms": round(tf, 4), "bwd_ms": round(tb, 4), "fwd_tflops": round(f / tf / 1e9, 1), "bwd_tflops": round(2.5 * f / tb / 1e9, 1),
                       "fwd_bwd_tflops": round(3.5 * f / (tf + tb) / 1e9, 1), "fwd_frac_of_peak": round(f / tf / 1e9 / PEAK_BF16_TFLOPS, 4),
                       "bwd_frac_of_peak": round(2.5 * f / tb / 1e9 / PEAK_BF16_TFLOPS, 4), "bwd_stages": stages, "kernels": plan.describe(),
                       "bwd_alg_bytes": alg_bwd, "bwd_alg_GBs": round(alg_bwd / tb / 1e6, 1),
                       "bwd_traffic": {k_: load_traffic(k_, S, "dense") for k_ in (("attn_bwd_fused", "bwd_stat2") if plan.bwd_launches() == 1 else ("attn_bwd_dq", "attn_bwd_dkdv"))}}
        del plan
        torch.cuda.empty_cache()
    out["what"] = "(4,12,S,64) bf16, non-causal, dense (1,12,S,S) bias + dbias, sm_scale 0.125, (B,S,H,D)-strided inputs; median of 3 event-timed batches"
    return out


def event_time(fn, iters, warmup=3, prewarm_s=0.2):
    """average ms per call of fn(), measured with HIP events on the current stream, at steady-state clocks: a
    wall-clock pre-warm first (after an idle gap -- plan construction, the previous measurement's teardown -- the GPU
    sits at its idle clock for the first milliseconds)"""
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < prewarm_s:
        for _ in range(5):
            fn()
        torch.cuda.synchronize()
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters


def event_stats(fn, iters, reps=5):
    """min / median / max over `reps` event-timed batches of `iters` calls (ms per call): the spread a single average hides
    (the clock under these kernels is power-limited and moves by several per cent between batches)"""
    ts = sorted(event_time(fn, iters, warmup=2, prewarm_s=0.05 if i else 0.2) for i in range(reps))
    return {"min": ts[0], "median": ts[len(ts) // 2], "max": ts[-1]}


def seq_rooflines(S, mode, plan, iters):
    """`roofline` block of one sequence length: per stage (fwd, dq, dk/dv) min / median / max launch time, algorithmic flops,
    fraction of the dense bf16 MFMA peak at the median, HBM traffic per launch from the PMC passes (profiles/pmc_traffic.json);
    `dominant` = the stage with the longest median launch"""
    f = fwd_flops(S)
    plan.forward()
    plan.backward()
    if plan.bwd_launches() == 1:
        # the step's backward is ONE launch holding both halves (the stand-alone dq / dkdv stages are not what runs)
        stages = {"attn_fwd": (plan.forward, f), "attn_bwd_fused": (lambda: plan.backward(3), 2.5 * f)}
    else:
        stages = {"attn_fwd": (plan.forward, f), "attn_bwd_dq": (lambda: plan.backward(1), 0.5 * f), "attn_bwd_dkdv": (lambda: plan.backward(2), 2.0 * f)}
    out = {}
    for name, (fn, fl) in stages.items():
        st = event_stats(fn, iters, reps=3)
        out[name] = {"us_min": round(st["min"] * 1e3, 2), "us_median": round(st["median"] * 1e3, 2), "us_max": round(st["max"] * 1e3, 2),
                     "alg_flops": fl, "achieved": round(fl / st["median"] / 1e9, 1), "frac": round(fl / st["median"] / 1e9 / PEAK_BF16_TFLOPS, 4),
                     "traffic": load_traffic(name, S, mode)}
    dom = max(out, key=lambda n: out[n]["us_median"])
    return {"bound": "mfma", "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s", "dominant": dom, "achieved": out[dom]["achieved"],
            "frac": out[dom]["frac"], "avg_launch_us": out[dom]["us_median"], "traffic": out[dom]["traffic"], "stages": out,
            "kernels": plan.describe()}


def cfg5_step_flops(cfg, B, S, T):
    """GEMM + attention flops of one FAT5 training step (forward + backward), B sequences of S encoder and T decoder tokens"""
    d, dff, inner, V = cfg.d_model, cfg.d_ff, cfg.num_heads * cfg.d_kv, cfg.vocab_size
    ne, nd = B * S, B * T
    ffn = lambda n: 2 * n * d * dff * (3 if cfg.use_glu_mlp else 2)
    enc_gemm = cfg.num_layers * (2 * ne * d * 3 * inner + 2 * ne * inner * d + ffn(ne))
    dec_gemm = cfg.num_decoder_layers * (2 * nd * d * 3 * inner + 2 * nd * inner * d          # self-attention
                                         + 2 * nd * d * inner + 2 * ne * d * 2 * inner + 2 * nd * inner * d  # cross-attention (K, V from the encoder)
                                         + ffn(nd))
    head = 2 * nd * d * V
    attn_f = lambda b, m, n, causal: 4 * b * cfg.num_heads * m * n * cfg.d_kv / (2 if causal else 1)
    attn = cfg.num_layers * attn_f(B, S, S, False) + cfg.num_decoder_layers * (attn_f(B, T, T, True) + attn_f(B, T, S, False))
    out = {"gemm_fwd": enc_gemm + dec_gemm + head, "attention_fwd": attn}
    out["total"] = 3 * out["gemm_fwd"] + 3.5 * out["attention_fwd"]
    return out


def n3_bench(device):
    """SURVEY 8(f) n3's fusions, kernel time by graph replay (bf16): the stacked projections behind the pre-norm (ONE library GEMM for q | k | v and for
    wi_0 | wi_1: flasht5_amd/fused_linear.py) against the reference's op sequence (layer_norm, then one nn.Linear per weight, modeling_flash_t5.py:304-318,
    :159-160); the residual add in the GEMM epilogue against add + linear; chunked lm_head -> loss against the full-logits form (time and peak memory).
    Round 6: every GEMM here is a library GEMM -- the hand-written `fat5_linear_fused` of rounds 3-5 (32-34 us for norm + QKV where norm kernel + library
    GEMM take 25-26) is gone."""
    from flasht5_amd import rmsnorm_linear, linear_residual, fast_rms_layernorm, lm_head_cross_entropy, cross_entropy_loss
    out = {}
    M, K = 4096, 768
    x = torch.randn(M, K, device=device).bfloat16()
    g = torch.ones(K, device=device).bfloat16()
    with torch.no_grad():
        for ns, tag in (((768, 768, 768), "qkv"), ((2048, 2048), "wi01")):
            ws = tuple((torch.randn(n, K, device=device) / K ** 0.5).bfloat16() for n in ns)
            N = sum(ns)

            def separate():
                y = fast_rms_layernorm(x, g, 1e-6)
                return [torch.nn.functional.linear(y, w) for w in ws]
            tf = graph_time(lambda: rmsnorm_linear(x, g, ws, 1e-6))
            ts = graph_time(separate)
            out[f"rmsnorm_linear_{tag}_{M}x{N}x{K}"] = {"fused_us": round(tf * 1e6, 2), "separate_us": round(ts * 1e6, 2),
                                                       "fused_tflops": round(2.0 * M * N * K / tf / 1e12, 1),
                                                       "what": f"fused = stack launch + norm kernel + ONE library GEMM (N = {N}); separate = norm kernel + {len(ns)} library GEMMs (the reference's sequence)"}
        W = (torch.randn(K, K, device=device) / K ** 0.5).bfloat16()
        r = torch.randn(M, K, device=device).bfloat16()
        tf = graph_time(lambda: linear_residual(x, W, r))
        ts = graph_time(lambda: r + torch.nn.functional.linear(x, W))
        out[f"linear_residual_{M}x{K}x{K}"] = {"fused_us": round(tf * 1e6, 2), "separate_us": round(ts * 1e6, 2), "what": "fused = torch.addmm (the add in the library GEMM's epilogue)"}
    # the config-5 step (FAT5-base, B = 4, 1024 / 512 tokens, one GPU) forward + backward in its three formulations: wall time (the eager
    # step is host-bound), kernel launches and the stand-alone RMSNorm launches among them, per step
    from flasht5_amd import FAT5Config, FAT5ForConditionalGeneration
    from torch.profiler import profile, ProfilerActivity
    steps = {}
    for flag in (None, "fuse_add_norm", "fuse_norm_linear"):
        cfg = FAT5Config()
        if flag:
            setattr(cfg, flag, True)
        torch.manual_seed(0)
        m = FAT5ForConditionalGeneration(cfg).to(device).bfloat16()
        ids = torch.randint(0, cfg.vocab_size, (4, 1024), device=device)
        labels = torch.randint(0, cfg.vocab_size, (4, 512), device=device)

        def step():
            m.zero_grad(set_to_none=True)
            m(ids, labels).backward()
        for _ in range(3):
            step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(8):
            step()
        torch.cuda.synchronize()
        wall = (time.perf_counter() - t0) / 8
        with profile(activities=[ProfilerActivity.CUDA]) as prof:
            step()
            torch.cuda.synchronize()
        ev = prof.key_averages()
        steps[flag or "separate_ops"] = {"wall_ms": round(wall * 1e3, 2), "kernel_ms": round(sum(e.device_time_total for e in ev) / 1e3, 2),
                                         "launches": int(sum(e.count for e in ev)),
                                         "rmsnorm_launches": int(sum(e.count for e in ev if "rmsnorm" in e.key and "unit_bwd" not in e.key)),
                                         "tokens_per_s": round(4 * 1536 / wall, 0)}
        del m
        torch.cuda.empty_cache()
    out["cfg5_step_fwd_bwd"] = steps
    # the complete optimizer step (forward, backward, clip at 1.0, fused AdamWScale bf16 + Kahan), eager and as ONE HIP graph replay
    from flasht5_amd import AdamWScale, train_step, GraphedTrainStep
    cfg = FAT5Config()
    cfg.fuse_norm_linear = True
    full = {}
    for graphed in (False, True):
        torch.manual_seed(0)
        m = FAT5ForConditionalGeneration(cfg).to(device).bfloat16()
        opt = AdamWScale(m.parameters(), lr=1e-3, kahan_sum=True, max_grad_norm=1.0)
        ids = torch.randint(0, cfg.vocab_size, (4, 1024), device=device)
        labels = torch.randint(0, cfg.vocab_size, (4, 512), device=device)
        fn = GraphedTrainStep(m, opt, warmup=2) if graphed else (lambda i, l: train_step(m, i, l, opt, max_grad_norm=None))
        for _ in range(5):
            fn(ids, labels)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            fn(ids, labels)
        t_host = time.perf_counter() - t0
        torch.cuda.synchronize()
        t = (time.perf_counter() - t0) / 10
        full["hip_graph_replay" if graphed else "eager"] = {"ms_per_step": round(t * 1e3, 2), "host_enqueue_ms": round(t_host / 10 * 1e3, 2),
                                                             "tokens_per_s": round(4 * 1536 / t, 0)}
        del m, opt, fn
        torch.cuda.empty_cache()
    full["what"] = "FAT5-base, B = 4, 1024 encoder + 512 decoder tokens, fuse_norm_linear; GraphedTrainStep = the same step captured once in a HIP graph"
    # model flops of the step (the convention of the reference's efficiency helper, benchmarks/benchmark_utils.py:270-271: forward flops,
    # backward = 2x for the GEMMs; attention by benchmarks/bench_fa2_bias.py:10-13, backward 2.5x; elementwise work not counted)
    fl = cfg5_step_flops(cfg, 4, 1024, 512)
    full["model_tflop_per_step"] = round(fl["total"] / 1e12, 3)
    full["flops_breakdown_tflop"] = {k: round(v / 1e12, 3) for k, v in fl.items() if k != "total"}
    for key in ("eager", "hip_graph_replay"):
        tf = fl["total"] / (full[key]["ms_per_step"] * 1e-3) / 1e12
        full[key]["tflops"] = round(tf, 1)
        full[key]["frac_of_peak"] = round(tf / PEAK_BF16_TFLOPS, 4)
    out["cfg5_optimizer_step"] = full
    # lm_head -> loss at the reference's CE benchmark size (16384 rows), V = 32768
    rows, V = 16384, 32768
    hid = torch.randn(rows, K, device=device).bfloat16().requires_grad_()
    Wl = (torch.randn(V, K, device=device) / K ** 0.5).bfloat16().requires_grad_()
    lab = torch.randint(0, V, (rows,), device=device)

    def chunked():
        l, _ = lm_head_cross_entropy(hid, Wl, lab, label_smoothing=0.1, lse_square_scale=1e-4)
        return torch.autograd.grad(l.mean(), (hid, Wl))

    def full():
        l, _ = cross_entropy_loss(hid @ Wl.t(), lab, label_smoothing=0.1, lse_square_scale=1e-4, inplace_backward=True)
        return torch.autograd.grad(l.mean(), (hid, Wl))

    def chunked_mean():  # round 4: reduction="mean" -- the gradients are formed in the forward pass, no recomputation of the logits
        l, _ = lm_head_cross_entropy(hid, Wl, lab, label_smoothing=0.1, lse_square_scale=1e-4, reduction="mean")
        return torch.autograd.grad(l, (hid, Wl))

    res = {}
    for name, fn in (("chunked", chunked_mean), ("chunked_per_row_losses", chunked), ("full_logits", full)):
        for _ in range(2):
            fn()
        torch.cuda.synchronize()
        torch.cuda.reset_peak_memory_stats()
        base = torch.cuda.memory_allocated()
        t = event_time(fn, 5, warmup=1, prewarm_s=0.05)
        res[name] = {"ms": round(t, 3), "peak_extra_MB": round((torch.cuda.max_memory_allocated() - base) / 2 ** 20, 1)}
    out[f"lm_head_ce_{rows}x{V}"] = res
    return out


def kernel_breakdown(plan, S, iters):
    """per-kernel average duration (us) with HIP events around each C-ABI stage"""
    t_fwd = event_time(plan.forward, iters) * 1e3
    plan.forward()
    t_dq = event_time(lambda: plan.backward(1), iters) * 1e3
    t_dkv = event_time(lambda: plan.backward(2), iters) * 1e3
    t_red = event_time(lambda: plan.backward(4), iters) * 1e3
    f = fwd_flops(S)
    # algorithmic flops per launch: fwd 2 GEMMs; dk/dv kernel 4 GEMMs (S, dP, dV, dK);
    # dq kernel 1 GEMM (its recomputation of S and dP is not counted)
    out = {
        "attn_fwd": {"us": t_fwd, "tflops": f / t_fwd / 1e6, "alg_flops": f},
        "attn_bwd_dq": {"us": t_dq, "tflops": 0.5 * f / t_dq / 1e6, "alg_flops": 0.5 * f},
        "attn_bwd_dkdv": {"us": t_dkv, "tflops": 2.0 * f / t_dkv / 1e6, "alg_flops": 2.0 * f},
        "bias_grad_reduce": {"us": t_red},
    }
    timed = ["attn_fwd", "attn_bwd_dq", "attn_bwd_dkdv"]  # the launches of the timed step
    if plan.bwd_launches() == 1:
        # short sequences: the step's backward is ONE launch holding both halves (attn_bwd_fused_kernel);
        # the stand-alone dq / dkdv numbers above are then informational only
        t_fused = event_time(lambda: plan.backward(3), iters) * 1e3
        out["attn_bwd_fused"] = {"us": t_fused, "tflops": 2.5 * f / t_fused / 1e6, "alg_flops": 2.5 * f}
        out["attn_bwd_dq"]["in_step"] = out["attn_bwd_dkdv"]["in_step"] = False
        timed = ["attn_fwd", "attn_bwd_fused"]
    return out, timed


def _cpu_attn_time(b, h, S, reps):
    import oracle
    q, k, v = (torch.randn(b, h, S, D, requires_grad=True) for _ in range(3))
    bias = torch.randn(1, h, S, S, requires_grad=True)
    do = torch.randn(b, h, S, D)
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        o = oracle.attn_ref(q, k, v, bias, 0.125, causal=False, upcast=True)
        torch.autograd.grad(o, (q, k, v, bias), do)
        ts.append(time.perf_counter() - t0)
    ts.sort()
    return ts[len(ts) // 2], ts[0]  # median, min


def _cpu_attn_lowp(b, h, S, dtype):
    """the reference's eager path as its model runs it: low-precision matmuls, fp32 softmax (attn_ref upcast=False)"""
    import oracle
    q, k, v = (torch.randn(b, h, S, D).to(dtype).requires_grad_() for _ in range(3))
    bias = torch.randn(1, h, S, S).to(dtype).requires_grad_()
    do = torch.randn(b, h, S, D).to(dtype)
    t0 = time.perf_counter()
    o = oracle.attn_ref(q, k, v, bias, 0.125, causal=False, upcast=False)
    t1 = time.perf_counter()
    torch.autograd.grad(o, (q, k, v, bias), do)
    return t1 - t0, time.perf_counter() - t1


def cpu_baseline(S=512, reps=5):
    """The reference's eager attention (oracle restatement of src/utils/attn_ref.py) forward+backward on the host cores:
    the full cfg2 batch in fp32 (the benched workload; `value`) and -- the other samples SURVEY 8(d) lists -- cfg1
    (2,8,128,64) fp32 forward, cfg2 in bf16, the full (4,12,2048,64) batch in fp32, and a (1,2,8192,64) slice of cfg3 scaled
    x24 (the full cfg3 eager pass needs ~25 GB and minutes).  A bounded sample: ~10-30 s of CPU work in total."""
    torch.manual_seed(0)
    # a pinned thread count (the intra-op pool's default follows the box: 0.038 -> 1.13 TFLOP/s over three driver runs of the same
    # code in rounds 1-3) and the MEDIAN of the runs after one untimed warm-up (allocator, thread pool start-up); the min beside it
    threads = min(64, os.cpu_count() or 1)
    aff = None
    try:  # keep the pool on `threads` CPUs of this process's set (worker threads started from here inherit the mask): no migration across the whole host
        aff = os.sched_getaffinity(0)
        os.sched_setaffinity(0, set(sorted(aff)[:threads]))
    except (AttributeError, OSError):
        aff = None
    torch.set_num_threads(threads)
    _cpu_attn_time(B, H, S, 3)  # untimed warm-up: allocator, thread-pool start-up, page faults of the 50 MB score tensors
    med, best = _cpu_attn_time(B, H, S, max(reps, 9))
    reps = max(reps, 9)
    # `value` = the best of the timed runs: the host is shared -- medians of 0.13 .. 1.3 TFLOP/s over this round's boxes for minima of
    # 1.1 .. 1.5 -- and the minimum is what the cores can do; the median stays beside it
    out = {"value": 3.5 * fwd_flops(S) / best / 1e12, "unit": "TFLOP/s", "cores": torch.get_num_threads(),
           "host_cpus": os.cpu_count(), "kind": "port", "value_is": "best (minimum time) of the timed runs on a shared host; value_median beside it",
           "value_median": 3.5 * fwd_flops(S) / med / 1e12, "median_over_min": round(med / best, 2),
           "sample": f"eager fp32 attention fwd+bwd (oracle.attn_ref + autograd), full (4,12,{S},64) batch with dense "
                     f"(1,12,{S},{S}) bias, {threads} threads (torch.set_num_threads) on {threads} CPUs of the process's affinity set, best of {reps} runs after three warm-ups: {best*1e3:.1f} ms (median {med*1e3:.1f} ms)"}
    try:
        import oracle
        q1, k1, v1 = (torch.randn(2, 8, 128, D) for _ in range(3))
        b1 = torch.randn(1, 8, 128, 128)
        t_c1 = min(_t for _t in (_timeit(lambda: oracle.attn_ref(q1, k1, v1, b1, 0.125, causal=False, upcast=True)) for _ in range(5)))
        out["cfg1_fwd_fp32"] = {"value": 4.0 * 2 * 8 * 128 * 128 * D / t_c1 / 1e12, "unit": "TFLOP/s", "sample": f"(2,8,128,64) forward, min of 5, {t_c1*1e3:.2f} ms"}
        tf_b, tb_b = _cpu_attn_lowp(B, H, S, torch.bfloat16)
        out["cfg2_bf16"] = {"value": 3.5 * fwd_flops(S) / (tf_b + tb_b) / 1e12, "unit": "TFLOP/s",
                            "sample": f"(4,12,{S},64) bf16 matmuls + fp32 softmax (attn_ref upcast=False), one run, fwd {tf_b*1e3:.0f} ms + bwd {tb_b*1e3:.0f} ms"}
        t2k = _cpu_attn_time(B, H, 2048, 1)[0]
        out["s2048_fp32"] = {"value": 3.5 * fwd_flops(2048) / t2k / 1e12, "unit": "TFLOP/s", "sample": f"full (4,12,2048,64) fp32 fwd+bwd, one run, {t2k:.2f} s"}
    except Exception as e:  # noqa: BLE001  (host memory)
        out["extra_samples_error"] = str(e)[:100]
    try:
        t3 = _cpu_attn_time(1, 2, 8192, 1)[0]
        out["cfg3_slice"] = {"value": 3.5 * 4.0 * 1 * 2 * 8192 * 8192 * D / t3 / 1e12, "unit": "TFLOP/s",
                             "sample": f"(1,2,8192,64) slice of cfg3, one run, {t3:.2f} s; full cfg3 = x24 = {24*t3:.0f} s at this rate"}
    except Exception as e:  # noqa: BLE001  (host memory)
        out["cfg3_slice"] = {"error": str(e)[:100]}
    if aff is not None:
        try:
            os.sched_setaffinity(0, aff)
        except OSError:
            pass
    return out


def _timeit(fn):
    t0 = time.perf_counter()
    fn()
    return time.perf_counter() - t0


def load_traffic(kernel_key, S, mode):
    path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    try:
        with open(path) as f:
            t = json.load(f)
        e = t.get(f"{kernel_key}:S{S}:{mode}")
        return None if e is None else e["bytes"]  # 2 x FETCH_SIZE + WRITE_SIZE, per launch (tools/pmc_traffic.py)
    except Exception:  # noqa: BLE001
        return None


def spawn_ranks(n):
    """re-exec this command line under torch.distributed.run with n local ranks (127.0.0.1 rendezvous on a free port)"""
    import socket
    import subprocess
    with socket.socket() as s_:
        s_.bind(("127.0.0.1", 0))
        port = s_.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC only on this host driver (RCCL needs it)
    env.setdefault("OMP_NUM_THREADS", "8")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def launch_plan(world, graph_steps, bucket, no_graph, steps, want_reduce=True):
    """How the timed region is launched -- ONE rule for every N (the 1 -> N curve must compare one method): U steps per HIP-graph
    replay (--graph-steps, default 16; a nearby divisor of --steps), and with more than one rank the all-reduce of EVERY step's bias(-table) gradient
    captured inside that graph (U collectives per replay: `allreduce` == "in-graph").  Returns what the JSON line reports; `allreduce` falls back to
    "per-step" (one replay + one all-reduce enqueued from Python per step) only when the collective cannot be captured (gloo dry runs; a failing
    capture), and is "bucketed" with --bucket-allreduce."""
    U = 1 if no_graph else max(1, graph_steps)
    if U > 1 and steps % U:
        div = [u for u in range(8, 33) if steps % u == 0]
        if div:
            U = min(div, key=lambda u: (abs(u - U), u))
    mode = None
    if world > 1 and want_reduce:
        mode = "bucketed" if (bucket and U > 1) else ("in-graph" if U > 1 else "per-step")
    return {"steps_per_replay": U, "allreduce": mode,
            "launch": ("eager C-ABI calls" if no_graph else f"hipGraph replay, {U} step(s) per replay")}


def drive_steps(steps, U, one_step, replay_u, reducer, grad, stash):
    """The loop of the timed region: exactly `steps` steps of the local path and the exchange of each step's bias(-table) gradient.
    U == 1 (the default with N > 1 ranks): after every step `grad` goes into ONE all-reduce (asynchronous, double-buffered: it overlaps
    the next step like a DDP bucket).  U > 1 (--bucket-allreduce): `replay_u` runs U steps that leave their gradients in `stash`, which
    travels in one all-reduce per replay.  Returns the number of all-reduces this rank submitted (tests/test_distributed_cpu.py checks it
    over gloo; the JSON line reports it as allreduces_per_step)."""
    n0 = reducer.n if reducer is not None else 0
    if U > 1 and replay_u is not None:
        for _ in range(steps // U):
            replay_u()
            if reducer is not None:
                reducer.submit(stash)
        for _ in range(steps % U):
            one_step()
            if reducer is not None:
                stash[0].copy_(grad)
                reducer.submit(stash)
    else:
        for _ in range(steps):
            one_step()
            if reducer is not None:
                reducer.submit(grad)
    if reducer is not None:
        reducer.drain()  # every step's all-reduce finishes inside the timed region
    return (reducer.n - n0) if reducer is not None else 0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1000)
    ap.add_argument("--warmup", type=int, default=100)
    ap.add_argument("--seq", type=int, default=512)
    ap.add_argument("--mode", default="rpe", choices=["rpe", "dense", "none"])
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"],
                    help="N > 1: weak = one (4,12,S,64) batch per rank; strong = ONE batch split over the ranks by (batch, head) units")
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--graph-steps", type=int, default=16, help="steps captured per HIP graph replay (the timed region still runs exactly --steps steps, "
                    "every one the same launches on the same batch; 1 = one replay per step: a replay boundary costs ~5 us on this stack).  "
                    "With more than one rank the default is ONE step per replay and one all-reduce per step (see --bucket-allreduce)")
    ap.add_argument("--bucket-allreduce", action="store_true", help="N > 1 only: keep --graph-steps steps per replay and send their bias(-table) gradients in ONE "
                    "all-reduce per replay (what DDP's bucketing does with small gradients).  NOT the default: a data-parallel step cannot defer its gradient "
                    "past its own optimizer step, so the default N > 1 line runs exactly one all-reduce per step; with this flag the line also reports the per-step form")
    ap.add_argument("--no-extras", action="store_true", help="skip by_seq / cpu baseline (profiling runs)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # bare `python bench.py --gpus N`: launch the N ranks ourselves, one process per GPU, exactly the way the driver's
        # torchrun line would (the reference is launched by torchrun too and divides by torch.cuda.device_count(),
        # train_flash_t5.py:95).  Rank 0 of the children prints the JSON line; this parent only forwards the exit code.
        sys.exit(spawn_ranks(args.gpus))
    rank = int(os.environ.get("RANK", 0))
    local_rank = int(os.environ.get("LOCAL_RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    if os.environ.get("FAT5_BENCH_RENDEZVOUS_ONLY") == "1":
        # control-flow check that runs without a GPU (tests/test_distributed_cpu.py): rendezvous + one all-reduce over gloo
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        t = torch.tensor([float(rank + 1)])
        if world > 1:
            dist.init_process_group("gloo", rank=rank, world_size=world)
            dist.all_reduce(t)
        # the N > 1 loop of the timed region with a stand-in for the local step (the HIP path needs a GPU): what is under test is how
        # many collectives a step costs
        from flasht5_amd.sharding import OverlappedGradReduce
        grad = torch.zeros(32, H)
        lp = launch_plan(world, args.graph_steps, args.bucket_allreduce, args.no_graph, args.steps)
        U = lp["steps_per_replay"]
        stash = torch.zeros((U, 32, H)) if lp["allreduce"] == "bucketed" else None
        cnt = {"i": 0, "ar": 0}

        def one_step():
            cnt["i"] += 1
            grad.fill_(float(cnt["i"] * (rank + 1)))

        def replay_u():  # stand-in for one replay of the U-step graph
            for u in range(U):
                one_step()
                if stash is not None:
                    stash[u].copy_(grad)
                if lp["allreduce"] == "in-graph":  # (the captured collective of step u)
                    dist.all_reduce(grad)
                    cnt["ar"] += 1
                    assert float(grad[0, 0]) == cnt["i"] * sum(r + 1 for r in range(world))  # this step's own gradient, summed over the ranks

        if lp["allreduce"] == "in-graph":
            for _ in range(args.steps // U):
                replay_u()
            for _ in range(args.steps % U):
                one_step()
                dist.all_reduce(grad)
                cnt["ar"] += 1
            n_ar = cnt["ar"]
        else:
            red = OverlappedGradReduce(stash if stash is not None else grad)
            n_ar = drive_steps(args.steps, U, one_step, replay_u if U > 1 else None, red, grad, stash)
        if rank == 0:
            print(json.dumps({"rendezvous": world, "n_gpus": world, "gpus_arg": args.gpus, "sum_of_ranks_plus_1": t.item(),
                              "scaling": args.scaling, "steps": args.steps, "allreduces": n_ar, "steps_run": cnt["i"],
                              "allreduces_per_step": round(n_ar / args.steps, 4), "launch": lp["launch"],
                              "steps_per_replay": U, "allreduce": lp["allreduce"]}), flush=True)
        if world > 1:
            dist.barrier()
            dist.destroy_process_group()
        return
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (the hot path has no CPU fallback)")
    if world > 1 and os.environ.get("FAT5_BENCH_SHARE_GPU") != "1" and torch.cuda.device_count() < world:
        raise SystemExit(f"bench.py: {world} ranks need {world} GPUs, this node shows {torch.cuda.device_count()} "
                         "(FAT5_BENCH_SHARE_GPU=1 runs the N-rank control flow on one GPU over gloo: a developer dry run)")
    # developer dry run of the N > 1 control flow on a one-GPU box: FAT5_BENCH_SHARE_GPU=1 puts every rank on cuda:0 and
    # uses gloo (RCCL refuses two ranks on one device); never set by the driver
    share = os.environ.get("FAT5_BENCH_SHARE_GPU") == "1"
    if share:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    # developer check of the N > 1 launch form on a one-GPU box with the REAL backend: FAT5_BENCH_FORCE_REDUCE=1 (N = 1 only) opens a one-rank "nccl" (= RCCL) group
    # and runs the line exactly as N > 1 does -- U steps per replay, every step's all-reduce captured inside the graph, the capture validated against the eager
    # collective.  The collective itself is trivial with one rank; ProcessGroupNCCL under stream capture and the RCCL launch as a graph node are the real ones.
    # Never set by the driver (profiles/r06_bench_force_reduce_1rank.log).
    force_reduce = world == 1 and os.environ.get("FAT5_BENCH_FORCE_REDUCE") == "1"
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if share:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)
    elif force_reduce:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29618")
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=device)

    import flasht5_amd  # noqa: F401  raises if libfat5.so is missing
    S, mode = args.seq, args.mode
    strong = args.scaling == "strong" and world > 1
    units = None
    if strong:
        # every rank describes the SAME batch (same seed) and runs only its head-major unit range of it
        from flasht5_amd.sharding import unit_range, heads_needing_reduction
        units = unit_range(B, H, world, rank)
    plan, table, idx = make_plan(S, mode, device, seed=0 if strong else rank, units=units)

    def step_local():
        plan.forward()
        plan.backward()  # rpe mode: plan.dbias is the (32, H) table gradient, written by the reduction launch

    # ---- capture the local part of a step in a HIP graph (launch-bound at S = 512) ----
    for _ in range(3):
        step_local()
    torch.cuda.synchronize()
    graph = None
    if not args.no_graph:
        graph = torch.cuda.CUDAGraph()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            step_local()
            with torch.cuda.graph(graph, stream=side):
                step_local()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
    from flasht5_amd.sharding import OverlappedGradReduce
    want_reduce = (world > 1 or force_reduce) and mode != "none"
    # ONE launch rule for every N (launch_plan): U steps per replay -- the same launches in the same order, one host call per U steps -- and with more
    # than one rank the all-reduce of every step's bias(-table) gradient INSIDE the captured graph (RCCL collectives are stream operations: U of them per
    # replay, each between its step's backward and the next step's forward, in place on the gradient).  --bucket-allreduce: the U gradients are kept (one
    # stream-ordered copy per step inside the graph) and travel in ONE all-reduce per replay instead.
    lp = launch_plan(2 if force_reduce else world, args.graph_steps, args.bucket_allreduce, graph is None, args.steps, want_reduce)
    U = lp["steps_per_replay"]
    graph_u, stash, ar_fallback = None, None, None

    def capture_u(with_allreduce):
        g_ = torch.cuda.CUDAGraph()
        with torch.cuda.stream(side):
            with torch.cuda.graph(g_, stream=side):
                for u in range(U):
                    step_local()
                    if stash is not None:
                        stash[u].copy_(plan.dbias)
                    if with_allreduce:
                        dist.all_reduce(plan.dbias)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        return g_

    def all_ranks_agree(ok):
        t = torch.tensor([1.0 if ok else 0.0], device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        return bool(t.item() > 0.5)

    if U > 1:
        if lp["allreduce"] == "bucketed":
            stash = torch.zeros((U,) + tuple(plan.dbias.shape), dtype=torch.float32, device=device)
        if lp["allreduce"] == "in-graph":
            ok, why = True, None
            if share:
                ok, why = False, "gloo dry run on one GPU: a host-side collective cannot be captured"
            if ok:
                try:
                    with torch.cuda.stream(side):
                        dist.all_reduce(plan.dbias)  # (communicator and its streams exist before the capture starts)
                    torch.cuda.synchronize()
                    graph_u = capture_u(True)
                except Exception as e:  # noqa: BLE001  (capture refused: every rank must learn it -- see all_ranks_agree)
                    ok, why, graph_u = False, f"capture failed: {str(e)[:120]}", None
                    torch.cuda.synchronize()
            if not share:
                agreed = all_ranks_agree(ok)
                if ok and not agreed:
                    ok, why, graph_u = False, "capture failed on another rank", None
            if ok:
                # the replayed collectives must give what the eager ones give: one step + one eager all-reduce against the last step of one replay
                step_local()
                want_t = plan.dbias.clone()
                dist.all_reduce(want_t)
                graph_u.replay()
                torch.cuda.synchronize()
                good = bool(torch.allclose(plan.dbias, want_t, rtol=1e-3, atol=1e-3 * float(want_t.abs().max())))
                if not all_ranks_agree(good):
                    ok, why, graph_u = False, "replayed all-reduce differs from the eager one", None
            if not ok:
                ar_fallback = why
                lp = dict(lp, allreduce="per-step", steps_per_replay=1, launch="hipGraph replay, 1 step(s) per replay")
                U = 1
        if U > 1 and graph_u is None:
            graph_u = capture_u(False)
    reducer = OverlappedGradReduce(stash if stash is not None else plan.dbias) if (want_reduce and lp["allreduce"] != "in-graph") else None

    def one_step():
        if graph is not None:
            graph.replay()
        else:
            step_local()

    # Untimed pre-warm, by wall clock: a step is ~60 us, so a fixed W would end long before the GPU has left its idle
    # clock (585 MHz -> ~1.95 GHz sustained) and before the host's graph-launch path is warm.
    # (LOCAL work only: a wall-clock loop runs a different number of iterations on every rank, so it must not
    # contain collectives)
    t_pre = time.perf_counter()
    while time.perf_counter() - t_pre < 0.5:
        for _ in range(50):
            if graph is not None:
                graph.replay()
            else:
                step_local()
        torch.cuda.synchronize()

    def run_region(steps, form):
        """exactly `steps` steps in launch form `form` ("replay-u": U steps per replay; "per-step"); returns the all-reduces this rank submitted"""
        if form == "replay-u" and lp["allreduce"] == "in-graph":
            for _ in range(steps // U):
                graph_u.replay()  # U steps, U captured all-reduces
            n = (steps // U) * U
            for _ in range(steps % U):  # (leftover steps: one replay + one eager all-reduce each)
                one_step()
                dist.all_reduce(plan.dbias)
                n += 1
            return n
        if form == "replay-u" and graph_u is not None:
            return drive_steps(steps, U, one_step, graph_u.replay, reducer, plan.dbias, stash)
        return drive_steps(steps, 1, one_step, None, reducer, plan.dbias, None)

    def timed(form, red=None):
        """exactly args.steps steps between two barriers; returns (max-over-ranks seconds, all-reduces submitted by this rank inside the region, host seconds spent enqueueing)"""
        nonlocal reducer
        if red is not None:
            reducer = red
        run_region(args.warmup, form)
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = run_region(args.steps, form)
        t_host = time.perf_counter() - t0
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        el = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([el], dtype=torch.float64, device=device)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            el = t.item()
        return el, n, t_host

    elapsed, n_ar, host_s = timed("replay-u")
    per_step_form = None
    if graph_u is not None:
        # the other launch form beside the headline: one replay (and, N > 1, one all-reduce enqueued from Python) per step
        e1, n1, h1 = timed("per-step", OverlappedGradReduce(plan.dbias) if want_reduce else None)
        per_step_form = {"ms_per_step": round(e1 / args.steps * 1e3, 5), "launch": "hipGraph replay, 1 step per replay",
                         "host_enqueue_ms_per_step": round(h1 / args.steps * 1e3, 5)}
        if want_reduce:
            per_step_form["allreduces_per_step"] = round(n1 / args.steps, 4)
    ms_per_step = elapsed / args.steps * 1e3
    flops_step = 3.5 * fwd_flops(S) * (1 if strong else world)
    value = flops_step / (ms_per_step * 1e-3) / 1e12
    ar_ms = None
    if world > 1 and plan.dbias is not None:  # the all-reduce alone (blocking), reported beside the step time
        buf = plan.dbias.float().clone()
        for _ in range(5):
            dist.all_reduce(buf)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(20):
            dist.all_reduce(buf)
        torch.cuda.synchronize()
        ar_ms = (time.perf_counter() - t1) / 20 * 1e3

    if rank == 0:
        out = {
            "metric": "FA2+T5-bias fwd+bwd TFLOP/s (bf16, d_head=64)", "value": round(value, 2), "unit": "TFLOP/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 5),
            "higher_is_better": True, "scaling": "strong" if strong else "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
            "config": {"workload": f"encoder self-attn fwd+bwd, (B,H,S,d)=({B},{H},{S},{D}) "
                                   f"{'in total, (batch, head) units split over the GPUs' if strong else 'per GPU'}, non-causal, "
                                   f"32-bucket T5 RPE bias ({mode} mode), sm_scale 0.125, (B,S,H,D)-strided inputs",
                       "global_batch": B if strong else B * world, "seq_len": S,
                       "parallelism": (f"units{world} ({units[1]} of {B * H} (batch, head) units per GPU)" if strong else f"dp{world}"),
                       "bias_mode": mode, "launch": lp["launch"]},
            "frac_of_peak": round(value / (PEAK_BF16_TFLOPS * world), 4),
            "per_gpu_tflops": round(value / world, 2),
            # host time spent enqueueing the timed region / steps (graph replays and, in the per-step form, collectives): the loop is device-bound while this stays below ms_per_step
            "host_enqueue_ms_per_step": round(host_s / args.steps * 1e3, 5),
        }
        if want_reduce:
            # collectives this rank submitted inside the timed region / steps: 1.0 = the north star's "single RCCL all-reduce of the bias
            # gradient" per backward (the default: every step's own all-reduce, captured in the graph beside its kernels); 1/U only with --bucket-allreduce
            out["allreduces_per_step"] = round(n_ar / args.steps, 4)
            out["allreduce_mode"] = {"in-graph": "one per step, captured inside the HIP graph (%d per replay)" % U,
                                     "bucketed": "bucketed: one per replay of %d steps" % U,
                                     "per-step": "one per step, enqueued from Python after each replay"}[lp["allreduce"]]
            out["allreduce_in_graph"] = lp["allreduce"] == "in-graph"
            if ar_fallback is not None:
                out["allreduce_in_graph_fallback"] = ar_fallback
        if per_step_form is not None:
            out["one_replay_per_step"] = per_step_form
        if ar_ms is not None:
            out["bias_grad_allreduce_ms"] = round(ar_ms, 4)
            if strong:
                out["heads_needing_reduction"] = heads_needing_reduction(B, H, world)
        if not args.no_extras:
            iters = max(10, min(args.steps, 50))
            kern, timed = kernel_breakdown(plan, S, iters)
            dom = max(timed, key=lambda n: kern[n]["us"])
            out["kernels"] = {n: {k_: (round(v_, 3) if isinstance(v_, float) else v_) for k_, v_ in d.items()} for n, d in kern.items()}
            out["roofline"] = {"kernel": dom, "bound": "mfma", "achieved": round(kern[dom]["tflops"], 2),
                               "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s", "frac": round(kern[dom]["tflops"] / PEAK_BF16_TFLOPS, 4),
                               "avg_launch_us": round(kern[dom]["us"], 3), "traffic": load_traffic(dom, S, mode),
                               "traffic_source": "profiles/pmc_traffic.json: HBM bytes per launch from separate rocprofv3 --pmc passes (tools/pmc_traffic.py), not measured in this run",
                               "kernel_name": "attn_bwd_fused64_kernel" if (dom == "attn_bwd_fused" and plan.describe().get("dq") == "64row") else dom}
            by_seq = {}
            rooflines = {}
            for s2 in (512, 2048, 8192):
                p2 = plan if (s2 == S and not strong) else make_plan(s2, mode, device, seed=rank)[0]
                it = 50 if s2 <= 2048 else 20
                sf = event_stats(p2.forward, it, reps=3)
                p2.forward()
                sb = event_stats(p2.backward, it, reps=3)
                tf, tb = sf["median"], sb["median"]
                f = fwd_flops(s2)
                if s2 != 512:  # (S = 512: the top-level `roofline` block)
                    rooflines[str(s2)] = seq_rooflines(s2, mode, p2, max(5, it // 2))
                by_seq[str(s2)] = {"fwd_ms": round(tf, 4), "bwd_ms": round(tb, 4), "fwd_tflops": round(f / tf / 1e9, 1),
                                   "bwd_tflops": round(2.5 * f / tb / 1e9, 1),
                                   "fwd_bwd_tflops": round(3.5 * f / (tf + tb) / 1e9, 1),
                                   "fwd_frac_of_peak": round(f / tf / 1e9 / PEAK_BF16_TFLOPS, 4),
                                   # backward: counted flops (5 GEMMs) and executed MFMA flops (the two-kernel design runs 7)
                                   "bwd_frac_of_peak": round(2.5 * f / tb / 1e9 / PEAK_BF16_TFLOPS, 4),
                                   "bwd_mfma_frac_of_peak": round(3.5 * f / tb / 1e9 / PEAK_BF16_TFLOPS, 4),
                                   "fwd_ms_min_max": [round(sf["min"], 4), round(sf["max"], 4)],
                                   "bwd_ms_min_max": [round(sb["min"], 4), round(sb["max"], 4)]}
                del p2
            out["by_seq"] = by_seq
            out["roofline_by_seq"] = rooflines
            if world == 1:
                out["eager_autograd"] = eager_autograd(S, device)
                out["rowwise"] = rowwise_bench(device)
                out["reference_shape"] = reference_shape_bench(device)
                out["d128_forward"] = d128_forward_bench(device)
                out["dense_by_seq"] = dense_by_seq_bench(device)
                out["n3_fusions"] = n3_bench(device)
                out["cpu_baseline"] = cpu_baseline(512)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()  # (rank 0 may still have been busy with its extras: everyone leaves together)
        dist.destroy_process_group()
    elif force_reduce:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
