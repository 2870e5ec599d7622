"""config-5 step (FAT5-base, B = 4, encoder 1024 / decoder 512 tokens) on one GPU: host enqueue time, wall time, sum of kernel
time and number of kernel launches per forward + backward, for the three formulations of the blocks --
plain (separate norm / add kernels), fuse_add_norm (residual add inside the next pre-norm), fuse_norm_linear (pre-norm inside the
projection GEMM, residual add as the output projection's epilogue: no stand-alone norm or add launch in the blocks) --
then the complete optimizer step (developer tool; bench.py's `n3_fusions.cfg5_step` key reports the same three)."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from flasht5_amd import FAT5Config, FAT5ForConditionalGeneration


def measure(flag, iters=10):
    cfg = FAT5Config()
    if flag:
        setattr(cfg, flag, True)
    torch.manual_seed(0)
    m = FAT5ForConditionalGeneration(cfg).cuda().bfloat16()
    ids = torch.randint(0, cfg.vocab_size, (4, 1024)).cuda(); labels = torch.randint(0, cfg.vocab_size, (4, 512)).cuda()

    def step():
        m.zero_grad(set_to_none=True)
        m(ids, labels).backward()
    for _ in range(3): step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(iters): step()
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    from torch.profiler import profile, ProfilerActivity
    with profile(activities=[ProfilerActivity.CUDA]) as prof:
        for _ in range(3): step()
        torch.cuda.synchronize()
    ev = prof.key_averages()
    if "--top" in sys.argv:  # where the kernel time of one step goes: the 30 largest kernels by total time
        tot = sum(e.device_time_total for e in ev)
        for e in sorted(ev, key=lambda e: -e.device_time_total)[:30]:
            print(f"  {e.device_time_total / 3 / 1e3:8.3f} ms {100 * e.device_time_total / tot:5.1f}%  x{e.count // 3:4d}  {e.key[:110]}")
    ktime = sum(e.device_time_total for e in ev) / 3 / 1e3
    launches = sum(e.count for e in ev) / 3
    norm_launches = sum(e.count for e in ev if "rmsnorm" in e.key) / 3
    torch.cuda.reset_peak_memory_stats(); step(); torch.cuda.synchronize()
    return {"host_ms": round((t1 - t0) / iters * 1e3, 2), "wall_ms": round((t2 - t0) / iters * 1e3, 2), "kernel_ms": round(ktime, 2),
            "launches": int(launches), "rmsnorm_launches": int(norm_launches), "peak_mem_GB": round(torch.cuda.max_memory_allocated() / 2 ** 30, 2)}, m, ids, labels


if __name__ == "__main__":
    m = None
    for flag in ((None, "fuse_add_norm", "fuse_norm_linear") if "--step-only" not in sys.argv else ("fuse_norm_linear",)):
        del m
        torch.cuda.empty_cache()
        r, m, ids, labels = measure(flag)
        print(f"cfg5 fwd+bwd [{flag or 'plain'}]: {r}", flush=True)
    # the complete optimizer step: forward + backward + gradient clipping (max_grad_norm 1.0) + fused AdamWScale (bf16 + Kahan)
    from flasht5_amd import AdamWScale, train_step
    tok = 4 * (1024 + 512)
    for fused_clip in (False, True):
        opt = AdamWScale(m.parameters(), lr=1e-3, weight_decay=0.0, kahan_sum=True, **({"max_grad_norm": 1.0} if fused_clip else {}))
        for _ in range(3): train_step(m, ids, labels, opt)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(10): train_step(m, ids, labels, opt)
        t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
        what = "clip inside AdamWScale" if fused_clip else "torch clip_grad_norm_ + AdamWScale"
        print(f"cfg5 train_step (fuse_norm_linear; fwd+bwd, {what}): host enqueue {(t1-t0)/10*1e3:.2f} ms, total {(t2-t0)/10*1e3:.2f} ms  ({tok/((t2-t0)/10)/1e3:.1f} k tokens/s)", flush=True)
    # the same step captured in a HIP graph (GraphedTrainStep): one replay per batch instead of ~3,400 launches
    from flasht5_amd import GraphedTrainStep
    opt = AdamWScale(m.parameters(), lr=1e-3, weight_decay=0.0, kahan_sum=True, max_grad_norm=1.0)
    gstep = GraphedTrainStep(m, opt, warmup=2)
    for _ in range(5): gstep(ids, labels)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): gstep(ids, labels)
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f"cfg5 GraphedTrainStep (fuse_norm_linear; fwd+bwd+clip+AdamWScale in one HIP graph): host enqueue {(t1-t0)/20*1e3:.2f} ms, total {(t2-t0)/20*1e3:.2f} ms  ({tok/((t2-t0)/20)/1e3:.1f} k tokens/s)", flush=True)
