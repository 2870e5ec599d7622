import sys, time
sys.path.insert(0, "/root/repo")
import torch
from flasht5_amd import FAT5Config, FAT5ForConditionalGeneration
cfg = FAT5Config(); cfg.fuse_add_norm = True
torch.manual_seed(0)
m = FAT5ForConditionalGeneration(cfg).cuda().bfloat16()
ids = torch.randint(0, cfg.vocab_size, (4, 1024)).cuda(); labels = torch.randint(0, cfg.vocab_size, (4, 512)).cuda()
def step():
    m.zero_grad(set_to_none=True)
    m(ids, labels).backward()
for _ in range(3): step()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(10): step()
t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print(f"cfg5 step: host enqueue {(t1-t0)/10*1e3:.2f} ms, total {(t2-t0)/10*1e3:.2f} ms")
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CUDA]) as prof:
    for _ in range(3): step()
    torch.cuda.synchronize()
ev = prof.key_averages()
tot = sum(e.device_time_total for e in ev) / 3 / 1e3
print(f"sum of kernel time per step: {tot:.2f} ms")
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
    print(f"cfg5 train_step (fwd+bwd, {what}): host enqueue {(t1-t0)/10*1e3:.2f} ms, total {(t2-t0)/10*1e3:.2f} ms  ({tok/((t2-t0)/10)/1e3:.1f} k tokens/s)")
