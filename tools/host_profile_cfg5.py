import sys, time, cProfile, pstats
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from flasht5_amd import FAT5Config, FAT5ForConditionalGeneration
cfg = FAT5Config()
setattr(cfg, sys.argv[1] if len(sys.argv) > 1 else 'fuse_norm_linear', True)
torch.manual_seed(0)
m = FAT5ForConditionalGeneration(cfg).cuda().bfloat16()
ids = torch.randint(0, cfg.vocab_size, (4, 1024)).cuda(); labels = torch.randint(0, cfg.vocab_size, (4, 512)).cuda()
def fwd(): return m(ids, labels)
for _ in range(3): fwd().backward()
torch.cuda.synchronize()
t0=time.perf_counter()
for _ in range(10):
    l = fwd()
t1=time.perf_counter(); torch.cuda.synchronize()
print(f"forward host enqueue {(t1-t0)/10*1e3:.2f} ms")
l = fwd(); torch.cuda.synchronize(); t0=time.perf_counter(); l.backward(); t1=time.perf_counter(); torch.cuda.synchronize(); t2=time.perf_counter()
print(f"backward host {(t1-t0)*1e3:.2f} ms, total {(t2-t0)*1e3:.2f} ms")
pr = cProfile.Profile(); pr.enable()
for _ in range(5): fwd()
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("tottime").print_stats(22)
# the backward runs on autograd's device thread: profile it there
import threading
pr2 = cProfile.Profile()
threading.setprofile(lambda *a: None)
def bwd_profile():
    l = fwd()
    torch.cuda.synchronize()
    pr2.enable(); l.backward(); pr2.disable()
for _ in range(3): bwd_profile()
torch.cuda.synchronize()
print("---- backward (calling thread only: the Python of custom Functions runs on the engine's device thread) ----")
pstats.Stats(pr2).sort_stats("tottime").print_stats(8)
