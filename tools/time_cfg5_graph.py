"""config 5 step (FAT5-base fwd+bwd, B=4, 1024/512) eager vs replayed from one HIP graph (developer timing)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from flasht5_amd import FAT5Config, FAT5ForConditionalGeneration
cfg = FAT5Config(); cfg.fuse_add_norm = os.environ.get("FUSE", "1") == "1"; cfg.num_layers = int(os.environ.get("NL", "12")); cfg.num_decoder_layers = int(os.environ.get("NL", "12"))
torch.manual_seed(0)
m = FAT5ForConditionalGeneration(cfg).cuda().bfloat16()
ids = torch.randint(0, cfg.vocab_size, (4, 1024)).cuda(); labels = torch.randint(0, cfg.vocab_size, (4, 512)).cuda()
def step():
    loss = m(ids, labels)
    loss.backward()
    return loss
# eager
SKIP = os.environ.get("SKIP", "")
for _ in range(0 if "e" in SKIP else 3):
    m.zero_grad(set_to_none=True); step()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(0 if "e" in SKIP else 10):
    m.zero_grad(set_to_none=True); step()
torch.cuda.synchronize()
print(f"eager fwd+bwd: {(time.perf_counter()-t0)/10*1e3:.2f} ms", flush=True)
ref_loss = 0.0 if "r" in SKIP else step().item()
# graph: warm up on a side stream, capture forward + backward with static inputs and static .grad tensors
s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(3):
        m.zero_grad(set_to_none=True); step()
torch.cuda.current_stream().wait_stream(s)
g = torch.cuda.CUDAGraph()
m.zero_grad(set_to_none=True)
with torch.cuda.graph(g):
    static_loss = step()
torch.cuda.synchronize()
# (no device -> host read between replays: on this ROCm / PyTorch build a .item() after a replay of the FULL step graph -- encoder +
#  decoder + vocabulary-size loss, with this library's or torch's own cross-entropy alike -- makes the following replays fault;
#  tools/graph_bisect.py narrows it down.  Every sub-graph, and the full graph without the read, replays fine.)
g.replay(); torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10): g.replay()
torch.cuda.synchronize()
print(f"graph replay fwd+bwd: {(time.perf_counter()-t0)/10*1e3:.2f} ms", flush=True)
print("graph loss", static_loss.item(), "eager loss", ref_loss, flush=True)
