"""Developer check (GPU box): is an RCCL all-reduce capturable inside a HIP graph on this ROCm / PyTorch stack?  One rank, backend "nccl" (= RCCL): the collective is
trivial, the capture mechanics are the real ones -- ProcessGroupNCCL under stream capture, the RCCL launch as a graph node, replay.  This is what bench.py --gpus N does
with N > 1 (U steps per replay, every step's all-reduce of the table gradient inside the graph); no multi-GPU box exists in this environment, so this is the part of that
path that can be exercised here.  usage: timeout 180 python tools/rccl_graph_check.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import torch.distributed as dist

os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29617")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
from bench import make_plan  # the bench's own problem: cfg2, T5 table mode
plan, table, idx = make_plan(512, "rpe", torch.device("cuda", 0), seed=0, units=None)


def step_local():
    plan.forward()
    plan.backward()


for _ in range(3):
    step_local()
    dist.all_reduce(plan.dbias)   # (eager warm-up: communicator creation must not happen under capture)
torch.cuda.synchronize()
want = plan.dbias.clone()
U = 4
g = torch.cuda.CUDAGraph()
side = torch.cuda.Stream()
side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    with torch.cuda.graph(g, stream=side):
        for u in range(U):
            step_local()
            dist.all_reduce(plan.dbias)
torch.cuda.current_stream().wait_stream(side)
torch.cuda.synchronize()
plan.dbias.fill_(float("nan"))
t0 = time.perf_counter()
for _ in range(50):
    g.replay()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / (50 * U) * 1e6
ok = torch.equal(plan.dbias, want)
print(f"RCCL all-reduce inside a {U}-step HIP graph (1 rank): captured and replayed, result {'identical to' if ok else 'DIFFERENT from'} the eager one; {dt:.1f} us per step incl. the collective")
dist.destroy_process_group()
sys.exit(0 if ok else 1)
