"""developer probe: GraphedTrainStep phase by phase with synchronisation points (where does a captured step fail?):
`fwd | bwd | full` capture inline, `class` goes through GraphedTrainStep; --full = FAT5-base at B = 4, --fuse, --noclip, --split, --v2 (a
device-wide synchronize between replays), --nosync, --loop30."""
import os, sys, faulthandler
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
faulthandler.enable()
import torch
from flasht5_amd import FAT5Config, FAT5ForConditionalGeneration, AdamWScale, train_step, GraphedTrainStep
FULL = "--full" in sys.argv
cfg = FAT5Config() if FULL else FAT5Config(num_layers=2, num_decoder_layers=2, vocab_size=4096)
cfg.fuse_norm_linear = "--fuse" in sys.argv
g = torch.Generator().manual_seed(5)
mk = lambda: (torch.randint(0, cfg.vocab_size, (4, 1024) if FULL else (2, 512), generator=g).cuda(), torch.randint(0, cfg.vocab_size, (4, 512) if FULL else (2, 128), generator=g).cuda())
torch.manual_seed(7)
model = FAT5ForConditionalGeneration(cfg).cuda().bfloat16()
opt = AdamWScale(model.parameters(), lr=1e-3, kahan_sum=True, max_grad_norm=1.0 if "--noclip" not in sys.argv else None)
what = [a for a in sys.argv[1:] if not a.startswith("--")]
what = what[0] if what else "full"
ids, labels = mk()
if "--prof" in sys.argv:
    from torch.profiler import profile, ProfilerActivity
    with profile(activities=[ProfilerActivity.CUDA]) as prof:
        train_step(model, ids, labels, opt, max_grad_norm=None)
        torch.cuda.synchronize()
    print("profiled", len(prof.key_averages()), flush=True)
if what == "class":
    gs = GraphedTrainStep(model, opt, warmup=2, split=("--split" in sys.argv))
    import time
    nosync = "--nosync" in sys.argv
    if "--noadv" in sys.argv:  # advance the optimizer scalars only for the first replay
        real = opt.graph_advance
        state = {"n": 0}
        def adv():
            state["n"] += 1
            if state["n"] == 1:
                real()
        opt.graph_advance = adv
    if "--loop30" in sys.argv:
        for i in range(30):
            v = float(gs(ids, labels))
        print("loop30 done", v, flush=True)
        sys.exit(0)
    for i in range(5):
        print("class step", i, float(gs(ids, labels)), flush=True)
    if "--v2" in sys.argv:
        torch.cuda.synchronize()
        print("device sync ok", flush=True)
        for i in range(5):
            print("class step", i, float(gs(ids, labels)), flush=True)
        sys.exit(0)
    if "--v3" in sys.argv:
        for i in range(5):
            l = gs(ids, labels)
            print("got tensor", flush=True)
            print("item", l.item(), flush=True)
        sys.exit(0)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(20):
        l = gs(ids, labels)
        if not nosync:
            l.item()
        if "--sleep" in sys.argv:
            time.sleep(0.05)
        if "--print" in sys.argv:
            print("step", i, flush=True)
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f"20 steps: host {(t1 - t0) / 20 * 1e3:.2f} ms/step, total {(t2 - t0) / 20 * 1e3:.2f} ms/step, loss {float(l):.4f}", flush=True)
    sys.exit(0)
for i in range(2):
    print("eager", i, float(train_step(model, ids, labels, opt, max_grad_norm=None)), flush=True)
torch.cuda.synchronize()
opt.init_state()
opt.zero_grad(set_to_none=True)
gr = torch.cuda.CUDAGraph()
print("capturing", what, flush=True)
with torch.cuda.graph(gr):
    loss = model(ids, labels)
    if what in ("bwd", "full"):
        loss.backward()
    if what == "full":
        opt.step()
print("captured", flush=True)
torch.cuda.synchronize()
if what == "full":
    opt.graph_advance()
    torch.cuda.synchronize()
    print("advanced", flush=True)
    import ctypes
    from flasht5_amd.adamw_scaled import _Desc
    for group, states, scalars, raw, host in opt._graph_jobs:
        b = raw.cpu().numpy().tobytes()
        n = len(b) // ctypes.sizeof(_Desc)
        tab = (_Desc * n).from_buffer_copy(b)
        params = [p for p in group["params"] if p.grad is not None]
        print("bucket: entries", n, "scalars", scalars.tolist(), "params with grad", len(params), flush=True)
        bad = 0
        ptrs = {p.data_ptr(): p for p in params}
        for i in range(n - 1):
            d = tab[i]
            p = ptrs.get(d.p)
            if p is None or p.grad.data_ptr() != d.g or p.numel() != d.numel:
                bad += 1
                if bad < 4:
                    print("  entry", i, hex(d.p or 0), hex(d.g or 0), d.numel, "param found" if p is not None else "NO PARAM", (hex(p.grad.data_ptr()), p.numel()) if p is not None else "")
        print("  mismatching entries:", bad, " last chunk_begin", tab[n - 1].chunk_begin, flush=True)
for i in range(3):
    gr.replay()
    torch.cuda.synchronize()
    print("replay", i, float(loss), flush=True)
