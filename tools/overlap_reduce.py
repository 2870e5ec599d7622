"""A/B (developer tool, VERDICT r5 #1a): the table-gradient reduction of the cfg2 backward (drpe_reduce_kernel, FAT5_BWD_REDUCE) as its own stage on a FORKED stream
inside the captured step -- it has no consumer inside the step, so it may overlap the next step's forward -- against the single-stream order.  Both graphs hold
U steps; the forked form joins the side stream before the next step's dQ | dK/dV launch (which rewrites the partial rows the reduction reads) and at the end.

    python tools/overlap_reduce.py [--S 512] [--U 16] [--reps 7]"""
import argparse, os, statistics, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench

ap = argparse.ArgumentParser()
ap.add_argument("--S", type=int, default=512); ap.add_argument("--U", type=int, default=16); ap.add_argument("--reps", type=int, default=7)
a = ap.parse_args()
dev = torch.device("cuda", 0)
plan, table, idx = bench.make_plan(a.S, "rpe", dev, seed=0)
plan.forward(); plan.backward(); torch.cuda.synchronize()
want = plan.dbias.clone()


def capture(forked):
    g = torch.cuda.CUDAGraph()
    main, aux = torch.cuda.Stream(), torch.cuda.Stream()
    main.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(main):
        with torch.cuda.graph(g, stream=main):
            pending = None
            for u in range(a.U):
                plan.forward()
                if pending is not None:
                    main.wait_event(pending)  # the reduction of step u - 1 has read its partial rows
                if not forked:
                    plan.backward(7)
                    continue
                plan.backward(3)
                ev = torch.cuda.Event(); ev.record(main)
                aux.wait_event(ev)
                with torch.cuda.stream(aux):
                    plan.backward(4)
                    pending = torch.cuda.Event(); pending.record(aux)
            if pending is not None:
                main.wait_event(pending)
    torch.cuda.current_stream().wait_stream(main)
    torch.cuda.synchronize()
    return g


def time_graph(g):
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.3:
        for _ in range(5):
            g.replay()
        torch.cuda.synchronize()
    ts = []
    for _ in range(a.reps):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(10):
            g.replay()
        e.record(); torch.cuda.synchronize()
        ts.append(s.elapsed_time(e) / (10 * a.U) * 1e3)
    return min(ts), statistics.median(ts), max(ts)


for name, forked in (("single stream", False), ("reduction on a forked stream", True), ("single stream", False), ("reduction on a forked stream", True)):
    g = capture(forked)
    plan.dbias.fill_(float("nan"))
    g.replay(); torch.cuda.synchronize()
    ok = torch.equal(plan.dbias, want)
    mn, md, mx = time_graph(g)
    print(f"S={a.S} U={a.U} {name:32s}: {mn:7.2f} / {md:7.2f} / {mx:7.2f} us per step (min / median / max), table gradient bit-identical: {ok}", flush=True)
    del g
