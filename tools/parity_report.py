"""Measured parity errors, tensor by tensor, next to the bounds the tests enforce (run ON the GPU box):

    python tools/parity_report.py [out.json]        (default gpurun_out/parity.json; copy to profiles/parity_rNN.json)

Cases: tests/parity_cases.py (the committed fixtures of the reference's eager path and Triton kernels, BASELINE.json configs
1-4 at full size, RMSNorm / cross-entropy fixtures) + config 5 (FAT5-base step: loss and both table gradients vs the eager
fp32 twin of tests/test_cfg5_gpu.py)."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import torch  # noqa: E402

from parity_cases import ALL_CASES, rec  # noqa: E402


def cfg5_records():
    from flasht5_amd import FAT5Config, FAT5ForConditionalGeneration
    from test_cfg5_gpu import _twin_loss
    from attn_helpers import maxdiff
    cfg = FAT5Config()
    B, S, T = 4, 1024, 512
    torch.manual_seed(2026)
    model = FAT5ForConditionalGeneration(cfg).cuda().bfloat16()
    g = torch.Generator().manual_seed(5)
    ids = torch.randint(0, cfg.vocab_size, (B, S), generator=g).cuda()
    labels = torch.randint(0, cfg.vocab_size, (B, T), generator=g)
    labels[1, -37:] = -100
    labels = labels.cuda()
    loss = model(ids, labels)
    loss.backward()
    sd = {n: p.detach().float().requires_grad_() for n, p in model.named_parameters()}
    rloss = _twin_loss(sd, cfg, ids, labels, model._shift_right(labels))
    names = [n for n in sd if "relative_attention_bias" in n]
    rg = torch.autograd.grad(rloss, [sd[n] for n in names])
    got = dict(model.named_parameters())
    name = "cfg5 FAT5-base step (B=4, enc 1024, dec 512, 12+12 layers)"
    out = [rec(name, "loss", abs(loss.item() - rloss.item()), 1e-2 * abs(rloss.item()))]
    for n, r in zip(names, rg):
        out.append(rec(name, "d " + n.split(".")[0] + " rpe table (32,12)", maxdiff(got[n].grad, r), 8e-2 * r.abs().max().item() + 1e-6))
    return out


def main():
    out_path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "parity.json")
    rows = []
    for name, fn in list(ALL_CASES) + [("cfg5", cfg5_records)]:
        t0 = time.time()
        recs = fn()
        for r in recs:
            r["ratio"] = r["err"] / r["bound"] if r["bound"] > 0 else float("inf")
            rows.append(r)
        worst = max(r["ratio"] for r in recs)
        print(f"{name:34s} {len(recs):2d} tensors, worst err/bound {worst:.3f}  ({time.time() - t0:.1f} s)", flush=True)
    doc = {"device": torch.cuda.get_device_name(0), "bound": "(1e-3 + u * half_ulp(dtype)) * max(1, max|ref|), u = 1 fwd / 3 grads; see tests/parity_cases.py",
           "all_within_bound": all(r["ratio"] < 1 for r in rows), "worst_ratio": max(r["ratio"] for r in rows), "records": rows}
    os.makedirs(os.path.dirname(out_path), exist_ok=True)
    json.dump(doc, open(out_path, "w"), indent=1)
    print(f"wrote {out_path}: {len(rows)} records, all within bound: {doc['all_within_bound']}, worst ratio {doc['worst_ratio']:.3f}")


if __name__ == "__main__":
    main()
