"""developer: measured elementwise excess (tests/test_attention_gpu.py::elem_excess with ELEM_C = 1) over the golden and
reference-shape cases -- the data ELEM_C is chosen from."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import test_attention_gpu as T
from attn_helpers import make_inputs, oracle_all, run_dense
from golden_io import load_attn, ATTN_CASES
T.ELEM_C = 1.0
worst = 0.0
for name in ATTN_CASES:
    c = T.to_dev(load_attn(name))
    got = run_dense(c["q"], c["k"], c["v"], c["bias"], c["do"], c["sm_scale"], c["causal"])
    ex = {k: round(T.elem_excess(got[k], c[k], c["dtype"], 1.0 if k == "o" else 3.0), 2) for k in ("o", "dq", "dk", "dv")}
    worst = max(worst, max(ex.values())); print(name, ex, flush=True)
for (B, H, M, N, D) in [(2, 4, 512, 612, 128), (2, 4, 1024, 1045, 64)]:
    for causal in (True, False):
        for dtype in (torch.float16, torch.bfloat16):
            q, k, v, b, do = make_inputs(B, H, M, N, D, dtype, "bh", seed=3)
            ref = oracle_all(q, k, v, b, do, 1.0, causal)
            got = run_dense(q, k, v, b, do, 1.0, causal)
            ex = {k_: round(T.elem_excess(got[k_], ref[k_], dtype, 1.0 if k_ == "o" else 3.0), 2) for k_ in ("o", "dq", "dk", "dv", "db")}
            worst = max(worst, max(ex.values())); print((B, H, M, N, D, causal, str(dtype)), ex, flush=True)
print("worst excess at ELEM_C = 1:", worst)
