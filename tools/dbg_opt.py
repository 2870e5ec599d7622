"""optimistic-softmax edge cases: growth (renormalise) and overflow (exact second pass)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import oracle
from flasht5_amd import flash_attention_v2_bias, flash_attention_v2_rpe
torch.manual_seed(0)
B, H, S, D = 1, 2, 1024, 64
for name, boost in (("normal", 0.0), ("growth", 40.0), ("overflow", 400.0), ("mixed", 1e3)):
    q = torch.randn(B, H, S, D).cuda().bfloat16()
    k = torch.randn(B, H, S, D).cuda().bfloat16()
    v = torch.randn(B, H, S, D).cuda().bfloat16()
    # keys >= 256 get a component along every query's direction: scores rise by ~boost (natural units)
    q[..., 0] = 4.0
    k[..., 256:, 0] = boost / 4.0
    if name == "mixed":   # only some rows overflow
        q[..., ::2, 0] = 0.0
    o = flash_attention_v2_bias(q, k, v, None, False, 1.0)
    ref_o, ref_L = oracle.attn_fwd_oracle(q.cpu(), k.cpu(), v.cpu(), None, 1.0, False)
    err = (o.float().cpu() - ref_o).abs().max().item()
    print(f"{name:9s} boost={boost:6.1f}: max|o-ref|={err:.3e} finite={torch.isfinite(o.float()).all().item()}")
