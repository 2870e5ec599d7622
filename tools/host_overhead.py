"""host-side cost of one eager call through the Python mirror (developer tool)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from attn_helpers import make_inputs
from flasht5_amd import flash_attention_v2_bias, flash_attention_v2_rpe, fast_rms_layernorm
from flasht5_amd.flash_attention_v2_bias import _attn_fwd
q, k, v, b, do = make_inputs(4, 12, 512, 512, 64, torch.bfloat16, "1h", seed=1)
table = (torch.randn(32, 12) * 0.5).cuda().requires_grad_()
def host(fn, n=300):
    for _ in range(20): fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n): fn()
    t1 = time.perf_counter()   # host time to ENQUEUE (GPU runs behind)
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    return (t1 - t0) / n * 1e6, (t2 - t0) / n * 1e6
ql, kl, vl = (t.clone().requires_grad_() for t in (q, k, v))
print("fwd only, no grad (dense bias)   host/total us: %.1f / %.1f" % host(lambda: flash_attention_v2_bias(q, k, v, b, False, 0.125)))
print("raw _attn_fwd (no custom-op)     host/total us: %.1f / %.1f" % host(lambda: _attn_fwd(q, k, v, b, None, 0, False, 0.125)))
def fb():
    o = flash_attention_v2_rpe(ql, kl, vl, table, True, 32, 128, False, 0.125)
    o.backward(do)
print("fwd+bwd autograd (rpe)           host/total us: %.1f / %.1f" % host(fb, 100))
x = torch.randn(4096, 768, device="cuda").bfloat16(); w = torch.ones(768, device="cuda").bfloat16()
print("rmsnorm fwd                      host/total us: %.1f / %.1f" % host(lambda: fast_rms_layernorm(x, w, 1e-6)))
from flasht5_amd.flash_attention_v2_bias import _attn_bwd
from flasht5_amd import positional_encoding as pe
r1 = pe.rpe1d_from_table(table.detach())
idx = pe.bucket_index32(128, True, 32, 128, "cuda")
o, L = _attn_fwd(q, k, v, None, r1, 128, False, 0.125)
print("raw _attn_fwd rpe                host/total us: %.1f / %.1f" % host(lambda: _attn_fwd(q, k, v, None, r1, 128, False, 0.125)))
print("raw _attn_bwd rpe (table grad)   host/total us: %.1f / %.1f" % host(lambda: _attn_bwd(o, do, q, k, v, None, r1, 128, L, False, 0.125, True, idx, 32)))
print("rpe1d_from_table                 host/total us: %.1f / %.1f" % host(lambda: pe.rpe1d_from_table(table.detach())))
import ctypes
from flasht5_amd import _lib
lib = _lib.load()
print("ctypes fat5_version call         host us: %.2f" % host(lambda: lib.fat5_version())[0])
