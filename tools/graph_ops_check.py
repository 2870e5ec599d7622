import os, sys
sys.path.insert(0, "/root/repo")
import torch
from flasht5_amd import (flash_attention_v2_rpe, flash_attention_v2_bias, fast_rms_layernorm, fused_add_rms_layernorm, cross_entropy_loss)
which = sys.argv[1]
dev = "cuda"
g0 = torch.Generator().manual_seed(0)
def mk(*s, rg=True): return torch.randn(*s, generator=g0).bfloat16().to(dev).requires_grad_(rg)
if which == "attn":
    q, k, v = mk(4, 12, 1024, 64), mk(4, 12, 1024, 64), mk(4, 12, 1024, 64)
    table = (torch.randn(32, 12, generator=g0) * 0.5).to(dev).requires_grad_()
    def f():
        o = flash_attention_v2_rpe(q, k, v, table, True, 32, 128, False, 0.125)
        o.float().sum().backward()
    params = [q, k, v, table]
elif which == "attn_nobias":
    q, k, v = mk(4, 12, 512, 64), mk(4, 12, 1024, 64), mk(4, 12, 1024, 64)
    def f():
        o = flash_attention_v2_bias(q, k, v, None, False, 0.125)
        o.float().sum().backward()
    params = [q, k, v]
elif which == "norm":
    x, r = mk(4096, 768), mk(4096, 768); w = torch.ones(768, device=dev).bfloat16().requires_grad_()
    def f():
        h, y = fused_add_rms_layernorm(x, r, w, 1e-6)
        z = fast_rms_layernorm(h, w, 1e-6)
        (y.float().sum() + z.float().sum()).backward()
    params = [x, r, w]
elif which == "ce":
    lg = mk(2048, 32768); lab = torch.randint(0, 32768, (2048,), generator=g0).to(dev)
    def f():
        l = lg * 1.0
        losses, z = cross_entropy_loss(l, lab, label_smoothing=0.1, lse_square_scale=1e-4, inplace_backward=True)
        losses.mean().backward()
    params = [lg]
elif which == "emb":
    emb = torch.nn.Embedding(32768, 768).cuda().bfloat16(); ids = torch.randint(0, 32768, (4, 1024), generator=g0).to(dev)
    def f():
        emb(ids).float().sum().backward()
    params = list(emb.parameters())
s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(3):
        for p in params: p.grad = None
        f()
torch.cuda.current_stream().wait_stream(s)
gr = torch.cuda.CUDAGraph()
for p in params: p.grad = None
with torch.cuda.graph(gr):
    f()
torch.cuda.synchronize()
for i in range(20):
    gr.replay()
torch.cuda.synchronize()
print(which, "OK", [float(p.grad.float().abs().sum()) for p in params][:2])
