"""CPU restatement of the reference T5 RMSNorm (TEST INFRASTRUCTURE ONLY).

Kernel math: reference src/model/ops/rms_norm.py:25-131.
Eager module math: reference src/model/modeling_flash_t5.py:100-112.
"""
import torch


def rmsnorm_fwd_oracle(x, w, eps):
    """rms_norm.py:45-60: rstd = 1/sqrt(mean(x^2)+eps) in fp32; y = x*rstd*w -> dtype of x."""
    xf = x.float()
    var = (xf * xf).mean(-1, keepdim=True)
    rstd = 1.0 / torch.sqrt(var + eps)
    y = xf * rstd * w.float()
    return y.to(x.dtype), rstd.squeeze(-1)


def rmsnorm_bwd_oracle(dy, x, w, rstd):
    """rms_norm.py:113-124: xhat = x*rstd; wdy = w*dy; dw = sum_rows(dy*xhat);
    c1 = mean(xhat*wdy); dx = (wdy - xhat*c1)*rstd.  dw cast to w dtype (:234)."""
    xf, dyf, wf = x.float(), dy.float(), w.float()
    xhat = xf * rstd.unsqueeze(-1)
    wdy = wf * dyf
    dw = (dyf * xhat).reshape(-1, x.shape[-1]).sum(0)
    c1 = (xhat * wdy).mean(-1, keepdim=True)
    dx = (wdy - xhat * c1) * rstd.unsqueeze(-1)
    return dx.to(x.dtype), dw.to(w.dtype)


def rmsnorm_eager(x, w, eps):
    """modeling_flash_t5.py:105-112 (the non-Triton module branch)."""
    variance = x.to(torch.float32).pow(2).mean(-1, keepdim=True)
    h = x * torch.rsqrt(variance + eps)
    if w.dtype in (torch.float16, torch.bfloat16):
        h = h.to(w.dtype)
    return w * h
