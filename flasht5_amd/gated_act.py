"""Gated activation of the T5 v1.1 feed-forward on MI355X -- `act(wi_0(x)) * wi_1(x)` of the reference's
`FlashT5DenseGatedAct.forward` (src/model/modeling_flash_t5.py:139-142; act = `GELU(approximate='tanh')` or `ReLU`, :134)
as ONE kernel forward and ONE backward (libfat5.so: fat5_gated_act_fwd / _bwd), instead of an activation kernel + a multiply
and, backward, an activation-backward kernel + two multiplies (+ a concatenation when wi_0 / wi_1 share one GEMM).

`gated_act(h0, h1, act)`: two (…, F) projections.  `gated_act_packed(h, act)`: one (…, 2F) tensor whose halves are the two
projections (the output of `rmsnorm_linear(x, g, (wi_0, wi_1))`) -- its gradient comes back as one (…, 2F) tensor, which is what
that projection's backward GEMMs take.  fp32 arithmetic inside, one rounding per output element (the reference rounds the
activation to the tensor dtype before the multiply)."""
from typing import Tuple

import torch

from . import _lib

__all__ = ["gated_act", "gated_act_packed"]

_ACTS = {"gelu_tanh": 0, "relu": 1}


def _vec(dtype):
    return 4 if dtype == torch.float32 else 8


def _rows(t):
    """(rows, F) view with unit inner stride and a 16-byte aligned, vector-multiple row stride (a copy only when needed)"""
    t2 = t.reshape(-1, t.shape[-1])
    v = _vec(t.dtype)
    if t2.stride(-1) != 1 or t2.data_ptr() % 16 or (t2.shape[0] > 1 and t2.stride(0) % v):
        t2 = t2.contiguous()
    return t2


def _check(h0, h1, act):
    if act not in _ACTS:
        raise ValueError(f"act must be one of {sorted(_ACTS)}")
    if h0.shape != h1.shape or h0.dtype != h1.dtype or h0.device != h1.device:
        raise ValueError("gated_act: h0 and h1 must agree in shape, dtype and device")
    if h0.dtype not in (torch.float32, torch.float16, torch.bfloat16):
        raise TypeError("gated_act: float32, float16 or bfloat16")
    if h0.shape[-1] % _vec(h0.dtype):
        raise ValueError(f"gated_act: the last dimension must be a multiple of {_vec(h0.dtype)}")


@torch.library.custom_op("fat5::gated_act_fwd", mutates_args=(), device_types="cuda")
def gated_act_fwd(h0: torch.Tensor, h1: torch.Tensor, act: int) -> torch.Tensor:
    a, b = _rows(h0), _rows(h1)
    out = torch.empty(a.shape, dtype=a.dtype, device=a.device)
    if a.shape[0]:
        with _lib.on_device(a.device):
            _lib.check(_lib.load().fat5_gated_act_fwd(a.data_ptr(), b.data_ptr(), out.data_ptr(), a.shape[0], a.shape[1], a.stride(0), b.stride(0),
                                                      out.stride(0), int(act), _lib.dtype_code(a.dtype), _lib.stream_ptr(a.device)),
                       "fat5_gated_act_fwd")
    return out.reshape(h0.shape)


@torch.library.register_fake("fat5::gated_act_fwd")
def _gated_act_fwd_fake(h0, h1, act):
    return torch.empty(h0.shape, dtype=h0.dtype, device=h0.device)


@torch.library.custom_op("fat5::gated_act_bwd", mutates_args=(), device_types="cuda")
def gated_act_bwd(dout: torch.Tensor, h0: torch.Tensor, h1: torch.Tensor, act: int) -> torch.Tensor:
    """-> (rows, 2F): [dh0 | dh1] side by side (the packed layout; the two-tensor entry point returns its halves)"""
    a, b = _rows(h0), _rows(h1)
    g = _rows(dout if dout.dtype == a.dtype else dout.to(a.dtype))
    F = a.shape[1]
    dh = torch.empty((a.shape[0], 2 * F), dtype=a.dtype, device=a.device)
    if a.shape[0]:
        esz = dh.element_size()
        with _lib.on_device(a.device):
            _lib.check(_lib.load().fat5_gated_act_bwd(g.data_ptr(), a.data_ptr(), b.data_ptr(), dh.data_ptr(), dh.data_ptr() + F * esz, a.shape[0], F,
                                                      g.stride(0), a.stride(0), b.stride(0), 2 * F, 2 * F, int(act), _lib.dtype_code(a.dtype),
                                                      _lib.stream_ptr(a.device)), "fat5_gated_act_bwd")
    return dh


@torch.library.register_fake("fat5::gated_act_bwd")
def _gated_act_bwd_fake(dout, h0, h1, act):
    rows = h0.numel() // h0.shape[-1]
    return torch.empty((rows, 2 * h0.shape[-1]), dtype=h0.dtype, device=h0.device)


class GatedAct(torch.autograd.Function):
    @staticmethod
    def forward(ctx, h0, h1, act):
        ctx.save_for_backward(h0, h1)
        ctx.act = act
        return gated_act_fwd(h0, h1, act)

    @staticmethod
    def backward(ctx, dout):
        h0, h1 = ctx.saved_tensors
        F = h0.shape[-1]
        dh = gated_act_bwd(dout, h0, h1, ctx.act)
        return dh[:, :F].reshape(h0.shape), dh[:, F:].reshape(h1.shape), None


class GatedActPacked(torch.autograd.Function):
    @staticmethod
    def forward(ctx, h, act):
        F = h.shape[-1] // 2
        ctx.save_for_backward(h)
        ctx.act = act
        return gated_act_fwd(h[..., :F], h[..., F:], act)

    @staticmethod
    def backward(ctx, dout):
        (h,) = ctx.saved_tensors
        F = h.shape[-1] // 2
        return gated_act_bwd(dout, h[..., :F], h[..., F:], ctx.act).reshape(h.shape), None


def gated_act(h0, h1, act="gelu_tanh"):
    """act(h0) * h1, differentiable in both (reference modeling_flash_t5.py:140-142)."""
    _check(h0, h1, act)
    nat = None if (torch.compiler.is_compiling() or not h0.is_cuda) else _lib.native()
    if nat is not None:  # (C++ autograd function: same kernels, less host time per call)
        return nat.gated_act_apply(h0, h1, _ACTS[act])
    return GatedAct.apply(h0, h1, _ACTS[act])


def gated_act_packed(h, act="gelu_tanh"):
    """act(h[..., :F]) * h[..., F:] for a (…, 2F) tensor holding both projections; the gradient is one (…, 2F) tensor."""
    if h.shape[-1] % 2:
        raise ValueError("gated_act_packed: the last dimension holds the two projections side by side")
    F = h.shape[-1] // 2
    _check(h[..., :F], h[..., F:], act)
    nat = None if (torch.compiler.is_compiling() or not h.is_cuda) else _lib.native()
    if nat is not None:
        return nat.gated_act_packed_apply(h, _ACTS[act])
    return GatedActPacked.apply(h, _ACTS[act])
