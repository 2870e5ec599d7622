"""Projections with the T5 pre-norm / the residual add attached.

    rmsnorm_linear(x, norm_weight, weight, eps)   == F.linear(fast_rms_layernorm(x, norm_weight, eps), weight)
        reference: `normed = self.layer_norm(hidden_states)` then Wq / Wk / Wv (src/model/modeling_flash_t5.py:304-318, :95-98);
        likewise layer_norm -> wi_0 / wi_1 (:159-160).  Up to three weights applied to the SAME normalised input run as ONE library GEMM
        on their stack; the backward forms every weight gradient in one GEMM and the norm's input gradient, the residual path's gradient
        and x * rstd in one pass (fat5_rmsnorm_unit_bwd).
    linear_residual(a, weight, residual)          == residual + F.linear(a, weight)
        reference: `hidden_states + self.o(...)` / `hidden_states + self.wo(...)` (:316, :162-163): the add rides in the library GEMM's
        epilogue (torch.addmm, beta = 1).

Round 6: the GEMMs are LIBRARY GEMMs (torch.matmul / addmm -> hipBLASLt).  Rounds 3-5 shipped a hand-written MFMA GEMM with the
norm statistics in its prologue and the residual in its epilogue (`fat5_linear_fused`, csrc/linear_fused.h): parity-green, but 453-669 TF/s
against hipBLASLt's 690-1020 on the FAT5-base shapes -- norm + QKV 32-34 us fused vs 25-26 us as norm kernel + library GEMM, wi_0 | wi_1
39-40 vs 30 (BENCH_r03 .. r05 `n3_fusions`) -- so the kernel is gone; what this module keeps is what did pay in the config-5 step: the stacked
projections (one GEMM for q | k | v and for wi_0 | wi_1, packed gradients) and the fused backward passes around the library GEMMs.

Both differentiable in every tensor argument."""
from typing import List, Optional, Tuple

import torch

from . import _lib

__all__ = ["rmsnorm_linear", "linear_residual", "fold_weights", "RMSNormLinear", "LinearResidual", "fused_linear_supported"]


def _python_path(t):
    """the Python autograd functions (custom ops with fakes) instead of the C++ ones: under torch.compile and for FakeTensors"""
    from torch._subclasses.fake_tensor import FakeTensor
    return torch.compiler.is_compiling() or isinstance(t, FakeTensor)


def fused_linear_supported(x, weight):
    """shapes / dtypes the stacked path takes (fat5_fold_weights / fat5_rmsnorm_unit_bwd / fat5_fold_weights_bwd): 16-bit, K a multiple of 64, N of 8"""
    return (x.is_cuda and x.dtype in (torch.float16, torch.bfloat16) and weight.dtype == x.dtype and weight.dim() == 2 and
            x.shape[-1] == weight.shape[1] and weight.shape[1] % 64 == 0 and weight.shape[0] % 8 == 0)


def _rows(t):
    t2 = t.reshape(-1, t.shape[-1])
    return t2 if (t2.stride(-1) == 1 and t2.data_ptr() % 16 == 0 and t2.stride(0) % 8 == 0) else t2.contiguous()


# The kernels below are registered as `fat5::` custom ops with fake (shape-only) implementations, like every other kernel of
# the library (the reference registers each of its kernels with a fake, flash_attention_v2_bias.py:83-89, :219-226): the Python
# autograd functions of this file trace under FakeTensor / torch.compile (tests/test_fused_linear_gpu.py::test_fused_block_traces).
def _wrows(w):
    return w if (w.stride(-1) == 1 and w.data_ptr() % 16 == 0 and w.stride(0) % 8 == 0) else w.contiguous()


@torch.library.custom_op("fat5::fold_weights", mutates_args=(), device_types="cuda")
def fold_weights_op(weights: List[torch.Tensor], norm_weight: Optional[torch.Tensor]) -> torch.Tensor:
    ws = [_wrows(w) for w in weights]
    assert 1 <= len(ws) <= 3
    K = ws[0].shape[1]
    out = torch.empty((sum(w.shape[0] for w in ws), K), dtype=ws[0].dtype, device=ws[0].device)
    g = None if norm_weight is None else norm_weight.to(ws[0].dtype).contiguous()
    ptr = [w.data_ptr() for w in ws] + [None] * (3 - len(ws))
    n = [w.shape[0] for w in ws] + [0] * (3 - len(ws))
    ld = [w.stride(0) for w in ws] + [0] * (3 - len(ws))
    with _lib.on_device(out.device):
        _lib.check(_lib.load().fat5_fold_weights(ptr[0], ptr[1], ptr[2], n[0], n[1], n[2], ld[0], ld[1], ld[2],
                                                 g.data_ptr() if g is not None else None, out.data_ptr(), K, _lib.dtype_code(out.dtype),
                                                 _lib.stream_ptr(out.device)), "fat5_fold_weights")
    return out


@torch.library.register_fake("fat5::fold_weights")
def _fold_weights_fake(weights, norm_weight):
    return torch.empty((sum(w.shape[0] for w in weights), weights[0].shape[1]), dtype=weights[0].dtype, device=weights[0].device)


def fold_weights(weights, norm_weight):
    """[w0; w1; ...] (up to three (n_i, K) weights stacked along n) times diag(norm_weight), ONE launch (fat5_fold_weights):
    the projection weight `rmsnorm_linear`'s kernel takes.  norm_weight None: the plain stack."""
    return fold_weights_op(list(weights), norm_weight)


@torch.library.custom_op("fat5::rmsnorm_unit_bwd", mutates_args=(), device_types="cuda")
def rmsnorm_unit_bwd_op(gy: torch.Tensor, x: torch.Tensor, rstd: torch.Tensor, dres: Optional[torch.Tensor]) -> Tuple[torch.Tensor, torch.Tensor]:
    """(dx, xhat): the gradient of x -> x * rstd given dL/dxhat = gy (+ the residual path's gradient `dres`, fp32 sum, one
    rounding) and xhat = x * rstd itself, in one pass (fat5_rmsnorm_unit_bwd)."""
    gy, x = _rows(gy), _rows(x)
    dres = _rows(dres) if dres is not None else None
    M, K = x.shape
    dx = torch.empty((M, K), dtype=x.dtype, device=x.device)
    xhat = torch.empty((M, K), dtype=x.dtype, device=x.device)
    if M == 0:
        return dx, xhat
    with _lib.on_device(x.device):
        _lib.check(_lib.load().fat5_rmsnorm_unit_bwd(gy.data_ptr(), x.data_ptr(), rstd.data_ptr(), dx.data_ptr(), xhat.data_ptr(), M, K,
                                                     gy.stride(0), x.stride(0), K, K, dres.data_ptr() if dres is not None else None,
                                                     dres.stride(0) if dres is not None else 0, _lib.dtype_code(x.dtype),
                                                     _lib.stream_ptr(x.device)), "fat5_rmsnorm_unit_bwd")
    return dx, xhat


@torch.library.register_fake("fat5::rmsnorm_unit_bwd")
def _rmsnorm_unit_bwd_fake(gy, x, rstd, dres):
    return torch.empty(x.shape, dtype=x.dtype, device=x.device), torch.empty(x.shape, dtype=x.dtype, device=x.device)


@torch.library.custom_op("fat5::fold_weights_bwd", mutates_args=(), device_types="cuda")
def fold_weights_bwd_op(dwg: torch.Tensor, weights: List[torch.Tensor], g: torch.Tensor) -> List[torch.Tensor]:
    """[dW_0, .., dg]: dW_i = dWg_i diag(g), dg = sum_n dWg W (column sums over the stacked rows), one launch + the in-order slab sum."""
    lib = _lib.load()
    dwg = _rows(dwg)
    K = dwg.shape[1]
    ws = [_wrows(w) for w in weights]
    gq = g.to(dwg.dtype).contiguous()
    dWs = [torch.empty((w.shape[0], K), dtype=dwg.dtype, device=dwg.device) for w in ws]
    dgq = torch.empty((K,), dtype=dwg.dtype, device=dwg.device)
    ptr = [w.data_ptr() for w in ws] + [None] * (3 - len(ws))
    n = [w.shape[0] for w in ws] + [0] * (3 - len(ws))
    ld = [w.stride(0) for w in ws] + [0] * (3 - len(ws))
    dptr = [t.data_ptr() for t in dWs] + [None] * (3 - len(ws))
    scratch = torch.empty((max(int(lib.fat5_fold_weights_bwd_scratch_bytes(sum(n), K)), 4) // 4,), dtype=torch.float32, device=dwg.device)
    with _lib.on_device(dwg.device):
        _lib.check(lib.fat5_fold_weights_bwd(dwg.data_ptr(), ptr[0], ptr[1], ptr[2], n[0], n[1], n[2], ld[0], ld[1], ld[2],
                                             gq.data_ptr(), dptr[0], dptr[1], dptr[2], dgq.data_ptr(), K, _lib.dtype_code(dwg.dtype),
                                             scratch.data_ptr(), scratch.numel() * 4, _lib.stream_ptr(dwg.device)), "fat5_fold_weights_bwd")
    return dWs + [dgq]


@torch.library.register_fake("fat5::fold_weights_bwd")
def _fold_weights_bwd_fake(dwg, weights, g):
    K = dwg.shape[1]
    return [torch.empty((w.shape[0], K), dtype=dwg.dtype, device=dwg.device) for w in weights] + [torch.empty((K,), dtype=dwg.dtype, device=dwg.device)]


class RMSNormLinear(torch.autograd.Function):
    """`weights`: one to three (n_i, K) projection weights applied to the SAME normalised input (Wq, Wk, Wv / wi_0, wi_1): the
    outputs come back concatenated along the last dim, the gradients per weight.

    forward : stack (one launch: [W_i]) + the norm kernel (fat5_rmsnorm_fwd: y, rstd) + ONE library GEMM y [W_i]^T        -- 3 launches
    backward: fold (one launch: [W_i] diag g) | gy = dout Wg (GEMM) | dx and xhat = x rstd in one pass (fat5_rmsnorm_unit_bwd) |
              dWg = dout^T xhat (GEMM) | dW_i = dWg g, dg = sum_n dWg W in one launch (fat5_fold_weights_bwd)                  -- 5 launches
    (the separate ops: norm + 3 GEMMs forward; 6 GEMMs + 2 norm-backward kernels + 2 gradient accumulations backward)"""

    @staticmethod
    def forward(ctx, x, norm_weight, eps, with_residual, *weights):
        shape = x.shape
        x2 = x.reshape(-1, shape[-1])
        from .rms_norm import rmsnorm_fwd
        y, rstd = rmsnorm_fwd(x2, norm_weight, eps)  # rounded to the activation dtype like the reference's layer_norm output
        wc = weights[0] if len(weights) == 1 else fold_weights(weights, None)
        out = y @ wc.t()
        ctx.save_for_backward(x2, norm_weight, rstd, *weights)  # (the normalised activation is NOT kept: the backward rebuilds x * rstd from x and rstd)
        ctx.shape = shape
        ctx.with_residual = bool(with_residual)
        ctx.set_materialize_grads(False)  # (an unused output's gradient arrives as None, not as a zero-filled tensor)
        out = out.reshape(*shape[:-1], wc.shape[0])
        if with_residual:
            # x handed back as a second output: the sub-layer adds IT to its result (h + f(norm(h))), so the gradient of that
            # residual path arrives here and joins dx inside fat5_rmsnorm_unit_bwd instead of in an add kernel of autograd's
            return out, x.view_as(x)
        return out

    @staticmethod
    def backward(ctx, dout, dres=None):
        x2, g, rstd, *weights = ctx.saved_tensors
        if dout is None:  # only the residual alias was used
            return (dres, None, None, None, *([None] * len(weights)))
        wg = fold_weights(weights, g)  # [W_i] diag(g): dL/dxhat = dout Wg
        if dres is not None:
            dres = (dres if dres.dtype == x2.dtype else dres.to(x2.dtype)).reshape(-1, dres.shape[-1])
        d2 = dout.reshape(-1, dout.shape[-1])
        if d2.dtype != x2.dtype:
            d2 = d2.to(x2.dtype)
        gy = d2 @ wg                                                     # dL/dxhat: (M, K)
        dx, xhat = rmsnorm_unit_bwd_op(gy, x2, rstd, dres)
        dg = None
        dWs = [None] * len(weights)
        if ctx.needs_input_grad[1] or any(ctx.needs_input_grad[4:]):
            dwg = d2.t() @ xhat                                          # gradient of the folded weight [W_i] diag(g): (N, K)
            *dWq, dgq = fold_weights_bwd_op(dwg, list(weights), g)
            dg = dgq.to(g.dtype) if ctx.needs_input_grad[1] else None
            dWs = [(t.to(w.dtype) if ctx.needs_input_grad[4 + i] else None) for i, (t, w) in enumerate(zip(dWq, weights))]
        return (dx.reshape(ctx.shape) if ctx.needs_input_grad[0] else None, dg, None, None, *dWs)


class LinearResidual(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, weight, residual):
        shape = residual.shape
        a2, r2 = a.reshape(-1, a.shape[-1]), residual.reshape(-1, shape[-1])
        out = torch.addmm(r2, a2, weight.t())  # (the residual add in the library GEMM's epilogue)
        ctx.save_for_backward(a2, weight)
        ctx.ashape = a.shape
        return out.reshape(shape)

    @staticmethod
    def backward(ctx, dout):
        a2, W = ctx.saved_tensors
        d2 = dout.reshape(-1, dout.shape[-1])
        da = (d2 @ W).reshape(ctx.ashape) if ctx.needs_input_grad[0] else None
        dW = (d2.t() @ a2).to(W.dtype) if ctx.needs_input_grad[1] else None
        return da, dW, (dout if ctx.needs_input_grad[2] else None)


def rmsnorm_linear(x, norm_weight, weight, eps=1e-6, return_residual=False):
    """F.linear(fast_rms_layernorm(x, norm_weight, eps), weight); x (..., K), weight (N, K) -> (..., N).
    `weight` may be a tuple of up to three weights applied to the same normalised input (Wq, Wk, Wv / wi_0, wi_1): their outputs
    come back concatenated along the last dim (one library GEMM on the stacked weight).  Shapes the stacked path does not take
    (K % 64, N % 8, fp32) run the norm and one GEMM per weight.
    return_residual: also return `x` itself (an alias) -- the tensor the caller adds to the sub-layer's result, `h + f(norm(h))`: the
    gradient of that residual path then joins the norm's input gradient inside the backward kernel (no add kernel)."""
    if not x.is_cuda:
        raise RuntimeError("flasht5_amd operators need tensors on the HIP device (no CPU fallback)")
    weights = tuple(weight) if isinstance(weight, (tuple, list)) else (weight,)
    if (len(weights) > 3 or not all(fused_linear_supported(x, w) for w in weights) or norm_weight.shape != (x.shape[-1],) or
            x.shape[-1] > 2048):  # (fat5_rmsnorm_unit_bwd keeps a row in registers: up to 2048 16-bit elements)
        from .rms_norm import fast_rms_layernorm
        y = fast_rms_layernorm(x, norm_weight, eps)
        out = torch.cat([torch.nn.functional.linear(y, w) for w in weights], -1) if len(weights) > 1 else torch.nn.functional.linear(y, weights[0])
        return (out, x) if return_residual else out
    nat = None if _python_path(x) else _lib.native()
    if nat is not None:  # (C++ autograd function: same host logic, a fraction of the per-call cost -- above all in the backward)
        r = nat.rmsnorm_linear_apply(x, norm_weight, float(eps), bool(return_residual), list(weights))
        return (r[0], r[1]) if return_residual else r[0]
    return RMSNormLinear.apply(x, norm_weight, float(eps), bool(return_residual), *weights)


def linear_residual(a, weight, residual):
    """residual + F.linear(a, weight) with the add as the library GEMM's epilogue (torch.addmm)."""
    if not a.is_cuda:
        raise RuntimeError("flasht5_amd operators need tensors on the HIP device (no CPU fallback)")
    if not fused_linear_supported(a, weight) or residual.dtype != a.dtype or residual.shape[-1] != weight.shape[0]:
        return residual + torch.nn.functional.linear(a, weight)
    nat = None if _python_path(a) else _lib.native()
    if nat is not None:
        return nat.linear_residual_apply(a, weight, residual)
    return LinearResidual.apply(a, weight, residual)
