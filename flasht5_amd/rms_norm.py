"""T5 RMSNorm on MI355X -- host-side mirror of the reference src/model/ops/rms_norm.py:
`fast_rms_layernorm(X, W, eps)` / `Fast_RMS_Layernorm` with the same semantics (fp32 statistics, output in
X's dtype, rstd saved and Y recomputed in backward), backed by bandwidth-bound HIP kernels (libfat5.so)."""
import ctypes
from typing import Tuple

import torch

from . import _lib

__all__ = ["fast_rms_layernorm", "Fast_RMS_Layernorm", "fused_add_rms_layernorm", "FusedAddRMSLayernorm"]


def _rows_ok(t):
    return t.stride(-1) == 1 and t.data_ptr() % 16 == 0


@torch.library.custom_op("fat5::rmsnorm_fwd", mutates_args=(), device_types="cuda")
def rmsnorm_fwd(X: torch.Tensor, weight: torch.Tensor, eps: float) -> Tuple[torch.Tensor, torch.Tensor]:
    """reference rmsnorm_triton_fwd (rms_norm.py:134-174)."""
    M, N = X.shape
    assert weight.shape == (N,)
    if not _rows_ok(X):
        X = X.contiguous()
    weight = weight.contiguous()
    Y = torch.empty((M, N), dtype=X.dtype, device=X.device)
    rstd = torch.empty((M,), dtype=torch.float32, device=X.device)
    if M == 0:
        return Y, rstd
    with _lib.on_device(X.device):
        _lib.check(_lib.load().fat5_rmsnorm_fwd(
            X.data_ptr(), weight.data_ptr(), Y.data_ptr(), rstd.data_ptr(), M, N, X.stride(0), Y.stride(0),
            float(eps), _lib.dtype_code(X.dtype), _lib.dtype_code(weight.dtype), _lib.stream_ptr(X.device)),
            "fat5_rmsnorm_fwd")
    return Y, rstd


@torch.library.register_fake("fat5::rmsnorm_fwd")
def _rmsnorm_fwd_fake(X, weight, eps):
    M, N = X.shape
    return torch.empty((M, N), dtype=X.dtype, device=X.device), torch.empty((M,), dtype=torch.float32, device=X.device)


@torch.library.custom_op("fat5::rmsnorm_bwd", mutates_args=(), device_types="cuda")
def rmsnorm_bwd(dy: torch.Tensor, x: torch.Tensor, weight: torch.Tensor, rstd: torch.Tensor, eps: float
                ) -> Tuple[torch.Tensor, torch.Tensor]:
    """reference rmsnorm_triton_bwd (rms_norm.py:186-236)."""
    M, N = x.shape
    assert dy.shape == (M, N)
    if dy.dtype != x.dtype:
        dy = dy.to(x.dtype)
    if not _rows_ok(x):
        x = x.contiguous()
    if not _rows_ok(dy):
        dy = dy.contiguous()
    weight = weight.contiguous()
    dx = torch.empty((M, N), dtype=x.dtype, device=x.device)
    dw = torch.empty((N,), dtype=weight.dtype, device=weight.device)
    if M == 0:
        return dx, dw.zero_()
    lib = _lib.load()
    nbytes = lib.fat5_rmsnorm_bwd_workspace_bytes(M, N)
    ws = torch.empty(max(nbytes, 16), dtype=torch.uint8, device=x.device)
    with _lib.on_device(x.device):
        _lib.check(lib.fat5_rmsnorm_bwd(
            dy.data_ptr(), x.data_ptr(), weight.data_ptr(), rstd.data_ptr(), dx.data_ptr(), dw.data_ptr(), M, N,
            dy.stride(0), x.stride(0), dx.stride(0), _lib.dtype_code(x.dtype), _lib.dtype_code(weight.dtype),
            ws.data_ptr(), ws.numel(), _lib.stream_ptr(x.device)), "fat5_rmsnorm_bwd")
    return dx, dw


@torch.library.register_fake("fat5::rmsnorm_bwd")
def _rmsnorm_bwd_fake(dy, x, weight, rstd, eps):
    return torch.empty(x.shape, dtype=x.dtype, device=x.device), torch.empty(weight.shape, dtype=weight.dtype, device=weight.device)


class Fast_RMS_Layernorm(torch.autograd.Function):
    """Same contract as the reference class (rms_norm.py:250-283)."""

    @staticmethod
    def forward(ctx, X, W, eps=1e-6):
        X_orig_shape = X.shape
        X = X.reshape(-1, X.shape[-1])
        y, rstd = torch.ops.fat5.rmsnorm_fwd(X, W, float(eps))
        ctx.save_for_backward(X, W, rstd)  # y is recomputed in backward, like the reference (:261-262)
        ctx.x_shape_og = X_orig_shape
        ctx.eps = eps
        return y.reshape(X_orig_shape)

    @staticmethod
    def backward(ctx, dY):
        X, weight, rstd = ctx.saved_tensors
        dY = dY.reshape(-1, dY.shape[-1])
        assert dY.shape == X.shape
        dx, dw = torch.ops.fat5.rmsnorm_bwd(dY, X, weight, rstd, float(ctx.eps))
        return dx.reshape(ctx.x_shape_og), dw, None


def fast_rms_layernorm(X, W, eps):
    """y = x * rsqrt(mean(x^2, -1) + eps) * W over the last dim; differentiable in X and W
    (reference rms_norm.py:285-287)."""
    nat = None if torch.compiler.is_compiling() else _lib.native()
    if nat is not None and X.is_cuda and X.dim() >= 1 and W.dim() == 1 and W.shape[0] == X.shape[-1]:
        return nat.rmsnorm_apply(X, W, float(eps))  # the same launches from C++ autograd functions (csrc/torch_binding.cpp)
    return Fast_RMS_Layernorm.apply(X, W, eps)


# ------------------------------------------------------------------------------------------------
# residual add + RMSNorm in one pass (the residual epilogue of a T5 sub-layer fused into the next pre-norm,
# reference modeling_flash_t5.py:159-164 / :304-318) -- bit-identical to `h = x + r; y = fast_rms_layernorm(h, W, eps)`
# ------------------------------------------------------------------------------------------------
@torch.library.custom_op("fat5::add_rmsnorm_fwd", mutates_args=(), device_types="cuda")
def add_rmsnorm_fwd(X: torch.Tensor, R: torch.Tensor, weight: torch.Tensor, eps: float) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
    M, N = X.shape
    assert R.shape == (M, N) and R.dtype == X.dtype and weight.shape == (N,)
    if not _rows_ok(X):
        X = X.contiguous()
    if not _rows_ok(R):
        R = R.contiguous()
    weight = weight.contiguous()
    H = torch.empty((M, N), dtype=X.dtype, device=X.device)
    Y = torch.empty((M, N), dtype=X.dtype, device=X.device)
    rstd = torch.empty((M,), dtype=torch.float32, device=X.device)
    if M == 0:
        return H, Y, rstd
    with _lib.on_device(X.device):
        _lib.check(_lib.load().fat5_add_rmsnorm_fwd(
            X.data_ptr(), R.data_ptr(), weight.data_ptr(), H.data_ptr(), Y.data_ptr(), rstd.data_ptr(), M, N, X.stride(0), R.stride(0),
            H.stride(0), Y.stride(0), float(eps), _lib.dtype_code(X.dtype), _lib.dtype_code(weight.dtype), _lib.stream_ptr(X.device)),
            "fat5_add_rmsnorm_fwd")
    return H, Y, rstd


@torch.library.register_fake("fat5::add_rmsnorm_fwd")
def _add_rmsnorm_fwd_fake(X, R, weight, eps):
    M, N = X.shape
    e = lambda: torch.empty((M, N), dtype=X.dtype, device=X.device)
    return e(), e(), torch.empty((M,), dtype=torch.float32, device=X.device)


@torch.library.custom_op("fat5::add_rmsnorm_bwd", mutates_args=(), device_types="cuda")
def add_rmsnorm_bwd(dy: torch.Tensor, h: torch.Tensor, weight: torch.Tensor, rstd: torch.Tensor, dres: torch.Tensor, has_dres: bool
                    ) -> Tuple[torch.Tensor, torch.Tensor]:
    """dx = round(rmsnorm_bwd_dx(dy, h)) + dres (the gradient of h's other consumer, the residual stream), dw as rmsnorm_bwd"""
    M, N = h.shape
    assert dy.shape == (M, N)
    if dy.dtype != h.dtype:
        dy = dy.to(h.dtype)
    if not _rows_ok(dy):
        dy = dy.contiguous()
    if has_dres:
        assert dres.shape == (M, N)
        if dres.dtype != h.dtype:
            dres = dres.to(h.dtype)
        if not _rows_ok(dres):
            dres = dres.contiguous()
    weight = weight.contiguous()
    dx = torch.empty((M, N), dtype=h.dtype, device=h.device)
    dw = torch.empty((N,), dtype=weight.dtype, device=weight.device)
    if M == 0:
        return dx, dw.zero_()
    lib = _lib.load()
    nbytes = lib.fat5_rmsnorm_bwd_workspace_bytes(M, N)
    ws = torch.empty(max(nbytes, 16), dtype=torch.uint8, device=h.device)
    with _lib.on_device(h.device):
        _lib.check(lib.fat5_add_rmsnorm_bwd(
            dy.data_ptr(), h.data_ptr(), weight.data_ptr(), rstd.data_ptr(), dres.data_ptr() if has_dres else None, dx.data_ptr(),
            dw.data_ptr(), M, N, dy.stride(0), h.stride(0), dres.stride(0) if has_dres else 0, dx.stride(0), _lib.dtype_code(h.dtype),
            _lib.dtype_code(weight.dtype), ws.data_ptr(), ws.numel(), _lib.stream_ptr(h.device)), "fat5_add_rmsnorm_bwd")
    return dx, dw


@torch.library.register_fake("fat5::add_rmsnorm_bwd")
def _add_rmsnorm_bwd_fake(dy, h, weight, rstd, dres, has_dres):
    return torch.empty(h.shape, dtype=h.dtype, device=h.device), torch.empty(weight.shape, dtype=weight.dtype, device=weight.device)


class FusedAddRMSLayernorm(torch.autograd.Function):
    """(h, y) = (x + r, rmsnorm(x + r) * W); the gradient of x and of r is d h_total = rmsnorm_bwd_dx(dy) + dh."""

    @staticmethod
    def forward(ctx, X, R, W, eps=1e-6):
        shape = X.shape
        h, y, rstd = torch.ops.fat5.add_rmsnorm_fwd(X.reshape(-1, shape[-1]), R.reshape(-1, shape[-1]), W, float(eps))
        ctx.save_for_backward(h, W, rstd)
        ctx.shape = shape
        return h.reshape(shape), y.reshape(shape)

    @staticmethod
    def backward(ctx, dH, dY):
        h, W, rstd = ctx.saved_tensors
        n = h.shape[-1]
        if dY is None:  # only the residual stream is used downstream
            return dH, dH, None, None
        has = dH is not None
        dres = dH.reshape(-1, n) if has else h
        dx, dw = torch.ops.fat5.add_rmsnorm_bwd(dY.reshape(-1, n), h, W, rstd, dres, has)
        dx = dx.reshape(ctx.shape)
        return dx, dx, dw, None


def fused_add_rms_layernorm(X, residual, W, eps):
    """(h, y) with h = X + residual (rounded to X's dtype) and y = fast_rms_layernorm(h, W, eps), in one pass; differentiable in
    X, residual and W.  Bit-identical to the two separate operations."""
    nat = None if torch.compiler.is_compiling() else _lib.native()
    if nat is not None and X.is_cuda and residual.shape == X.shape and residual.dtype == X.dtype and W.dim() == 1 and W.shape[0] == X.shape[-1]:
        return nat.add_rmsnorm_apply(X, residual, W, float(eps))
    return FusedAddRMSLayernorm.apply(X, residual, W, eps)
