"""Cross-entropy with label smoothing and z-loss on MI355X -- host-side mirror of the reference
src/model/ops/cross_entropy_loss.py: `cross_entropy_loss(...)` / `CrossEntropyLoss` with the same signature,
return values `(losses, z_losses)` (fp32 per row) and in-place backward option, backed by bandwidth-bound HIP
kernels (libfat5.so).  The vocab-parallel `process_group` path of the reference is dead code there
(SURVEY #14b) and is rejected here."""
import ctypes
from typing import Optional, Tuple

import torch

from . import _lib

__all__ = ["cross_entropy_loss", "CrossEntropyLoss"]


@torch.library.custom_op("fat5::cross_entropy_fwd", mutates_args=(), device_types="cuda")
def cross_entropy_fwd(logits: torch.Tensor, labels: torch.Tensor, precomputed_lse: Optional[torch.Tensor],
                      smoothing: float, logit_scale: float, lse_square_scale: float, ignore_index: int
                      ) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
    """reference cross_entropy_triton_fwd (cross_entropy_loss.py:164-217), single rank."""
    if logits.stride(-1) != 1 or logits.data_ptr() % 16 != 0:
        logits = logits.contiguous()
    n_rows, n_cols = logits.shape
    labels = labels.to(torch.int64).contiguous()
    losses = torch.empty(n_rows, dtype=torch.float32, device=logits.device)
    z_losses = torch.empty(n_rows, dtype=torch.float32, device=logits.device)
    use_pre = precomputed_lse is not None
    if use_pre:
        assert precomputed_lse.shape == (n_rows,)
        lse = precomputed_lse.to(torch.float32).contiguous().clone()
    else:
        lse = torch.empty(n_rows, dtype=torch.float32, device=logits.device)
    if n_rows == 0:
        return losses, z_losses, lse
    with _lib.on_device(logits.device):
        _lib.check(_lib.load().fat5_ce_fwd(
            logits.data_ptr(), labels.data_ptr(), losses.data_ptr(), z_losses.data_ptr(), lse.data_ptr(), n_rows, n_cols,
            logits.stride(0), float(smoothing), float(logit_scale), float(lse_square_scale), int(ignore_index),
            int(use_pre), _lib.dtype_code(logits.dtype), _lib.stream_ptr(logits.device)), "fat5_ce_fwd")
    return losses, z_losses, lse


@torch.library.register_fake("fat5::cross_entropy_fwd")
def _ce_fwd_fake(logits, labels, precomputed_lse, smoothing, logit_scale, lse_square_scale, ignore_index):
    n = logits.shape[0]
    mk = lambda: torch.empty(n, dtype=torch.float32, device=logits.device)  # noqa: E731
    return mk(), mk(), mk()


@torch.library.custom_op("fat5::cross_entropy_bwd", mutates_args={"logits"}, device_types="cuda")
def cross_entropy_bwd(dlosses: torch.Tensor, logits: torch.Tensor, lse: torch.Tensor, labels: torch.Tensor,
                      inplace_backward: bool, smoothing: float, logit_scale: float, lse_square_scale: float,
                      ignore_index: int) -> torch.Tensor:
    """reference cross_entropy_triton_bwd (cross_entropy_loss.py:228-274).  With `inplace_backward` the
    gradient overwrites `logits` and an empty tensor is returned."""
    n_rows, n_cols = logits.shape
    src = logits
    if logits.stride(-1) != 1 or logits.data_ptr() % 16 != 0:
        if inplace_backward:
            raise RuntimeError("inplace_backward needs logits with unit inner stride and 16-byte alignment")
        src = logits.contiguous()
    dlogits = src if inplace_backward else torch.empty((n_rows, n_cols), dtype=logits.dtype, device=logits.device)
    labels = labels.to(torch.int64).contiguous()
    dlosses = dlosses.to(torch.float32)
    if n_rows > 0:
        with _lib.on_device(logits.device):
            _lib.check(_lib.load().fat5_ce_bwd(
                dlosses.data_ptr(), dlosses.stride(0), src.data_ptr(), lse.data_ptr(), labels.data_ptr(), dlogits.data_ptr(),
                n_rows, n_cols, src.stride(0), dlogits.stride(0), float(smoothing), float(logit_scale),
                float(lse_square_scale), int(ignore_index), _lib.dtype_code(logits.dtype), _lib.stream_ptr(logits.device)),
                "fat5_ce_bwd")
    if inplace_backward:
        return torch.empty(0, dtype=logits.dtype, device=logits.device)
    return dlogits


@torch.library.register_fake("fat5::cross_entropy_bwd")
def _ce_bwd_fake(dlosses, logits, lse, labels, inplace_backward, smoothing, logit_scale, lse_square_scale, ignore_index):
    if inplace_backward:
        return torch.empty(0, dtype=logits.dtype, device=logits.device)
    return torch.empty(logits.shape, dtype=logits.dtype, device=logits.device)


def cross_entropy_fwd_bwd_(logits: torch.Tensor, labels: torch.Tensor, dlosses: torch.Tensor, losses: torch.Tensor, z_losses: torch.Tensor,
                           lse: torch.Tensor, smoothing: float, logit_scale: float, lse_square_scale: float, ignore_index: int) -> None:
    """forward AND backward of every row in one launch (fat5_ce_fwd_bwd; the row is read once): writes losses / z_losses / lse (fp32, one
    per row, caller-provided slices) and overwrites `logits` with d loss / d logits for the upstream gradients `dlosses` (fp32, one per row
    or one element with stride 0 via expand).  Same values as cross_entropy_fwd followed by cross_entropy_bwd(inplace).  For callers that
    know the upstream gradient before the forward (lm_head_cross_entropy, reduction="mean")."""
    if logits.stride(-1) != 1 or logits.data_ptr() % 16 != 0:
        raise RuntimeError("cross_entropy_fwd_bwd_ needs logits with unit inner stride and 16-byte alignment")
    n_rows, n_cols = logits.shape
    if n_rows == 0:
        return
    labels = labels.to(torch.int64).contiguous()
    assert dlosses.dtype == torch.float32 and losses.dtype == z_losses.dtype == lse.dtype == torch.float32
    assert losses.is_contiguous() and z_losses.is_contiguous() and lse.is_contiguous()
    with _lib.on_device(logits.device):
        _lib.check(_lib.load().fat5_ce_fwd_bwd(
            logits.data_ptr(), labels.data_ptr(), dlosses.data_ptr(), dlosses.stride(0), losses.data_ptr(), z_losses.data_ptr(), lse.data_ptr(),
            logits.data_ptr(), n_rows, n_cols, logits.stride(0), logits.stride(0), float(smoothing), float(logit_scale),
            float(lse_square_scale), int(ignore_index), _lib.dtype_code(logits.dtype), _lib.stream_ptr(logits.device)), "fat5_ce_fwd_bwd")


class CrossEntropyLoss(torch.autograd.Function):
    """Same contract as the reference class (cross_entropy_loss.py:280-385)."""

    @staticmethod
    def forward(ctx, logits, labels, precomputed_lse=None, smoothing=0.0, logit_scale=1.0, lse_square_scale=0.0,
                ignore_index=-100, inplace_backward=False, process_group=None):
        if process_group is not None:
            raise NotImplementedError("vocab-parallel cross entropy (process_group) is not part of this path")
        n_rows, n_cols = logits.shape
        assert labels.shape == (n_rows,)
        # the reference only honours a precomputed LSE without scaling / smoothing (:307)
        use_pre = precomputed_lse is not None and logit_scale == 1.0 and smoothing == 0.0
        losses, z_losses, lse = torch.ops.fat5.cross_entropy_fwd(
            logits, labels, precomputed_lse if use_pre else None, float(smoothing), float(logit_scale),
            float(lse_square_scale), int(ignore_index))
        ctx.save_for_backward(logits, lse, labels)
        ctx.mark_non_differentiable(z_losses)
        ctx.cfg = (float(smoothing), float(logit_scale), float(lse_square_scale), int(ignore_index), bool(inplace_backward))
        return losses, z_losses

    @staticmethod
    def backward(ctx, grad_losses, grad_z_losses):
        del grad_z_losses  # z_losses are only for logging (reference :367)
        logits, lse, labels = ctx.saved_tensors
        smoothing, logit_scale, lse_square_scale, ignore_index, inplace = ctx.cfg
        dlogits = torch.ops.fat5.cross_entropy_bwd(grad_losses.contiguous(), logits, lse, labels, inplace, smoothing,
                                                   logit_scale, lse_square_scale, ignore_index)
        if inplace:
            dlogits = logits
        return dlogits, None, None, None, None, None, None, None, None


def cross_entropy_loss(logits: torch.Tensor, labels: torch.Tensor, precomputed_lse: Optional[torch.Tensor] = None,
                       label_smoothing: float = 0.0, logit_scale: float = 1.0, lse_square_scale: float = 0.0,
                       ignore_index=-100, inplace_backward: bool = False, process_group=None
                       ) -> Tuple[torch.Tensor, torch.Tensor]:
    """Same arguments and return as the reference (cross_entropy_loss.py:388-426):
    returns (losses, z_losses), both (rows,) fp32; z_losses is not differentiable."""
    return CrossEntropyLoss.apply(logits.view(-1, logits.shape[-1]), labels.view(-1), precomputed_lse, label_smoothing,
                                  logit_scale, lse_square_scale, ignore_index, inplace_backward, process_group)
