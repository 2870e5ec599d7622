"""lm_head + cross-entropy (+ label smoothing, z-loss) without the (rows, vocab) logits tensor  (SURVEY 8(f) n3).

The reference computes `lm_logits = self.lm_head(sequence_output)` and hands the whole `(B*T, V)` tensor to the loss
(src/model/modeling_flash_t5.py:725-730): 268 MB of bf16 logits at (4096, 32768) plus their gradient (in place at best,
cross_entropy_loss.py:247).  Here the rows are processed in chunks: a chunk's logits live only between its GEMM and the
cross-entropy kernel that consumes them -- forward keeps the per-row log-sum-exp (4 bytes per row), backward recomputes the
chunk's logits, turns them into their gradient in place (the HIP cross-entropy backward) and feeds the two gradient GEMMs:

    forward,  per chunk c:  logits_c = h_c W^T          -> fat5_ce_fwd  -> losses_c, z_losses_c, lse_c
    backward, per chunk c:  logits_c = h_c W^T          -> fat5_ce_bwd (in place, with lse_c and dlosses_c) -> dlogits_c
                            dh_c = dlogits_c W,   dW += dlogits_c^T h_c

GEMMs are plain library GEMMs (torch.matmul -> hipBLASLt); the cross-entropy kernels are the path's own.  One extra lm_head GEMM
(the recomputation) buys a peak activation footprint of one chunk instead of the full logits and their gradient.
Same semantics as `cross_entropy_loss(hidden @ weight.T, labels, ...)`: per-row `(losses, z_losses)`, z-loss inside the loss,
ignored rows zero, `z_losses` not differentiable.
"""
from typing import Optional, Tuple

import torch

from .cross_entropy_loss import cross_entropy_fwd, cross_entropy_bwd

__all__ = ["lm_head_cross_entropy", "LMHeadCrossEntropy"]


def _chunks(rows, chunk_rows):
    return [(s, min(rows, s + chunk_rows)) for s in range(0, rows, chunk_rows)]


class LMHeadCrossEntropy(torch.autograd.Function):
    @staticmethod
    def forward(ctx, hidden, weight, labels, smoothing, logit_scale, lse_square_scale, ignore_index, chunk_rows):
        rows = hidden.shape[0]
        losses = torch.empty(rows, dtype=torch.float32, device=hidden.device)
        z_losses = torch.empty(rows, dtype=torch.float32, device=hidden.device)
        lse = torch.empty(rows, dtype=torch.float32, device=hidden.device)
        wt = weight.t()
        for s, e in _chunks(rows, chunk_rows):
            logits = hidden[s:e] @ wt                                    # (chunk, V), dies at the end of the iteration
            l, z, ls = cross_entropy_fwd(logits, labels[s:e], None, smoothing, logit_scale, lse_square_scale, ignore_index)
            losses[s:e], z_losses[s:e], lse[s:e] = l, z, ls
        ctx.save_for_backward(hidden, weight, labels, lse)
        ctx.cfg = (smoothing, logit_scale, lse_square_scale, ignore_index, chunk_rows)
        ctx.mark_non_differentiable(z_losses)
        return losses, z_losses

    @staticmethod
    def backward(ctx, grad_losses, grad_z):
        del grad_z
        hidden, weight, labels, lse = ctx.saved_tensors
        smoothing, logit_scale, lse_square_scale, ignore_index, chunk_rows = ctx.cfg
        need_h, need_w = ctx.needs_input_grad[0], ctx.needs_input_grad[1]
        rows = hidden.shape[0]
        dh = torch.empty_like(hidden) if need_h else None
        dw = torch.zeros(weight.shape, dtype=torch.float32, device=weight.device) if need_w else None  # fp32 accumulation over chunks
        g = grad_losses.contiguous().float()
        wt = weight.t()
        for s, e in _chunks(rows, chunk_rows):
            logits = hidden[s:e] @ wt
            cross_entropy_bwd(g[s:e], logits, lse[s:e], labels[s:e], True, smoothing, logit_scale, lse_square_scale, ignore_index)
            if need_h:
                dh[s:e] = logits @ weight                                # logits now holds dlogits (in place)
            if need_w:
                dw.addmm_(logits.t().float(), hidden[s:e].float()) if weight.dtype == torch.float32 else dw.add_(logits.t() @ hidden[s:e])
        return dh, (dw.to(weight.dtype) if need_w else None), None, None, None, None, None, None


def lm_head_cross_entropy(hidden: torch.Tensor, weight: torch.Tensor, labels: torch.Tensor, label_smoothing: float = 0.0,
                          logit_scale: float = 1.0, lse_square_scale: float = 0.0, ignore_index: int = -100,
                          chunk_rows: Optional[int] = None) -> Tuple[torch.Tensor, torch.Tensor]:
    """hidden (..., d_model), weight (vocab, d_model) -- `lm_head.weight` --, labels (...): returns (losses, z_losses), fp32 per row,
    equal to `cross_entropy_loss(hidden @ weight.T, labels, ...)`.  chunk_rows: rows per chunk (default: ~64 MB of logits)."""
    if not hidden.is_cuda:
        raise RuntimeError("flasht5_amd operators need tensors on the HIP device (no CPU fallback)")
    h2 = hidden.reshape(-1, hidden.shape[-1])
    lab = labels.reshape(-1)
    if lab.shape[0] != h2.shape[0] or weight.shape[1] != h2.shape[1]:
        raise ValueError(f"hidden {tuple(hidden.shape)}, weight {tuple(weight.shape)}, labels {tuple(labels.shape)} do not match")
    if chunk_rows is None:
        chunk_rows = max(256, (64 << 20) // (weight.shape[0] * h2.element_size()) // 256 * 256)
    return LMHeadCrossEntropy.apply(h2, weight, lab, float(label_smoothing), float(logit_scale), float(lse_square_scale),
                                    int(ignore_index), int(chunk_rows))
