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

from .cross_entropy_loss import cross_entropy_fwd, cross_entropy_bwd, cross_entropy_fwd_bwd_

__all__ = ["lm_head_cross_entropy", "LMHeadCrossEntropy", "LMHeadCrossEntropyMean"]


def _chunks(rows, chunk_rows):
    return [(s, min(rows, s + chunk_rows)) for s in range(0, rows, chunk_rows)]


def _dh_gemm(dlogits, weight, out=None):
    """dlogits (m, V) @ weight (V, K) for a row chunk: m x K is a few dozen output tiles with a contraction over the whole vocabulary
    -- the library runs it on a fraction of the chip (212 us against 117 us for the chunk's logits GEMM of the same flops, measured
    at m = 2048, V = 32768, K = 768).  Split the contraction four ways (one batched GEMM, fp32 partials) and add the parts."""
    m, V = dlogits.shape
    K = weight.shape[1]
    if dlogits.dtype == torch.float32 or V % 4 or (m // 256) * (K // 128) >= 192 or not weight.is_contiguous():
        r = dlogits @ weight
        return r if out is None else out.copy_(r)
    part = torch.bmm(dlogits.view(m, 4, V // 4).transpose(0, 1), weight.view(4, V // 4, K), out_dtype=torch.float32)
    if out is None:
        return part.sum(0).to(dlogits.dtype)
    return out.copy_(torch.sum(part, 0, out=torch.empty((m, K), dtype=torch.float32, device=part.device)))   # (one fp32 sum, one converting copy into the caller's rows)


class LMHeadCrossEntropy(torch.autograd.Function):
    @staticmethod
    def forward(ctx, hidden, weight, labels, smoothing, logit_scale, lse_square_scale, ignore_index, chunk_rows):
        rows = hidden.shape[0]
        losses = torch.empty(rows, dtype=torch.float32, device=hidden.device)
        z_losses = torch.empty(rows, dtype=torch.float32, device=hidden.device)
        lse = torch.empty(rows, dtype=torch.float32, device=hidden.device)
        wt = weight.t()
        buf = torch.empty((min(rows, chunk_rows), weight.shape[0]), dtype=hidden.dtype, device=hidden.device)  # ONE chunk of logits, reused (a fresh tensor per chunk is allocated before the previous one dies: two chunks alive)
        for s, e in _chunks(rows, chunk_rows):
            logits = torch.mm(hidden[s:e], wt, out=buf[:e - s])          # (chunk, V)
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
        buf = torch.empty((min(rows, chunk_rows), weight.shape[0]), dtype=hidden.dtype, device=hidden.device)  # ONE chunk of logits, reused (a fresh tensor per chunk is allocated before the previous one dies: two chunks alive)
        for s, e in _chunks(rows, chunk_rows):
            logits = torch.mm(hidden[s:e], wt, out=buf[:e - s])
            cross_entropy_bwd(g[s:e], logits, lse[s:e], labels[s:e], True, smoothing, logit_scale, lse_square_scale, ignore_index)
            if need_h:
                dh[s:e] = _dh_gemm(logits, weight)                       # logits now holds dlogits (in place)
            if need_w:
                if weight.dtype == torch.float32:
                    dw.addmm_(logits.t(), hidden[s:e])
                else:
                    torch.addmm(dw, logits.t(), hidden[s:e], out_dtype=torch.float32, out=dw)
        return dh, (dw.to(weight.dtype) if need_w else None), None, None, None, None, None, None


class LMHeadCrossEntropyMean(torch.autograd.Function):
    """The MEAN of the per-row losses (what the model takes: `cross_entropy_loss(...)[0].mean()`, modeling_flash_t5.py:64-68) with the
    gradients formed in the FORWARD pass, chunk by chunk (round 4): the upstream gradient of a mean is one scalar, so every row's
    dlogits = (1 / rows) d loss_r / d logits_r is known as soon as the chunk's loss is -- no recomputation of the logits in the
    backward (three lm_head-sized GEMMs per step like the unfused form instead of four), which only scales dh and dW by that scalar.

        per chunk c:  logits_c = h_c W^T -> fat5_ce_fwd -> sum of losses | fat5_ce_bwd in place (dlosses = 1 / rows) -> dlogits_c
                      dh_c = dlogits_c W,   dW += dlogits_c^T h_c   (fp32 accumulation over the chunks)"""

    @staticmethod
    def forward(ctx, hidden, weight, labels, smoothing, logit_scale, lse_square_scale, ignore_index, chunk_rows):
        rows = hidden.shape[0]
        need_h, need_w = ctx.needs_input_grad[0], ctx.needs_input_grad[1]
        dev = hidden.device
        losses = torch.empty(rows, dtype=torch.float32, device=dev)   # per-row values: summed ONCE at the end (fixed order)
        zs = torch.empty(rows, dtype=torch.float32, device=dev)
        lse = torch.empty(rows, dtype=torch.float32, device=dev)
        dh = torch.empty_like(hidden) if need_h else None
        dw = torch.empty(weight.shape, dtype=torch.float32, device=dev) if need_w else None
        wt = weight.t()
        g = torch.full((1,), 1.0 / max(rows, 1), dtype=torch.float32, device=dev)
        buf = torch.empty((min(rows, chunk_rows), weight.shape[0]), dtype=hidden.dtype, device=hidden.device)  # ONE chunk of logits, reused (a fresh tensor per chunk is allocated before the previous one dies: two chunks alive)
        for s, e in _chunks(rows, chunk_rows):
            logits = torch.mm(hidden[s:e], wt, out=buf[:e - s])
            if need_h or need_w:
                # loss and gradient of the chunk's rows in one launch, the logits read once (fat5_ce_fwd_bwd)
                cross_entropy_fwd_bwd_(logits, labels[s:e], g.expand(e - s), losses[s:e], zs[s:e], lse[s:e], smoothing, logit_scale,
                                       lse_square_scale, ignore_index)
                if need_h:
                    _dh_gemm(logits, weight, out=dh[s:e])                # logits now holds dlogits (in place)
                if need_w:  # (fp32 accumulation over the chunks inside the GEMM: bf16 operands, fp32 output added to the accumulator;
                    #          the first chunk writes the accumulator: no 100 MB memset, no read of zeros)
                    if weight.dtype == torch.float32:
                        torch.mm(logits.t(), hidden[s:e], out=dw) if s == 0 else dw.addmm_(logits.t(), hidden[s:e])
                    elif s == 0:
                        torch.mm(logits.t(), hidden[s:e], out_dtype=torch.float32, out=dw)
                    else:
                        torch.addmm(dw, logits.t(), hidden[s:e], out_dtype=torch.float32, out=dw)
            else:
                l, z, _ = cross_entropy_fwd(logits, labels[s:e], None, smoothing, logit_scale, lse_square_scale, ignore_index)
                losses[s:e], zs[s:e] = l, z
        if rows == 0 and dw is not None:
            dw.zero_()
        total, ztotal = losses.sum(), zs.sum()
        if need_w and dw.dtype != weight.dtype:
            dw = dw.to(weight.dtype)
        ctx.save_for_backward(dh, dw)
        zmean = ztotal / max(rows, 1)
        ctx.mark_non_differentiable(zmean)
        return total / max(rows, 1), zmean

    @staticmethod
    def backward(ctx, grad_loss, grad_z):
        del grad_z
        dh, dw = ctx.saved_tensors
        # the upstream scalar is applied in fp32 (rounded to bf16 first, a factor like 1/3 under gradient accumulation is off by
        # up to 2^-9 and scales every gradient upstream of lm_head; the per-row form and the reference apply dloss in fp32 inside the CE kernel)
        gl = grad_loss.float()

        def rescale(t):
            return None if t is None else (t if t.dtype == torch.float32 else t.float()).mul(gl).to(t.dtype)
        return (rescale(dh), rescale(dw), None, None, None, None, None, None)


def lm_head_cross_entropy(hidden: torch.Tensor, weight: torch.Tensor, labels: torch.Tensor, label_smoothing: float = 0.0,
                          logit_scale: float = 1.0, lse_square_scale: float = 0.0, ignore_index: int = -100,
                          chunk_rows: Optional[int] = None, reduction: str = "none") -> Tuple[torch.Tensor, torch.Tensor]:
    """hidden (..., d_model), weight (vocab, d_model) -- `lm_head.weight` --, labels (...): returns (losses, z_losses), fp32 per row,
    equal to `cross_entropy_loss(hidden @ weight.T, labels, ...)`.  chunk_rows: rows per chunk (default: ~64 MB of logits, ~128 MB in the mean form).
    reduction="mean": returns (losses.mean(), z_losses.mean()) over ALL rows -- two scalars -- with the gradients formed during the
    forward pass (LMHeadCrossEntropyMean): no recomputation, the fast form for a training step."""
    if not hidden.is_cuda:
        raise RuntimeError("flasht5_amd operators need tensors on the HIP device (no CPU fallback)")
    if reduction not in ("none", "mean"):
        raise ValueError("reduction must be 'none' or 'mean'")
    h2 = hidden.reshape(-1, hidden.shape[-1])
    lab = labels.reshape(-1)
    if lab.shape[0] != h2.shape[0] or weight.shape[1] != h2.shape[1]:
        raise ValueError(f"hidden {tuple(hidden.shape)}, weight {tuple(weight.shape)}, labels {tuple(labels.shape)} do not match")
    if chunk_rows is None:  # ~64 MB of logits per chunk; the mean form (three GEMMs per chunk, no recomputation) takes chunks twice as long
        chunk_rows = max(256, ((128 if reduction == "mean" else 64) << 20) // (weight.shape[0] * h2.element_size()) // 256 * 256)
    fn = LMHeadCrossEntropyMean if reduction == "mean" else LMHeadCrossEntropy
    return fn.apply(h2, weight, lab, float(label_smoothing), float(logit_scale), float(lse_square_scale), int(ignore_index), int(chunk_rows))
