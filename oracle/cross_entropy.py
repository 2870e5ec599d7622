"""CPU restatement of the reference cross-entropy + label smoothing + z-loss
(TEST INFRASTRUCTURE ONLY).  Reference: src/model/ops/cross_entropy_loss.py:35-162
(single-rank path: SPLIT=False, class_start_idx=0, total_classes=n_cols)."""
import torch


def ce_fwd_oracle(logits, labels, smoothing=0.0, logit_scale=1.0, lse_square_scale=0.0,
                  ignore_index=-100):
    """:60-111.  Returns (losses, z_losses, lse) fp32 per row."""
    lg = logits.float() * logit_scale
    lse = torch.logsumexp(lg, dim=-1)
    V = lg.shape[-1]
    ignored = labels == ignore_index
    safe = labels.clamp(min=0, max=V - 1)
    picked = lg.gather(-1, safe.unsqueeze(-1)).squeeze(-1)
    inb = (labels >= 0) & (labels < V)
    if smoothing > 0.0:
        sum_logits = lg.sum(-1)
        loss_in = lse - smoothing * sum_logits / V - (1 - smoothing) * picked      # :90-95
        loss_out = smoothing * (lse - sum_logits / V)                               # :100-101
    else:
        loss_in = lse - picked                                                      # :97
        loss_out = torch.zeros_like(lse)                                            # :103
    loss = torch.where(inb, loss_in, loss_out)
    z = lse_square_scale * lse * lse                                                # :105
    loss = loss + z
    loss = torch.where(ignored, torch.zeros_like(loss), loss)                       # :83-85
    z = torch.where(ignored, torch.zeros_like(z), z)
    return loss, z, lse


def ce_bwd_oracle(dlosses, logits, lse, labels, smoothing=0.0, logit_scale=1.0,
                  lse_square_scale=0.0, ignore_index=-100):
    """:137-162.  dlogits in logits dtype."""
    lg = logits.float() * logit_scale
    V = lg.shape[-1]
    probs = torch.exp(lg - lse.unsqueeze(-1))
    probs = probs + 2.0 * lse_square_scale * lse.unsqueeze(-1) * probs              # :153-154
    onehot = torch.zeros_like(probs)
    inb = (labels >= 0) & (labels < V)
    rows = torch.nonzero(inb).squeeze(-1)
    onehot[rows, labels[rows]] = 1.0
    if smoothing > 0.0:
        probs = probs - (1.0 - smoothing) * onehot - smoothing / V                  # :156-159
    else:
        probs = probs - onehot                                                      # :161
    dl = torch.where(labels == ignore_index, torch.zeros_like(dlosses.float()), dlosses.float())
    return ((dl * logit_scale).unsqueeze(-1) * probs).to(logits.dtype)              # :145-148,:162
