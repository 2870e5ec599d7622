"""CPU restatement of the reference's op pairs that `fat5_linear_fused` fuses (TEST INFRASTRUCTURE ONLY).

  * pre-norm -> projection: `normed = self.layer_norm(hidden_states)` then `self.Wq(normed)` etc.
    (reference src/model/modeling_flash_t5.py:304-318 with FlashT5LayerNorm.forward :95-112 and FlashT5Attention.forward :226-231;
     feed-forward: :159-160 with :126-131)
  * projection -> residual: `hidden_states + self.o(attn)` / `hidden_states + self.wo(...)` (:316, :162-163)
fp32 math on the given (low-precision) values -- the truth the GPU results are compared with; gradients by autograd on it."""
import torch

from .rmsnorm import rmsnorm_fwd_oracle


def rmsnorm_linear_oracle(x, norm_weight, weight, eps):
    """(x rstd g) W^T in fp32; y is NOT rounded to the activation dtype in between (the reference rounds it: that rounding is part
    of what the comparison's tolerance covers).  Returns (out fp32, rstd)."""
    xf = x.float()
    var = (xf * xf).mean(-1, keepdim=True)
    rstd = torch.rsqrt(var + eps)
    y = xf * rstd * norm_weight.float()
    return y @ weight.float().t(), rstd.squeeze(-1)


def rmsnorm_linear_reference_rounding(x, norm_weight, weight, eps):
    """the same with the reference's intermediate rounding: y = layer_norm(x) rounded to x's dtype (rms_norm.py:45-60), then
    the Linear in that dtype's values"""
    y, _ = rmsnorm_fwd_oracle(x, norm_weight, eps)
    return y.float() @ weight.float().t()


def linear_residual_oracle(a, weight, residual):
    return residual.float() + a.float() @ weight.float().t()
