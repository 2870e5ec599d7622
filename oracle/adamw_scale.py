"""CPU restatement of the reference optimizer step `AdamWScale` (src/utils/adamw_scaled.py:10-211, the per-tensor path
`_adamwscaled` :154-211) -- TEST INFRASTRUCTURE ONLY: checker of the fused HIP optimizer kernel.

AdamW whose step is additionally scaled by max(1e-3, rms(p)) (Adafactor's relative step, :186), optional Kahan-compensated
parameter update for 16-bit parameters (:188-198), decoupled weight decay applied after the step with the unscaled lr (:209-210).
Every line below is one in-place torch op of the reference, in its order and in the tensors' own dtypes (so the intermediate
roundings of bf16 / fp16 states are the reference's).  Pinned by tests/golden/make_golden.py (gen_adamw: imports the reference
class, runs both for several steps on the same seeded tensors, asserts torch.equal, freezes fixtures)."""
import math

import torch


def rms(t):
    return t.norm(2) / (t.numel() ** 0.5)                                   # :69-70


def adamw_scale_step(p, grad, exp_avg, exp_avg_sq, kahan_comp, step, lr, beta1, beta2, weight_decay, eps, correct_bias=True):
    """One step on one tensor, in place (p, exp_avg, exp_avg_sq, kahan_comp; `grad` is clobbered in the Kahan path like in the
    reference, which uses it as scratch :193-197).  `step` = the incremented step count (1 for the first call)."""
    step = torch.as_tensor(step, dtype=torch.int32)                          # the reference keeps it as an int32 tensor (:98, :120):
    #                                                                          beta ** step is then a float32 tensor, not a double
    exp_avg.mul_(beta1).add_(grad, alpha=(1.0 - beta1))                      # :173
    exp_avg_sq.mul_(beta2).addcmul_(grad, grad, value=(1.0 - beta2))         # :174
    denom = exp_avg_sq.sqrt().add_(eps)                                      # :175
    step_size = lr
    if correct_bias:                                                         # :178-181
        bias_correction1 = 1.0 - beta1 ** step
        bias_correction2 = 1.0 - beta2 ** step
        step_size = step_size * math.sqrt(bias_correction2) / bias_correction1
    step_size = step_size * max(1e-3, rms(p.data))                           # :184  (a 0-dim tensor in p's dtype from here on)
    if kahan_comp is not None:
        kahan_comp.addcdiv_(exp_avg, denom, value=-step_size)                # :190
        grad.copy_(p)                                                        # :193
        p.add_(kahan_comp)                                                   # :194
        grad.sub_(p, alpha=1)                                                # :197
        kahan_comp.add_(grad, alpha=1)                                       # :198
    else:
        p.addcdiv_(exp_avg, denom, value=-step_size)                         # :200
    if weight_decay > 0.0:
        p.add_(p, alpha=(-lr * weight_decay))                                # :210
    return p
