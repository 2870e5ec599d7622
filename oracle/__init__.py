"""CPU oracle for the FA2 + T5-bias hot path (TEST INFRASTRUCTURE ONLY).

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import this package.  The product path (``flasht5_amd``) never does.
"""
from .attention import (attn_ref, attn_fwd_oracle, attn_bwd_oracle,
                        attn_varlen_oracle, attn_varlen_bwd_oracle)
from .rpe import (relative_position_bucket, compute_bias, bias1d_from_table,
                  table_grad_from_dbias1d, toeplitz_from_bias1d)
from .rmsnorm import rmsnorm_fwd_oracle, rmsnorm_bwd_oracle, rmsnorm_eager
from .cross_entropy import ce_fwd_oracle, ce_bwd_oracle
from .adamw_scale import adamw_scale_step
from .fused_linear import rmsnorm_linear_oracle, rmsnorm_linear_reference_rounding, linear_residual_oracle
from .gated_act import gated_act_oracle, gated_act_bwd_oracle
