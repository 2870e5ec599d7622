"""flasht5_amd -- MI355X-native (gfx950) FlashAttention-2 with T5 relative-position bias, T5 RMSNorm and
cross-entropy + z-loss, as drop-in replacements of the reference's three operator callables
(modeling_flash_t5.py:15-28):

    from flasht5_amd import flash_attention_v2_bias, fast_rms_layernorm, cross_entropy_loss

Everything runs in hand-written HIP kernels behind the C ABI of `lib/libfat5.so` (include/fat5.h); there is
no CPU / PyTorch fallback -- importing raises if the library has not been built.
"""
from . import _lib

_lib.load()  # fail loudly at import time when the HIP library is missing

from .flash_attention_v2_bias import (flash_attention_v2_bias, FlashAttentionAdditiveBias,  # noqa: E402
                                      flash_attention_v2_rpe, FlashAttentionRPE, flash_attention_v2_rpe1d, FlashAttentionRPE1D,
                                      flash_attn_varlen_fwd, flash_attn_varlen_bwd,
                                      flash_attn_varlen_func, FlashAttentionVarlen)
from .rms_norm import fast_rms_layernorm, Fast_RMS_Layernorm, fused_add_rms_layernorm, FusedAddRMSLayernorm  # noqa: E402
from .cross_entropy_loss import cross_entropy_loss, CrossEntropyLoss  # noqa: E402
from .lm_head_cross_entropy import lm_head_cross_entropy, LMHeadCrossEntropy  # noqa: E402
from .fused_linear import rmsnorm_linear, linear_residual, RMSNormLinear, LinearResidual  # noqa: E402
from .gated_act import gated_act, gated_act_packed  # noqa: E402
from .positional_encoding import (relative_position_bucket, compute_bias, rpe1d_from_table,  # noqa: E402
                                  RelativePositionalEncoding)

from .attention_module import FlashT5Attention  # noqa: E402
from .modules import FlashT5LayerNorm, FlashT5CrossEntropyLoss  # noqa: E402
from .adamw_scaled import AdamWScale  # noqa: E402
from .fat5_step import FAT5Config, FAT5ForConditionalGeneration, allreduce_gradients, train_step, GraphedTrainStep  # noqa: E402

__version__ = "0.1.0"
