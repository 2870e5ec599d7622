"""GPU parity fuzz of the public operators at the library's own dispatch: 32 random problems per run from a fixed seed (tests/api_fuzz.py: every bias broadcast form
the reference accepts -- flash_attention_v2_bias.py:45-52 --, the T5 table mode, head_dim 16 .. 128, 1 .. 700 rows / keys, -inf and finfo.min key padding as the model
builds it -- modeling_flash_t5.py:267-277 --, strided operands), each tensor against the fp32 oracle with the bounds of tests/test_attention_gpu.py or, past them,
the reference's own rule (tests/fa2_triton/test_fa2_bias.py:64-67).  tools/fuzz_api.py runs the same generator for any count / seed."""
import random

import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("block", range(4))
def test_public_operators_on_random_problems(block):
    from api_fuzz import run_case
    rng = random.Random(100 + block)
    failures = []
    for i in range(8):
        desc, msgs = run_case(8 * block + i, rng)
        if msgs:
            failures.append(f"{desc}: {'; '.join(msgs)}")
    assert not failures, failures


def test_packed_batches_on_random_problems():
    """flash_attn_varlen_func: random packed batches incl. empty and one-token sequences, cross-attention or the in-kernel T5 bias"""
    from api_fuzz import run_varlen_case
    rng = random.Random(300)
    failures = []
    for i in range(12):
        desc, msgs = run_varlen_case(i, rng)
        if msgs:
            failures.append(f"{desc}: {'; '.join(msgs)}")
    assert not failures, failures


def test_rowwise_operators_on_random_problems():
    """fast_rms_layernorm and cross_entropy_loss (label smoothing, z-loss, logit scale, ignored / out-of-range labels, in-place backward) on random shapes"""
    from api_fuzz import run_rowwise_case
    rng = random.Random(400)
    failures = []
    for i in range(24):
        desc, msgs = run_rowwise_case(i, rng)
        if msgs:
            failures.append(f"{desc}: {'; '.join(msgs)}")
    assert not failures, failures
