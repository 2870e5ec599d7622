"""GPU parity tests of -inf entries in a dense bias (an additive mask written as -inf instead of finfo.min) -- round 6, ADVICE r5 (medium).

The bodies that add the bias ON THE MATRIX PIPE (S' = Q K^T + E B with a 0 / (1 / scale) selector E: csrc/attn_bwd_qdb64.h, the dense instantiation of
csrc/attn_bwd64.h, and their one-launch form) hand the raw 16-bit bias words to MFMAs: a -inf word would meet the selector's zeros as -inf * 0 = NaN and
poison every score of its k-slot group.  They clamp the words in their packed form first (attn_common.h: bias_mfma_limit, one v_pk_min_u16 per two values).
Reference semantics: p = exp(-inf) = 0 (flash_attention_v2_bias.py:436-443, :452-456); the forward bodies (per-element bias add) are checked beside them.
"""
import pytest
import torch

from attn_helpers import make_inputs, oracle_all, maxdiff
from test_attention_gpu import bound, gbound

pytestmark = pytest.mark.gpu


def _fwd(q, k, v, b, causal, scale, bits):
    from flasht5_amd.flash_attention_v2_bias import AttentionPlan
    plan = AttentionPlan(q, k, v, torch.zeros_like(q), bias=b, causal=causal, sm_scale=scale, variant=bits, need_dbias=False)
    plan.o.fill_(float("nan"))
    plan.forward()
    torch.cuda.synchronize()
    return plan, plan.o.clone(), plan.lse.clone()


def _minus_inf_bias(B, H, M, N, dtype, seed):
    q, k, v, b, do = make_inputs(B, H, M, N, 64, dtype, "1h", seed=seed, strided=True)
    g = torch.Generator().manual_seed(seed)
    hole = (torch.rand(1, H, M, N, generator=g) < 0.05).cuda()
    hole[..., :32] = False                     # every row keeps visible keys
    b = b.masked_fill(hole, float("-inf"))
    b[..., N - 72:N - 8] = float("-inf")       # a band of keys masked for every row (whole 8- and 16-row groups of a tile at -inf)
    return q, k, v, b, do


@pytest.mark.parametrize("B,H,M,N,causal,scale", [
    (4, 2, 512, 512, False, 0.125),   # one selector term
    (4, 2, 512, 512, True, 1.3),      # two terms, causal (mask in the C operand)
    (6, 2, 256, 384, False, 0.25),    # two groups of batch elements (fp32 slabs)
])
@pytest.mark.parametrize("form", ["two-launches", "one-launch"])
def test_minus_inf_bias_through_the_dense_64_wide_backward(B, H, M, N, causal, scale, form):
    from flasht5_amd import _lib
    from flasht5_amd.flash_attention_v2_bias import AttentionPlan
    dtype = torch.bfloat16
    q, k, v, b, do = _minus_inf_bias(B, H, M, N, dtype, seed=B + M)
    ref = oracle_all(q, k, v, b, do, scale, causal)
    bits = _lib.V_QDB64_ON | _lib.V_KV64_ON | (_lib.V_FUSED64_ON if form == "one-launch" else _lib.V_FUSED64_OFF)
    plan = AttentionPlan(q, k, v, do, bias=b, causal=causal, sm_scale=scale, variant=bits)
    d = plan.describe()
    assert d["dq"] == "64row-batch4" and d["dkdv"] == "64key" and d["fused"] == ("1" if form == "one-launch" else "0")
    plan.forward()
    dq, dk, dv, db = (t.clone() for t in plan.backward())
    torch.cuda.synchronize()
    # (o: a row that one key dominates carries the rounding of P to bf16 (reference :458) on top of its own output rounding: two units)
    assert maxdiff(plan.o, ref["o"]) <= bound(ref["o"], dtype, ulps=2.0)
    for got, key in ((dq, "dq"), (dk, "dk"), (dv, "dv")):
        assert torch.isfinite(got.float()).all(), key
        assert maxdiff(got, ref[key]) <= gbound(ref[key], dtype), key
    assert torch.isfinite(db.float()).all()
    assert maxdiff(db, ref["db"]) <= gbound(ref["db"], dtype) * (1 + B)
    assert bool((db[0][torch.isinf(b[0])] == 0).all())  # p = 0 -> dS = 0 exactly where the bias is -inf


@pytest.mark.parametrize("bits_name", ["32row", "64row"])
def test_minus_inf_bias_through_the_forward(bits_name):
    from flasht5_amd import _lib
    bits = {"32row": _lib.V_FWD64_OFF, "64row": _lib.V_FWD64_ON}[bits_name]
    q, k, v, b, do = _minus_inf_bias(2, 2, 512, 512, torch.bfloat16, seed=11)
    ref = oracle_all(q, k, v, b, do, 0.125, False)
    _, o, L = _fwd(q, k, v, b, False, 0.125, bits)
    assert maxdiff(o, ref["o"]) <= bound(ref["o"], torch.bfloat16)
    assert maxdiff(L, ref["L"]) <= 1e-4 * max(1.0, float(ref["L"].abs().max()))
