"""The mixed launch of the 64-row forward (round 4): 256-row workgroups and key-split 128-row workgroups of the same problem in ONE launch
(attn_fwd64_mixed_kernel) -- every query row must be owned by exactly one workgroup of one kind, whatever the split of a pair's rows."""
import pytest
import torch

from attn_helpers import make_inputs, maxdiff, oracle_all
from test_attention_gpu import bound, gbound, _rpe_case

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("B,H,M,N,mode,dtype", [
    (2, 4, 1024, 1024, "none", torch.bfloat16),    # 8 pairs, one per XCD: 2 + 4 workgroups per pair
    (1, 8, 640, 900, "rpe", torch.bfloat16),       # rows not a multiple of 256 / 128: the last 128-row workgroup is partly empty
    (2, 8, 1152, 1536, "rpe", torch.float16),      # 16 pairs: two per XCD with different numbers of 256-row workgroups
    (1, 8, 2048, 2048, "rpe", torch.bfloat16),
    (3, 8, 400, 1300, "none", torch.bfloat16),     # 384 <= M < 512: one 256-row workgroup and two 128-row ones per pair
])
def test_fwd64_mixed_launch_matches_oracle(B, H, M, N, mode, dtype):
    from flasht5_amd import _lib, flash_attention_v2_bias, flash_attention_v2_rpe
    kw = dict(bias_mode=_lib.BIAS_RPE1D, radius=128) if mode == "rpe" else {}
    with _lib.variant(_lib.V_FWD64_ON | _lib.V_FWD64_MIX_ON):
        assert _lib.describe(B=B, H=H, M=M, N=N, dtype=(_lib.FAT5_F16 if dtype == torch.float16 else _lib.FAT5_BF16),
                             variant=_lib.V_FWD64_ON | _lib.V_FWD64_MIX_ON, **kw)["fwd"] == "64row-mixed"
        if mode == "rpe":
            q, k, v, do, table, bias = _rpe_case(B, H, M, N, dtype, False, True, 128, seed=M + N)
            leaves = [t.detach().clone().requires_grad_() for t in (q, k, v)]
            o = flash_attention_v2_rpe(leaves[0], leaves[1], leaves[2], table.cuda(), True, 32, 128, False, 0.125)
        else:
            q, k, v, bias, do = make_inputs(B, H, M, N, 64, dtype, None, seed=M + N, strided=True)
            leaves = [t.detach().clone().requires_grad_() for t in (q, k, v)]
            o = flash_attention_v2_bias(leaves[0], leaves[1], leaves[2], None, False, 0.125)
        dq, dk, dv = torch.autograd.grad(o, leaves, do)
    ref = oracle_all(q, k, v, bias, do, 0.125, False)
    assert torch.isfinite(o.float()).all()
    assert maxdiff(o, ref["o"]) <= bound(ref["o"], dtype)
    for got, key in ((dq, "dq"), (dk, "dk"), (dv, "dv")):   # (the backward consumes this forward's lse)
        assert maxdiff(got, ref[key]) <= gbound(ref[key], dtype), key
    # the same problem through the pure forms: identical rows (each row's arithmetic does not depend on its workgroup's form)
    with _lib.variant(_lib.V_FWD64_ON | _lib.V_FWD64_MIX_OFF | _lib.V_FWD64_KSPLIT_OFF):
        if mode == "rpe":
            o2 = flash_attention_v2_rpe(q, k, v, table.cuda(), True, 32, 128, False, 0.125)
        else:
            o2 = flash_attention_v2_bias(q, k, v, None, False, 0.125)
    assert maxdiff(o.detach(), o2) <= bound(ref["o"], dtype)


@pytest.mark.parametrize("boost,rows,at,dtype", [(72.0, "all", 1024, torch.bfloat16), (1000.0, "even", 768, torch.bfloat16), (-60.0, "all", 0, torch.bfloat16),
                                                 (14.0, "all", 512, torch.float16), (30.0, "all", 768, torch.float16)])
def test_fwd64_mixed_launch_exact_pass_fallback(boost, rows, at, dtype):
    """rows whose sums leave the range of the sweep without a running maximum (attn_fwd64.h) rerun their workgroup through the exact pass --
    in both workgroup forms of the mixed launch (8 pairs of 2048 rows: 256-row workgroups for the first rows of a pair, key-split ones
    for the rest); fp16: scores beyond the first tile's row maximum by more than the format's range."""
    from flasht5_amd import _lib, flash_attention_v2_bias
    from attn_helpers import eager_lowprec_errors
    B, H, S, D = 1, 8, 2048, 64
    g = torch.Generator().manual_seed(11)
    q = torch.randn(B, H, S, D, generator=g).to(dtype)
    k = torch.randn(B, H, S, D, generator=g).to(dtype)
    v = torch.randn(B, H, S, D, generator=g).to(dtype)
    q[..., 0] = 4.0
    if rows == "even":
        q[..., 1::2, 0] = 0.0
    k[..., at:, 0] = boost / 4.0
    q, k, v = q.cuda(), k.cuda(), v.cuda()
    do = torch.randn(B, H, S, D, generator=g).to(dtype).cuda()
    with _lib.variant(_lib.V_FWD64_ON | _lib.V_FWD64_MIX_ON):
        leaves = [t.detach().clone().requires_grad_() for t in (q, k, v)]
        o = flash_attention_v2_bias(leaves[0], leaves[1], leaves[2], None, False, 1.0)
        grads = torch.autograd.grad(o, leaves, do)
    ref = oracle_all(q, k, v, None, do, 1.0, False)
    assert torch.isfinite(o.float()).all()
    assert maxdiff(o, ref["o"]) <= bound(ref["o"], dtype)
    lp = eager_lowprec_errors(q, k, v, None, do, 1.0, False, ref)
    for got, key in zip(grads, ("dq", "dk", "dv")):
        e = maxdiff(got, ref[key])
        assert torch.isfinite(got.float()).all(), key
        assert e <= max(gbound(ref[key], dtype), 3 * lp[key]), (key, e, lp[key])
