"""GPU parity tests of the table gradient's per-diagonal sums formed by the dQ workgroups (round 6: csrc/attn_bwd64.h, attn_bwd_q64_body<..., QDG> beside
attn_bwd_kv64_body<..., NODIAG>) -- the form the one-launch 64-wide backward of a short sequence takes by itself where every workgroup is resident at once (cfg2),
forced here with FAT5_V_QDIAG_ON on shapes the oracle finishes in seconds.

Reference: the bias gradient is the dS tensor summed over the batch (src/model/ops/flash_attention_v2_bias.py:214-215); with the Toeplitz T5 bias
(src/utils/positional_encoding.py:100-101) that sum runs along diagonals and lands in the (num_buckets, H) table.  Checked: against the fp32 oracle (same allowance
as tests/test_bwd64_gpu.py), against the dK/dV-side form (FAT5_V_QDIAG_OFF: same terms, another fp32 summation order), dq / dk / dv bit-identical between the two
forms (the diagonal machinery touches none of them), the stage-by-stage call, unit ranges."""
import pytest
import torch

from attn_helpers import oracle_all, maxdiff
from test_attention_gpu import gbound, _rpe_case
from test_bwd64_gpu import _table_truth

pytestmark = pytest.mark.gpu


def _plan(q, k, v, do, table, md, causal, bits, units=None, scale=0.125):
    from flasht5_amd.flash_attention_v2_bias import AttentionPlan
    from flasht5_amd import positional_encoding as pe
    H = q.shape[1]
    tb = table.cuda()
    plan = AttentionPlan(q, k, v, do, causal=causal, sm_scale=scale, variant=bits, units=units,
                         rpe1d=pe.rpe1d_from_table(tb, True, 32, md), radius=pe.rpe_radius(md),
                         rpe_bucket=pe.bucket_index32(pe.rpe_radius(md), True, 32, md, q.device), num_buckets=32)
    plan.forward()
    plan.ws.view(torch.uint8).fill_(255)  # (NaN patterns in the workspace: nothing may be read that was not written)
    plan.dbias.fill_(float("nan"))
    return plan


CASES = [
    (4, 12, 512, 512, False, 128),    # cfg2 itself: 96 + 96 workgroups
    (2, 3, 1024, 1024, False, 128),   # far-negative range, band, far-positive range in every row block
    (1, 2, 1024, 1024, False, 32),    # narrow band: pipelined far trips on both sides, general steps between
    (1, 2, 1000, 1100, False, 128),   # ragged: rows past M inside the last block (dO = 0 there), a key tail (general steps)
    (1, 2, 640, 512, False, 128),     # three row blocks against two key blocks: the partial rows are counted on the dQ side
    (1, 2, 512, 1280, False, 128),    # ... two against five
    (1, 2, 2048, 2048, True, 128),    # causal: the mask rides in the bias table (P = 0)
    (1, 2, 1000, 1100, True, 128),    # ... 0 < P < R, ragged
    (1, 2, 1100, 1000, True, 128),    # ... P < 0: dead rows
    (1, 1, 3000, 520, False, 64),     # many row blocks, remainder iterations
    (1, 2, 90, 70, False, 128),       # fewer steps than ring slots
    (1, 2, 768, 768, False, 512),     # radius 512: everything inside the band
    (1, 2, 700, 800, False, 512),     # ... ragged, operands straight from global (no room for the staging images): rows past M are copies of row M - 1 there
]


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("B,H,M,N,causal,md", CASES)
def test_qdiag_matches_oracle_and_the_dkdv_side_form(B, H, M, N, causal, md, dtype):
    from flasht5_amd import _lib
    q, k, v, do, table, bias = _rpe_case(B, H, M, N, dtype, causal, True, md, seed=M + 3 * N)
    ref = oracle_all(q, k, v, bias, do, 0.125, causal)
    outs = {}
    for name, bits in (("q", _lib.V_FUSED64_ON | _lib.V_QDIAG_ON), ("kv", _lib.V_FUSED64_ON | _lib.V_QDIAG_OFF)):
        plan = _plan(q, k, v, do, table, md, causal, bits)
        d = plan.describe()
        assert d["fused"] == "1" and d["qdiag"] == ("1" if name == "q" else "0"), d
        outs[name] = [t.clone() for t in plan.backward()] + [plan.o.clone()]
        torch.cuda.synchronize()
    dq, dk, dv, dt, o = outs["q"]
    for got, key in ((dq, "dq"), (dk, "dk"), (dv, "dv")):
        assert torch.isfinite(got.float()).all(), key
        assert maxdiff(got, ref[key]) <= gbound(ref[key], dtype), key
    for i in range(3):  # the diagonal sums touch none of dq / dk / dv: the same bits whichever side forms them
        assert torch.equal(outs["q"][i], outs["kv"][i]), ("dq", "dk", "dv")[i]
    want, allow = _table_truth(q, k, v, bias, o, ref["L"], do, 0.125, causal, table, M, N, True, md)
    assert torch.isfinite(dt).all()
    err = (dt.cpu() - want).abs()
    assert bool((err <= allow + 2e-3 * max(1.0, want.abs().max().item()) + 1e-2).all()), (err.max().item(), allow.max().item())
    # the dK/dV-side form sums the same dS -- unrounded in band / general steps, rounded to the input dtype in far trips; which tiles are far trips follows
    # the wave's 64 rows here and its 64 keys there, so the two differ by rounding noise of the far bins: inside twice the allowance
    diff = (dt - outs["kv"][3]).abs().cpu()
    assert bool((diff <= 2 * allow + 1e-3 * max(1.0, want.abs().max().item()) + 2e-3).all()), diff.max().item()


def test_qdiag_is_the_default_at_cfg2_and_deterministic():
    from flasht5_amd import _lib
    assert _lib.describe(B=4, H=12, M=512, N=512, bias_mode=_lib.BIAS_RPE1D, radius=128, need_dbias=True)["qdiag"] == "1"       # 192 workgroups: all resident
    assert _lib.describe(B=4, H=12, M=2048, N=2048, bias_mode=_lib.BIAS_RPE1D, radius=128, need_dbias=True)["qdiag"] == "0"     # 768: several rounds
    assert _lib.describe(B=4, H=12, M=512, N=512, bias_mode=_lib.BIAS_NONE)["qdiag"] == "0"
    q, k, v, do, table, bias = _rpe_case(4, 12, 512, 512, torch.bfloat16, False, True, 128, seed=7)
    a = _plan(q, k, v, do, table, 128, False, 0)
    assert a.describe()["qdiag"] == "1"
    ra = [t.clone() for t in a.backward()]
    b = _plan(q, k, v, do, table, 128, False, 0)
    rb = [t.clone() for t in b.backward()]
    torch.cuda.synchronize()
    for x, y in zip(ra, rb):
        assert torch.equal(x, y)


@pytest.mark.parametrize("B,H,M,N,causal", [(2, 3, 512, 768, False), (1, 2, 1000, 1100, True)])
def test_qdiag_stage_by_stage_and_unit_ranges(B, H, M, N, causal):
    """the layout decides the side: a call that runs the stages one by one launches the same two forms as separate kernels; unit ranges add up to the whole"""
    from flasht5_amd import _lib
    q, k, v, do, table, bias = _rpe_case(B, H, M, N, torch.bfloat16, causal, True, 128, seed=11)
    bits = _lib.V_FUSED64_ON | _lib.V_QDIAG_ON
    whole = _plan(q, k, v, do, table, 128, causal, bits)
    want = [t.clone() for t in whole.backward()]
    st = _plan(q, k, v, do, table, 128, causal, bits)
    st.backward(1)
    st.backward(2)
    st.backward(4)
    torch.cuda.synchronize()
    assert maxdiff(st.dq, want[0]) == 0.0  # (the same dQ body)
    for i, key in ((1, "dk"), (2, "dv")):  # (the stand-alone dK/dV kernel reads the dQ kernel's row statistics, the one-launch form makes its own: an output rounding apart)
        assert maxdiff(getattr(st, key), want[i]) <= 2.0 ** -6 * max(1.0, float(want[i].float().abs().max())), key
    assert maxdiff(st.dbias, want[3]) <= 1e-5 * max(1.0, float(want[3].abs().max()))
    # two unit ranges (head-major units u = h * B + b) -> partial tables that add up; dq / dk / dv of the units each range owns
    n = B * H
    cut = n // 2 + (1 if n > 2 else 0)
    parts = []
    for rng in ((0, cut), (cut, n - cut)):
        pl = _plan(q, k, v, do, table, 128, causal, bits, units=rng)
        pl.dq.zero_(); pl.dk.zero_(); pl.dv.zero_()
        pl.backward()
        parts.append(pl)
    torch.cuda.synchronize()
    assert maxdiff(parts[0].dbias + parts[1].dbias, want[3]) <= 1e-4 * max(1.0, float(want[3].abs().max()))
    assert maxdiff(parts[0].dq + parts[1].dq, want[0]) == 0.0
