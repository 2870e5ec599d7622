"""GPU parity tests of the attention hot path: HIP kernels (through the C ABI / ctypes / autograd.Function)
vs the CPU oracle, the committed golden fixtures, and the reference's own acceptance rule.

Tolerances
  * reference rule (tests/fa2_triton/test_fa2_bias.py:26-28,64-67): err <= 2 * err(eager low precision) + 1e-5, unmodified
  * elementwise, for o and dv: |got - ref| <= ELEM_C * (1e-3 + u * half-ulp) * max(1, rms of the entry's ROW) + u * half-ulp * |ref|
    -- an `atol + rtol * |ref|` check whose absolute part follows the row the entry sits in: unlike the global bound it does
    not let a row of small values borrow the tolerance of the tensor's largest entry (measured excess at ELEM_C = 1 over the
    fixtures and the reference's test shapes: o <= 1.12, dv <= 0.58; tools/calib_elem.py).  NOT applied to dq / dk / dbias: their
    error is not proportional to anything local.  dS = P * (dP - delta) is a difference of nearly equal numbers on rows with one
    dominant key, and delta comes from the STORED, rounded o (FA2's backward, the reference kernels alike, :516-556): an entry of
    dq can be wrong by 10x the tolerance of its own row's rms while well inside the reference's rule (measured: up to 13.3 at
    ELEM_C = 1 on test_fa2_bias.py's shapes).  For those tensors the reference rule and the global bound are the yardsticks.
  * lse (fp32 output): 1e-4 * max(1, max|L|)
  * fixed bound: err <= (1e-3 + u * half-ulp(dtype)) * max(1, max|ref|) -- 1e-3 is the north-star atol on the
    arithmetic; the half-ulp term (2^-8 bf16, 2^-11 fp16 of max|ref|) is the unavoidable rounding of the OUTPUT
    tensor (u = 1 forward; u = 3 gradients: their MFMA operands P and dS are rounded once more and delta is formed from
    the stored, rounded o -- all exactly as in the reference kernels).
"""
import math

import pytest
import torch

import oracle
from attn_helpers import make_inputs, oracle_all, run_dense, errors, maxdiff, eager_lowprec_errors
from golden_io import load_attn, ATTN_CASES, TRITON_CASES

pytestmark = pytest.mark.gpu

HALF_ULP = {torch.bfloat16: 2.0 ** -8, torch.float16: 2.0 ** -11}  # worst case (value just above a power of two)


def bound(ref_t, dtype, atol=1e-3, ulps=1.0):
    return (atol + ulps * HALF_ULP[dtype]) * max(1.0, ref_t.float().abs().max().item())


def gbound(ref_t, dtype):
    """gradients: P and dS enter the dV/dK/dQ contractions rounded to the input dtype (exactly like the
    reference kernels, flash_attention_v2_bias.py:702,:720-722) -> one more rounding than the forward."""
    return bound(ref_t, dtype, ulps=3.0)


ELEM_C = 2.0


def elem_excess(got, ref_t, dtype, ulps=1.0, nsum=1):
    """max over elements of |got - ref| / (atol + rtol * |ref|) (<= 1 passes); see the module docstring"""
    ref_f, got_f = ref_t.float(), got.float()
    rms = torch.nan_to_num(ref_f, nan=0.0, posinf=0.0, neginf=0.0).square().mean(dim=-1, keepdim=True).sqrt().clamp(min=1.0)
    lim = ELEM_C * (1e-3 + ulps * HALF_ULP[dtype]) * rms * nsum + ulps * HALF_ULP[dtype] * ref_f.abs() * nsum
    d = (got_f - ref_f).abs()
    d = torch.where(got_f == ref_f, torch.zeros_like(d), d)
    d = torch.where(torch.isnan(d), torch.full_like(d, float("inf")), d)
    return (d / lim).max().item()


def lse_bound(L_ref):
    fin = torch.isfinite(L_ref)
    return 1e-4 * max(1.0, L_ref[fin].abs().max().item() if fin.any() else 0.0)


def to_dev(c):
    return {k: (v.cuda() if isinstance(v, torch.Tensor) else v) for k, v in c.items()}


@pytest.mark.parametrize("name", ATTN_CASES)
def test_golden_fixture(name):
    c = to_dev(load_attn(name))
    got = run_dense(c["q"], c["k"], c["v"], c["bias"], c["do"], c["sm_scale"], c["causal"])
    dt = c["dtype"]
    lp = c["eager_lp_err"].tolist()  # [o, dq, dk, dv, dbias] of the reference's eager low-precision path
    for i, key in enumerate(("o", "dq", "dk", "dv")):
        e = maxdiff(got[key], c[key])
        assert e <= (bound(c[key], dt) if key == "o" else gbound(c[key], dt)), (key, e)
        if key in ("o", "dv"):
            assert elem_excess(got[key], c[key], dt, 1.0 if key == "o" else 3.0) <= 1.0, (key, "elementwise")
        if dt != torch.float32:
            assert lp[i] > 0, (name, key, "fixture carries no eager low-precision error")
            assert e <= 2 * lp[i] + 1e-5, (key, e, lp[i])  # the reference's rule as its tests state it
    # lse is part of the operator contract (reference :59, :476: natural log, fp32); -inf on rows without a visible key
    assert maxdiff(got["L"], c["L"]) <= lse_bound(c["L"]), ("L", maxdiff(got["L"], c["L"]))
    if "o_ref" in c:  # the REFERENCE's own eager fp32 output (the `o` above is the oracle's, asserted < 2e-5 from it)
        assert maxdiff(got["o"], c["o_ref"]) <= bound(c["o_ref"], dt) + 2e-5
    if c["bias"] is not None:
        e = maxdiff(got["db"], c["dbias"])
        # dS is rounded to the bias dtype BEFORE the batch/head sum, like the reference (:720,:214): one extra
        # half-ulp per summed term
        nsum = (c["B"] if c["bias"].shape[0] == 1 else 1) * (c["H"] if c["bias"].shape[1] == 1 else 1)
        assert e <= gbound(c["dbias"], dt) * (1 + nsum), ("db", e)


def test_cfg1_fwd_numerics():
    """config 1: t5-small encoder self-attn fwd (2,8,128,64); fp32 eager reference vs the bf16 kernel."""
    c = to_dev(load_attn("attn_cfg1_fp32"))
    from flasht5_amd import flash_attention_v2_bias
    o = flash_attention_v2_bias(c["q"].bfloat16(), c["k"].bfloat16(), c["v"].bfloat16(), c["bias"].bfloat16(),
                                False, c["sm_scale"])
    assert maxdiff(o, c["o"]) <= bound(c["o"], torch.bfloat16)


@pytest.mark.parametrize("name", TRITON_CASES)
def test_vs_reference_triton_kernels(name):
    """Same inputs as the reference's Triton kernels (run under the interpreter, fp16): outputs agree within
    the two kernels' combined rounding (each is within ~half an output ulp + 1e-3 of fp32 truth)."""
    c = to_dev(load_attn(name))
    got = run_dense(c["q"], c["k"], c["v"], c["bias"], c["do"], c["sm_scale"], c["causal"])
    for key, tk in (("o", "o_triton"), ("dq", "dq_triton"), ("dk", "dk_triton"), ("dv", "dv_triton"), ("db", "dbias_triton")):
        e = maxdiff(got[key], c[tk])
        scale = max(1.0, c[tk].float().abs().max().item())
        nsum = c["B"] if (key == "db" and c["bias"].shape[0] == 1) else 1
        assert e <= 2 * (1e-3 + HALF_ULP[torch.float16]) * scale * nsum, (key, e)
    # the reference kernel's own log-sum-exp (`L`, fp32: no output rounding on either side)
    assert maxdiff(got["L"], c["L_triton"]) <= lse_bound(c["L_triton"]), ("L", maxdiff(got["L"], c["L_triton"]))


@pytest.mark.parametrize("B,H,M,N,D", [(2, 4, 512, 612, 128), (2, 4, 1024, 1045, 64)])
@pytest.mark.parametrize("causal", [True, False])
@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
def test_reference_shapes_fwd_bwd(B, H, M, N, D, causal, dtype):
    """The reference's own test shapes (test_fa2_bias.py:10-13): M != N, N not divisible by any tile."""
    q, k, v, b, do = make_inputs(B, H, M, N, D, dtype, "bh", seed=3)
    ref = oracle_all(q, k, v, b, do, 1.0, causal)
    lp = eager_lowprec_errors(q, k, v, b, do, 1.0, causal, ref)
    got = run_dense(q, k, v, b, do, 1.0, causal)
    for key in ("o", "dq", "dk", "dv", "db"):
        e = maxdiff(got[key], ref[key])
        assert e <= 2 * lp[key] + 1e-5, (key, e, lp[key])  # the reference's rule, unmodified (test_fa2_bias.py:26-28)
        assert e <= (bound(ref[key], dtype) if key == "o" else gbound(ref[key], dtype)), (key, e)
        if key in ("o", "dv"):
            assert elem_excess(got[key], ref[key], dtype, 1.0 if key == "o" else 3.0) <= 1.0, (key, "elementwise")
    assert maxdiff(got["L"], ref["L"]) <= lse_bound(ref["L"]), "L"


@pytest.mark.parametrize("kind", ["11", "b1", "1h"])
def test_broadcast_bias_gradient(kind):
    """(1,1,M,N) / (B,1,M,N) / (1,H,M,N): dbias is the mathematically correct sum (the reference races on
    head-broadcast biases -- SURVEY Q4)."""
    q, k, v, b, do = make_inputs(2, 3, 100, 77, 64, torch.bfloat16, kind, seed=11)
    ref = oracle_all(q, k, v, b, do, 1.0, False)
    got = run_dense(q, k, v, b, do, 1.0, False)
    assert got["db"].shape == b.shape and got["db"].dtype == b.dtype
    nsum = (2 if b.shape[0] == 1 else 1) * (3 if b.shape[1] == 1 else 1)
    assert maxdiff(got["db"], ref["db"]) <= gbound(ref["db"], torch.bfloat16) * (1 + nsum)


def test_strided_inputs_and_layout():
    """(B,S,H,D)-backed permuted views: outputs keep the input layout."""
    q, k, v, b, do = make_inputs(2, 4, 200, 264, 64, torch.bfloat16, "1h", seed=5, strided=True)
    from flasht5_amd import flash_attention_v2_bias
    o = flash_attention_v2_bias(q, k, v, b, False, 0.125)
    assert o.stride() == q.stride()
    ref = oracle_all(q, k, v, b, do, 0.125, False)
    got = run_dense(q, k, v, b, do, 0.125, False)
    for key in ("o", "dq", "dk", "dv", "db"):
        assert maxdiff(got[key], ref[key]) <= (bound if key == "o" else gbound)(ref[key], torch.bfloat16) * (3 if key == "db" else 1), key


def test_causal_m_larger_than_n_empty_rows():
    """causal with M > N: the first M-N rows see no key -> o = 0, finite gradients (reference :470-473)."""
    q, k, v, b, do = make_inputs(1, 2, 96, 64, 64, torch.bfloat16, "1h", seed=9)
    from flasht5_amd.flash_attention_v2_bias import _attn_fwd
    o, L = _attn_fwd(q, k, v, b, None, 0, True, 1.0)
    assert torch.all(o[:, :, :32] == 0) and torch.all(torch.isinf(L[:, :, :32])) and torch.all(L[:, :, :32] < 0)
    got = run_dense(q, k, v, b, do, 1.0, True)
    ref = oracle_all(q, k, v, b, do, 1.0, True)
    for key in ("o", "dq", "dk", "dv", "db"):
        assert torch.isfinite(got[key].float()).all(), key
        assert maxdiff(got[key], ref[key]) <= (bound if key == "o" else gbound)(ref[key], torch.bfloat16) * (3 if key == "db" else 1), key


@pytest.mark.parametrize("D", [16, 32, 128])
def test_head_dims(D):
    q, k, v, b, do = make_inputs(1, 2, 72, 100, D, torch.float16, "1h", seed=D)
    ref = oracle_all(q, k, v, b, do, 1.0 / math.sqrt(D), True)
    got = run_dense(q, k, v, b, do, 1.0 / math.sqrt(D), True)
    for key in ("o", "dq", "dk", "dv", "db"):
        assert maxdiff(got[key], ref[key]) <= (bound if key == "o" else gbound)(ref[key], torch.float16), key


def test_default_scale_and_none_bias():
    q, k, v, _, do = make_inputs(2, 2, 128, 128, 64, torch.bfloat16, None, seed=2)
    got = run_dense(q, k, v, None, do, None, False)
    ref = oracle_all(q, k, v, None, do, 1.0 / 8.0, False)
    for key in ("o", "dq", "dk", "dv"):
        assert maxdiff(got[key], ref[key]) <= (bound if key == "o" else gbound)(ref[key], torch.bfloat16), key


@pytest.mark.parametrize("boost,rows", [(0.0, "all"), (40.0, "all"), (400.0, "all"), (1000.0, "even")])
def test_fwd_optimistic_softmax_edge_cases(boost, rows):
    """bf16 all-visible tiles run without a running row maximum once a row has a baseline (attn_fwd.h, "optimistic").
    Scores that rise by `boost` nats after the baseline tiles exercise: nothing (0), the power-of-two renormalisation
    of O / l (40 nats ~ 2^58), and the overflow -> exact second pass of the workgroup (400 / 1000 nats; "even" = only
    every other row overflows).  All must match the oracle like any other input."""
    B, H, S, D = 1, 2, 1024, 64
    g = torch.Generator().manual_seed(11)
    q = torch.randn(B, H, S, D, generator=g).bfloat16()
    k = torch.randn(B, H, S, D, generator=g).bfloat16()
    v = torch.randn(B, H, S, D, generator=g).bfloat16()
    q[..., 0] = 4.0
    if rows == "even":
        q[..., 1::2, 0] = 0.0
    k[..., 256:, 0] = boost / 4.0
    q, k, v = q.cuda(), k.cuda(), v.cuda()
    do = torch.randn(B, H, S, D, generator=g).bfloat16().cuda()
    got = run_dense(q, k, v, None, do, 1.0, False)
    ref = oracle_all(q, k, v, None, do, 1.0, False)
    assert torch.isfinite(got["o"].float()).all()
    assert maxdiff(got["o"], ref["o"]) <= bound(ref["o"], torch.bfloat16)
    # the backward consumes the L written by either pass.  Every key carries the same large component, so dq/dk are
    # sums that cancel (rows of the softmax Jacobian sum to zero): the yardstick is the reference tests' own rule,
    # a small multiple of the eager bf16 error (test_fa2_bias.py:64-67).
    lp = eager_lowprec_errors(q, k, v, None, do, 1.0, False, ref)
    for key in ("dq", "dk", "dv"):
        e = maxdiff(got[key], ref[key])
        assert torch.isfinite(got[key].float()).all(), key
        assert e <= max(gbound(ref[key], torch.bfloat16), 3 * lp[key]), (key, e, lp[key])


def test_deterministic():
    """two runs are bit-identical (no atomics on dQ/dK/dV/dense dBias) -- would have caught the reference's Q4 race."""
    q, k, v, b, do = make_inputs(2, 3, 256, 300, 64, torch.bfloat16, "11", seed=1)
    a = run_dense(q, k, v, b, do, 1.0, True)
    c = run_dense(q, k, v, b, do, 1.0, True)
    for key in a:
        assert torch.equal(a[key], c[key]), key


# ---- linear-memory RPE mode ----------------------------------------------------------------------------------
def _rpe_case(B, H, M, N, dtype, causal, bidir, max_distance=128, seed=0):
    q, k, v, _, do = make_inputs(B, H, M, N, 64, dtype, None, seed=seed)
    g = torch.Generator().manual_seed(seed + 100)
    table = torch.randn(32, H, generator=g) * 0.5
    bias = oracle.compute_bias(table, M, N, bidir, 32, max_distance).contiguous().cuda()
    return q, k, v, do, table, bias


@pytest.mark.parametrize("B,H,M,N,causal,bidir,md", [
    (2, 2, 256, 256, False, True, 128), (2, 2, 96, 160, False, True, 128), (2, 2, 128, 128, True, False, 128),
    (1, 2, 512, 512, False, True, 128), (2, 2, 300, 200, False, True, 64), (1, 2, 520, 333, True, True, 32)])
def test_rpe_mode_matches_dense_oracle(B, H, M, N, causal, bidir, md):
    from flasht5_amd import flash_attention_v2_rpe
    dtype = torch.bfloat16
    q, k, v, do, table, bias = _rpe_case(B, H, M, N, dtype, causal, bidir, md, seed=M + N)
    ref = oracle_all(q, k, v, bias, do, 1.0, causal)  # fp32 bias: the RPE mode never rounds it
    leaves = [t.detach().clone().requires_grad_() for t in (q, k, v)]
    tb = table.cuda().requires_grad_()
    o = flash_attention_v2_rpe(leaves[0], leaves[1], leaves[2], tb, bidir, 32, md, causal, 1.0)
    dq, dk, dv, dt = torch.autograd.grad(o, leaves + [tb], do)
    # table-gradient truth: the oracle's dS with delta = rowsum(o * do) formed from the STORED (rounded) o, which is
    # what FA2 backward defines (reference _bwd_preprocess reads the bf16 o, :516-556); with the unrounded o the
    # per-row offsets of ~2^-9 |o||do| sqrt(D) pile up over the B*M rows of a bucket.
    _, _, _, _, db_alg = oracle.attn_bwd_oracle(q, k, v, bias, o.detach(), ref["L"], do, 1.0, causal)
    tl = table.clone().requires_grad_()
    oracle.compute_bias(tl, M, N, bidir, 32, md).backward(db_alg.cpu())
    for got, key in ((o, "o"), (dq, "dq"), (dk, "dk"), (dv, "dv")):
        assert maxdiff(got, ref[key]) <= (bound if key == "o" else gbound)(ref[key], dtype), key
    assert maxdiff(dt.cpu(), tl.grad) <= 5e-3 * max(1.0, tl.grad.abs().max().item()) + 2e-2



def test_shared_rpe1d_across_layers():
    """one `(H, 2R+1)` generator built once from the T5 table feeds several layers; autograd sums the
    per-layer diagonal gradients and scatters them into the table once.  Truth: the dense oracle per layer."""
    from flasht5_amd import flash_attention_v2_rpe1d
    from flasht5_amd.positional_encoding import RelativePositionalEncoding
    B, H, S, D, layers = 2, 2, 256, 64, 3
    torch.manual_seed(21)
    mod = RelativePositionalEncoding(32, 128, H).cuda()
    table = mod.relative_attention_bias.weight.detach().clone()
    bias = oracle.compute_bias(table.cpu(), S, S, True, 32, 128).contiguous().cuda()
    r1, R = mod.forward_1d()
    total, outs, db_sum = 0.0, [], torch.zeros(1, H, S, S)
    for layer in range(layers):
        q, k, v, _, do = make_inputs(B, H, S, S, D, torch.bfloat16, None, seed=50 + layer)
        o = flash_attention_v2_rpe1d(q, k, v, r1, R, False, 1.0)
        total = total + (o.float() * do.float()).sum()
        ref = oracle_all(q, k, v, bias, do, 1.0, False)
        assert maxdiff(o, ref["o"]) <= bound(ref["o"], torch.bfloat16)
        _, _, _, _, db = oracle.attn_bwd_oracle(q, k, v, bias, o.detach(), ref["L"], do, 1.0, False)
        db_sum += db.cpu()
    total.backward()
    tl = table.cpu().clone().requires_grad_()
    oracle.compute_bias(tl, S, S, True, 32, 128).backward(db_sum)
    got = mod.relative_attention_bias.weight.grad.cpu()
    assert maxdiff(got, tl.grad) <= 5e-3 * max(1.0, tl.grad.abs().max().item()) + 2e-2 * layers


def test_rpe_equals_dense_kernel_full_cfg2():
    """config 2 at full size: the RPE-mode kernels and the dense-bias kernels agree (same table)."""
    from flasht5_amd import flash_attention_v2_rpe, flash_attention_v2_bias, compute_bias
    q, k, v, _, do = make_inputs(4, 12, 512, 512, 64, torch.bfloat16, None, seed=42, strided=True)
    table = (torch.randn(32, 12, generator=torch.Generator().manual_seed(1)) * 0.5).cuda()
    bias = compute_bias(table, 512, 512).contiguous()  # fp32 values
    o1 = flash_attention_v2_rpe(q, k, v, table, True, 32, 128, False, 0.125)
    o2 = flash_attention_v2_bias(q, k, v, bias.bfloat16(), False, 0.125)
    # the dense path rounds the bias to bf16 (like the reference, positional_encoding.py:108)
    assert maxdiff(o1, o2) <= 3e-2
    ref, _ = oracle.attn_fwd_oracle(q, k, v, bias, 0.125, False)
    assert maxdiff(o1, ref) <= bound(ref, torch.bfloat16)


# ---- full-size, size-independent properties (config 3: S = 8192) ---------------------------------------------
def test_cfg3_properties_s8192():
    from flasht5_amd.flash_attention_v2_bias import AttentionPlan
    from flasht5_amd import positional_encoding as pe
    B, H, S, D = 4, 12, 8192, 64
    q, k, v, _, do = make_inputs(B, H, S, S, D, torch.bfloat16, None, seed=8, strided=True)
    table = (torch.randn(32, H, generator=torch.Generator().manual_seed(2)) * 0.5).cuda()
    rpe1d = pe.rpe1d_from_table(table)
    plan = AttentionPlan(q, k, v, do, rpe1d=rpe1d, radius=128, sm_scale=0.125)
    o = plan.forward().clone()
    dq, dk, dv, d1 = (t.clone() for t in plan.backward())
    torch.cuda.synchronize()
    for t in (o, dq, dk, dv, d1):
        assert torch.isfinite(t.float()).all()
    # (1) a (b,h) slice against the fp32 oracle run on the device (full S x S scores for 2 heads)
    bias = pe.compute_bias(table[:, :2], S, S).contiguous()
    sl = slice(0, 1), slice(0, 2)
    qs, ks, vs, dos = (t[sl[0], sl[1]] for t in (q, k, v, do))
    ref_o, ref_L = oracle.attn_fwd_oracle(qs, ks, vs, bias, 0.125, False)
    assert maxdiff(o[sl[0], sl[1]], ref_o) <= bound(ref_o, torch.bfloat16)
    assert maxdiff(plan.lse[sl[0], sl[1]], ref_L) <= 1e-3
    rdq, rdk, rdv, rds, _ = oracle.attn_bwd_oracle(qs, ks, vs, bias, ref_o, ref_L, dos, 0.125, False)
    assert maxdiff(dq[sl[0], sl[1]], rdq) <= gbound(rdq, torch.bfloat16)
    assert maxdiff(dk[sl[0], sl[1]], rdk) <= gbound(rdk, torch.bfloat16)
    assert maxdiff(dv[sl[0], sl[1]], rdv) <= gbound(rdv, torch.bfloat16)
    # (2) softmax Jacobian: every row of dS sums to zero => the diagonal sums of each head sum to ~0
    # (the far bins sum the dS that were rounded to bf16 for the dK GEMM -- the reference's bias gradient is made of the same
    #  rounded values, :720 -- so the total carries n independent roundings: 4 sigma = 4 * 2^-9 / sqrt(3) * sqrt(sum ds^2), with
    #  sum ds^2 of a head estimated from the oracle's dS of the two checked heads of batch 0, times B)
    allow = max(4.0 * 2.0 ** -9 / 3 ** 0.5 * (B * (rds[0, hh].float() ** 2).sum().item()) ** 0.5 for hh in range(2))
    tot = d1.sum(-1).abs()
    assert tot[:2].max().item() <= allow + 1e-3 * d1.abs().sum(-1).max().item() + 1e-2, (tot, allow)
    assert tot.max().item() <= 3 * allow + 1e-3 * d1.abs().sum(-1).max().item() + 1e-2, (tot, allow)
    # (3) linearity in V and dO: o(2v) = 2 o(v) exactly in bf16 (power of two); dq(2 do) = 2 dq
    plan2 = AttentionPlan(q, k, (v * 2).contiguous(), (do * 2).contiguous(), rpe1d=rpe1d, radius=128, sm_scale=0.125)
    o2 = plan2.forward()
    assert torch.equal(o2, o * 2)
    # (4) key permutation invariance without bias: shuffling (k, v) rows jointly leaves o unchanged up to rounding
    plan3 = AttentionPlan(q[:1], k[:1], v[:1], do[:1], sm_scale=0.125)
    o3 = plan3.forward().clone()
    perm = torch.randperm(S, device="cuda")
    plan4 = AttentionPlan(q[:1], k[:1][:, :, perm].contiguous(), v[:1][:, :, perm].contiguous(), do[:1], sm_scale=0.125)
    o4 = plan4.forward()
    assert maxdiff(o3, o4) <= 2 * HALF_ULP[torch.bfloat16] * max(1.0, o3.float().abs().max().item()) + 1e-3


def test_varlen_cross_attention_cfg4():
    """config 4: packed decoder cross-attention, q_len 256 / kv_len 4096 class, via cu_seqlens."""
    from flasht5_amd import flash_attn_varlen_fwd
    H, D = 12, 64
    cu_q = [0, 256, 512, 704, 768]
    cu_k = [0, 4096, 7168, 11264, 12288]
    g = torch.Generator().manual_seed(4)
    q = torch.randn(cu_q[-1], H, D, generator=g).bfloat16().cuda()
    k = torch.randn(cu_k[-1], H, D, generator=g).bfloat16().cuda()
    v = torch.randn(cu_k[-1], H, D, generator=g).bfloat16().cuda()
    o, lse = flash_attn_varlen_fwd(q, k, v, torch.tensor(cu_q, dtype=torch.int32).cuda(),
                                   torch.tensor(cu_k, dtype=torch.int32).cuda(), 256, 4096, False, 0.125)
    ref = oracle.attn_varlen_oracle(q, k, v, cu_q, cu_k, 0.125).to(q.device)
    assert maxdiff(o, ref) <= bound(ref, torch.bfloat16)
    # ragged edge cases: an empty query sequence, an empty key sequence, length-1 sequences
    cu_q2, cu_k2 = [0, 5, 5, 6, 70], [0, 9, 12, 12, 141]
    q2, k2, v2 = q[:70].contiguous(), k[:141].contiguous(), v[:141].contiguous()
    o2, _ = flash_attn_varlen_fwd(q2, k2, v2, torch.tensor(cu_q2, dtype=torch.int32).cuda(),
                                  torch.tensor(cu_k2, dtype=torch.int32).cuda(), 64, 129, False, 0.125)
    ref2 = oracle.attn_varlen_oracle(q2, k2, v2, cu_q2, cu_k2, 0.125).to(q.device)
    assert maxdiff(o2, ref2) <= bound(ref2, torch.bfloat16)



@pytest.mark.parametrize("causal", [False, True])
def test_varlen_backward(causal):
    """packed batches are differentiable -- dq/dk/dv of the cu_seqlens path vs the per-sequence oracle,
    at config-4 class sizes and on ragged edge cases (empty query / key sequences, length-1 sequences)."""
    from flasht5_amd import flash_attn_varlen_func
    H = 12
    g = torch.Generator().manual_seed(14)
    for cu_q, cu_k, mq, mk, D in (([0, 256, 512, 704, 768], [0, 1024, 1792, 2816, 3072], 256, 1024, 64),
                                  ([0, 5, 5, 6, 70, 71], [0, 9, 12, 12, 141, 142], 64, 129, 64),
                                  ([0, 40, 41, 130], [0, 33, 97, 200], 89, 103, 16)):   # head_dim 16: native (no padded copies)
        q = torch.randn(cu_q[-1], H, D, generator=g).bfloat16().cuda()
        k = torch.randn(cu_k[-1], H, D, generator=g).bfloat16().cuda()
        v = torch.randn(cu_k[-1], H, D, generator=g).bfloat16().cuda()
        do = torch.randn(cu_q[-1], H, D, generator=g).bfloat16().cuda()
        leaves = [t.clone().requires_grad_() for t in (q, k, v)]
        o = flash_attn_varlen_func(leaves[0], leaves[1], leaves[2], torch.tensor(cu_q, dtype=torch.int32).cuda(),
                                   torch.tensor(cu_k, dtype=torch.int32).cuda(), mq, mk, causal, 0.125)
        dq, dk, dv = torch.autograd.grad(o, leaves, do)
        ref_o = oracle.attn_varlen_oracle(q.cpu(), k.cpu(), v.cpu(), cu_q, cu_k, 0.125, causal)
        rdq, rdk, rdv = oracle.attn_varlen_bwd_oracle(q.cpu(), k.cpu(), v.cpu(), do.cpu(), cu_q, cu_k, 0.125, causal)
        assert maxdiff(o.cpu(), ref_o) <= bound(ref_o, torch.bfloat16)
        for got, ref, key in ((dq, rdq, "dq"), (dk, rdk, "dk"), (dv, rdv, "dv")):
            assert torch.isfinite(got.float()).all(), key
            assert maxdiff(got.cpu(), ref) <= gbound(ref, torch.bfloat16), (key, cu_q)


def test_bad_arguments_raise():
    from flasht5_amd import flash_attention_v2_bias
    q, k, v, b, _ = make_inputs(1, 1, 32, 32, 64, torch.bfloat16, "1h")
    with pytest.raises(AssertionError):
        flash_attention_v2_bias(q[..., :48], k[..., :48], v[..., :48], None)  # head_dim 48 (reference asserts too)
    with pytest.raises(TypeError):
        flash_attention_v2_bias(q.float(), k.float(), v.float(), None)
    with pytest.raises(RuntimeError):
        flash_attention_v2_bias(q.cpu(), k.cpu(), v.cpu(), None)  # no CPU fallback


# ---- seeded shape fuzz: every bias mode / dtype / mask / tail combination the tile classifiers can see ------------
def _fuzz_cases(seed=20260928, count=36, dims=(32, 64, 64, 64, 128)):
    import random
    rnd = random.Random(seed)
    cases = []
    for i in range(count):
        D = rnd.choice(list(dims))
        M = rnd.choice([1, 17, 33, 64, 95, 128, 200, 257, 320, 449, 512, 700])
        N = rnd.choice([1, 19, 32, 64, 100, 128, 191, 256, 333, 448, 512, 640])
        cases.append((i, rnd.choice([1, 2, 3]), rnd.choice([1, 2, 5]), M, N, D, rnd.choice([False, True]),
                      rnd.choice(["none", "dense_bh", "dense_1h", "dense_11", "rpe", "rpe", "rpe_uni"]),
                      rnd.choice([torch.bfloat16, torch.bfloat16, torch.float16]), rnd.choice([16, 32, 64, 128]),
                      rnd.choice([1.0, 0.125, 0.37])))
    return cases


@pytest.mark.parametrize("case", _fuzz_cases(), ids=lambda c: f"{c[0]}-B{c[1]}H{c[2]}M{c[3]}N{c[4]}D{c[5]}{'c' if c[6] else ''}-{c[7]}")
def test_fuzz_shapes_modes(case):
    _run_fuzz_case(case, strided=False)


@pytest.mark.parametrize("case", _fuzz_cases(seed=1616, count=20, dims=(16,)), ids=lambda c: f"{c[0]}-B{c[1]}H{c[2]}M{c[3]}N{c[4]}D{c[5]}{'c' if c[6] else ''}-{c[7]}")
def test_head_dim_16_native(case):
    """head_dim 16 (the reference's smallest, flash_attention_v2_bias.py:233-234) through the C ABI as it is -- no padded copies: the D = 32
    bodies read columns 16..31 as zeros and never write them.  (B, S, H, D)-strided inputs: the 16 columns behind a row belong to the NEXT head,
    so a kernel that read or wrote them would show up in o / dq / dk / dv of the neighbour."""
    _run_fuzz_case(case, strided=True)


def _run_fuzz_case(case, strided):
    from flasht5_amd import flash_attention_v2_rpe
    _, B, H, M, N, D, causal, mode, dtype, md, scale = case
    if mode == "rpe_uni" and md <= 16:
        md = 64  # unidirectional: max_exact = 16, max_distance must exceed it (the reference formula divides by log(md/16))
    if mode.startswith("rpe"):
        q, k, v, _, do = make_inputs(B, H, M, N, D, dtype, None, seed=case[0], strided=strided)
        g = torch.Generator().manual_seed(case[0] + 500)
        table = torch.randn(32, H, generator=g) * 0.5
        bidir = mode == "rpe"
        bias = oracle.compute_bias(table, M, N, bidir, 32, md).contiguous().cuda()
        ref = oracle_all(q, k, v, bias, do, scale, causal)
        leaves = [t.detach().clone().requires_grad_() for t in (q, k, v)]
        tb = table.cuda().requires_grad_()
        o = flash_attention_v2_rpe(leaves[0], leaves[1], leaves[2], tb, bidir, 32, md, causal, scale)
        dq, dk, dv, dt = torch.autograd.grad(o, leaves + [tb], do)
        got = {"o": o.detach(), "dq": dq, "dk": dk, "dv": dv}
        _, _, _, _, db_alg = oracle.attn_bwd_oracle(q, k, v, bias, o.detach(), ref["L"], do, scale, causal)
        tl = table.clone().requires_grad_()
        oracle.compute_bias(tl, M, N, bidir, 32, md).backward(db_alg.cpu())
        assert torch.isfinite(dt).all()
        assert maxdiff(dt.cpu(), tl.grad) <= 1e-2 * max(1.0, tl.grad.abs().max().item()) + 3e-2
    else:
        kind = {"none": None, "dense_bh": "bh", "dense_1h": "1h", "dense_11": "11"}[mode]
        q, k, v, b, do = make_inputs(B, H, M, N, D, dtype, kind, seed=case[0], strided=strided)
        ref = oracle_all(q, k, v, b, do, scale, causal)
        got = run_dense(q, k, v, b, do, scale, causal)
        if b is not None:
            assert maxdiff(got["db"], ref["db"]) <= gbound(ref["db"], dtype) * (2.0 if kind != "bh" else 1.0)
    for key in ("o", "dq", "dk", "dv"):
        assert torch.isfinite(got[key].float()).all(), key
        assert maxdiff(got[key], ref[key]) <= (bound if key == "o" else gbound)(ref[key], dtype), key


@pytest.mark.parametrize("M,N,causal,md,bidir,dtype", [
    (1536, 1536, False, 32, True, torch.bfloat16), (1100, 1300, False, 64, True, torch.bfloat16),
    (1300, 1100, True, 32, True, torch.bfloat16), (1024, 1600, True, 128, False, torch.float16),
    (1601, 999, False, 16, True, torch.float16), (2048, 2048, True, 64, True, torch.bfloat16)])
def test_rpe_long_rows_far_and_near_tiles(M, N, causal, md, bidir, dtype):
    """Sequences several times the RPE radius: far-constant FAST tiles on both sides of the band, generic tiles on the band
    and at the causal / tail edges, and the carried diagonal sums across many consecutive near blocks."""
    from flasht5_amd import flash_attention_v2_rpe
    B, H = 1, 2
    q, k, v, do, table, bias = _rpe_case(B, H, M, N, dtype, causal, bidir, md, seed=M + 7 * N)
    ref = oracle_all(q, k, v, bias, do, 0.125, causal)
    leaves = [t.detach().clone().requires_grad_() for t in (q, k, v)]
    tb = table.cuda().requires_grad_()
    o = flash_attention_v2_rpe(leaves[0], leaves[1], leaves[2], tb, bidir, 32, md, causal, 0.125)
    dq, dk, dv, dt = torch.autograd.grad(o, leaves + [tb], do)
    _, _, _, _, db_alg = oracle.attn_bwd_oracle(q, k, v, bias, o.detach(), ref["L"], do, 0.125, causal)
    tl = table.clone().requires_grad_()
    oracle.compute_bias(tl, M, N, bidir, 32, md).backward(db_alg.cpu())
    for got, key in ((o, "o"), (dq, "dq"), (dk, "dk"), (dv, "dv")):
        assert maxdiff(got, ref[key]) <= (bound if key == "o" else gbound)(ref[key], dtype), key
    assert maxdiff(dt.cpu(), tl.grad) <= 1e-2 * max(1.0, tl.grad.abs().max().item()) + 3e-2


@pytest.mark.parametrize("scale", [-0.5, 0.0, 1e-3])
@pytest.mark.parametrize("kind", [None, "1h"])
def test_negative_zero_and_tiny_sm_scale(scale, kind):
    """sm_scale is the caller's (reference :239-240 only fills the default): a negative scale disables the folded fast
    tiles (max(s*c) = c*max(s) needs c > 0) and flips the sign of the staged -L/scale; an exact zero makes the softmax
    depend on the bias alone, dq = dk = 0 (the backward substitutes 1e-30: |dq|, |dk| <= 1e-28)."""
    q, k, v, b, do = make_inputs(2, 2, 200, 264, 64, torch.bfloat16, kind, seed=5)
    ref = oracle_all(q, k, v, b, do, scale, True)
    got = run_dense(q, k, v, b, do, scale, True)
    for key in got:
        assert torch.isfinite(got[key].float()).all(), key
        assert maxdiff(got[key], ref[key]) <= (bound if key == "o" else gbound)(ref[key], torch.bfloat16), key


def test_torch_compile_aot_eager_traces_the_ops():
    """The reference registers its kernels as custom ops with fake impls so `torch.compile` can trace through them
    (flash_attention_v2_bias.py:27,:83-89); same here -- dynamo + AOT autograd (backend "aot_eager": no Triton needed)
    captures forward and backward and reproduces the eager result bit for bit."""
    from flasht5_amd import flash_attention_v2_bias
    q, k, v, b, do = make_inputs(2, 2, 128, 160, 64, torch.bfloat16, "1h", seed=9)

    def f(q, k, v, b):
        return flash_attention_v2_bias(q, k, v, b, True, 0.125)

    eager = [t.detach().clone().requires_grad_() for t in (q, k, v, b)]
    oe = f(*eager)
    ge = torch.autograd.grad(oe, eager, do)
    comp = [t.detach().clone().requires_grad_() for t in (q, k, v, b)]
    oc = torch.compile(f, backend="aot_eager", fullgraph=True)(*comp)
    gc = torch.autograd.grad(oc, comp, do)
    assert torch.equal(oe, oc)
    for a, c in zip(ge, gc):
        assert torch.equal(a, c)


@pytest.mark.parametrize("attention_type", ["triton", "fat5_rpe"])
@pytest.mark.parametrize("decoder", [False, True])
def test_flasht5_attention_module_two_blocks(attention_type, decoder):
    """The `FlashT5Attention`-compatible module (reference modeling_flash_t5.py:166-287): block 0 builds the position
    bias and hands it to block 1 (dense `(1,H,M,N)` tensor, or the `(H,2R+1)` generator in the linear-memory type);
    outputs and the gradients of every parameter -- including the shared T5 table through both blocks -- against the
    same computation in eager fp32 (`oracle.attn_ref`)."""
    from types import SimpleNamespace
    from flasht5_amd import FlashT5Attention
    cfg = SimpleNamespace(d_model=128, d_kv=64, num_heads=2, relative_attention_num_buckets=32,
                          relative_attention_max_distance=64, is_decoder=decoder, attention_type=attention_type,
                          position_encoding_type="t5", attention_scale=None)
    torch.manual_seed(31)
    blk0 = FlashT5Attention(cfg, has_positional_encoding=True, is_causal=decoder).cuda().bfloat16()
    blk1 = FlashT5Attention(cfg, has_positional_encoding=False, is_causal=decoder).cuda().bfloat16()
    B, S = 2, 200
    x = torch.randn(B, S, cfg.d_model, device="cuda").bfloat16()
    gy = torch.randn(B, S, cfg.d_model, device="cuda").bfloat16()
    y0, pb = blk0(x)
    y1, pb1 = blk1(y0, position_bias=pb)
    params = [p for m in (blk0, blk1) for p in m.parameters()]
    grads = torch.autograd.grad(y1, params, gy)

    # eager fp32 restatement with the same weights
    def eager(blk, h, table):
        H, Dh = cfg.num_heads, cfg.d_kv
        q = (h @ blk.Wq.weight.float().t()).view(B, S, H, Dh).permute(0, 2, 1, 3)
        k = (h @ blk.Wk.weight.float().t()).view(B, S, H, Dh).permute(0, 2, 1, 3)
        v = (h @ blk.Wv.weight.float().t()).view(B, S, H, Dh).permute(0, 2, 1, 3)
        bias = oracle.compute_bias(table, S, S, not decoder, 32, 64)
        o = oracle.attn_ref(q, k, v, bias, 1.0 / math.sqrt(H), causal=decoder, upcast=True)
        return o.permute(0, 2, 1, 3).reshape(B, S, H * Dh) @ blk.o.weight.float().t()
    ref_params = [p.detach().float().clone().requires_grad_() for p in params]
    class _W:  # same attribute names, fp32 leaves
        pass
    def wrap(offset, has_table):
        w = _W()
        names = (["table"] if has_table else []) + ["Wq", "Wk", "Wv", "o"]
        for i, n in enumerate(names):
            setattr(w, n, SimpleNamespace(weight=ref_params[offset + i]))
        return w
    names0 = [n for n, _ in blk0.named_parameters()]
    assert names0 == ["pe_encoding.relative_attention_bias.weight", "Wq.weight", "Wk.weight", "Wv.weight", "o.weight"]
    w0, w1 = wrap(0, True), wrap(5, False)
    table = ref_params[0]
    r0 = eager(w0, x.float(), table)
    r1 = eager(w1, r0, table)
    ref_grads = torch.autograd.grad(r1, ref_params, gy.float())
    assert maxdiff(y1, r1) <= 3e-2 * max(1.0, r1.abs().max().item())
    for (n, _), g, rg in zip(list(blk0.named_parameters()) + list(blk1.named_parameters()), grads, ref_grads):
        assert torch.isfinite(g.float()).all(), n
        assert maxdiff(g, rg) <= 4e-2 * max(1.0, rg.abs().max().item()), (n, maxdiff(g, rg), rg.abs().max().item())
    if attention_type == "fat5_rpe":
        assert isinstance(pb, tuple) and pb[0].shape == (2, 2 * 64 + 1) and pb1 is pb
    else:
        assert pb.shape == (1, 2, S, S) and pb1 is pb


def test_mini_encoder_training_step_all_three_operators():
    """A two-block pre-norm T5 encoder slice + LM head + loss built ONLY from the path's three operators (RMSNorm,
    attention with the shared T5 bias, cross-entropy + z-loss) and plain Linear layers: one training step's loss and
    parameter gradients against the same network in eager fp32.  (The shape of config 5 at toy size.)"""
    from types import SimpleNamespace
    from flasht5_amd import FlashT5Attention, FlashT5LayerNorm, FlashT5CrossEntropyLoss
    torch.manual_seed(77)
    B, S, dm, H, Dh, V = 2, 160, 128, 2, 64, 512
    cfg = SimpleNamespace(d_model=dm, d_kv=Dh, num_heads=H, relative_attention_num_buckets=32,
                          relative_attention_max_distance=128, is_decoder=False, attention_type="fat5_rpe",
                          position_encoding_type="t5", attention_scale=None)
    norms = [FlashT5LayerNorm(dm).cuda().bfloat16() for _ in range(3)]
    attns = [FlashT5Attention(cfg, has_positional_encoding=(i == 0)).cuda().bfloat16() for i in range(2)]
    head = torch.nn.Linear(dm, V, bias=False).cuda().bfloat16()
    crit = FlashT5CrossEntropyLoss(z_loss_factor=1e-4, label_smoothing=0.0)
    with torch.no_grad():
        for n in norms:
            n.weight.copy_(1.0 + 0.1 * torch.randn(dm))
    x = (torch.randn(B, S, dm, device="cuda") * 0.5).bfloat16()
    labels = torch.randint(0, V, (B, S), device="cuda")
    labels[1, -7:] = -100

    h, pb = x, None
    for i in range(2):
        a, pb = attns[i](norms[i](h), position_bias=pb)
        h = h + a
    loss = crit(head(norms[2](h)), labels)
    params = [p for m in (*norms, *attns, head) for p in m.parameters()]
    grads = torch.autograd.grad(loss, params)

    # eager fp32 twin
    rp = [p.detach().float().clone().requires_grad_() for p in params]
    it = iter(rp)
    nw = [next(it) for _ in range(3)]
    a0 = {n: next(it) for n in ("table", "Wq", "Wk", "Wv", "o")}
    a1 = {n: next(it) for n in ("Wq", "Wk", "Wv", "o")}
    hw = next(it)
    def rms(t, w):
        return w * (t * torch.rsqrt(t.pow(2).mean(-1, keepdim=True) + 1e-6))
    def att(t, w):
        q = (t @ w["Wq"].t()).view(B, S, H, Dh).permute(0, 2, 1, 3)
        k = (t @ w["Wk"].t()).view(B, S, H, Dh).permute(0, 2, 1, 3)
        v = (t @ w["Wv"].t()).view(B, S, H, Dh).permute(0, 2, 1, 3)
        bias = oracle.compute_bias(a0["table"], S, S, True, 32, 128)
        o = oracle.attn_ref(q, k, v, bias, 1.0 / math.sqrt(H), causal=False, upcast=True)
        return o.permute(0, 2, 1, 3).reshape(B, S, H * Dh) @ w["o"].t()
    hr = x.float()
    hr = hr + att(rms(hr, nw[0]), a0)
    hr = hr + att(rms(hr, nw[1]), a1)
    logits = rms(hr, nw[2]) @ hw.t()
    flat, lab = logits.view(-1, V), labels.view(-1)
    per = torch.nn.functional.cross_entropy(flat, lab, reduction="none", ignore_index=-100)
    per = per + 1e-4 * torch.logsumexp(flat, -1).square() * (lab != -100)
    rloss = per.mean()
    rgrads = torch.autograd.grad(rloss, rp)
    assert abs(loss.item() - rloss.item()) <= 2e-2 * max(1.0, abs(rloss.item())), (loss.item(), rloss.item())
    for i, (g, rg) in enumerate(zip(grads, rgrads)):
        assert torch.isfinite(g.float()).all(), i
        assert maxdiff(g, rg) <= 6e-2 * max(rg.abs().max().item(), 1e-3) + 1e-5, (i, maxdiff(g, rg), rg.abs().max().item())


@pytest.mark.parametrize("causal", [False, True])
def test_varlen_with_rpe_bias(causal):
    """Packed self-attention with the in-kernel T5 bias (positions local to each sequence) -- what a packed UL2 batch
    needs instead of padding: o, dq, dk, dv and the generator gradient vs per-sequence eager fp32 attention whose dense
    bias is gathered from the same generator."""
    from flasht5_amd import flash_attn_varlen_func
    H, D, R = 3, 64, 32
    cu = [0, 200, 200, 331, 400, 912]          # an empty sequence, a long one (far tiles), short ones
    T = cu[-1]
    g = torch.Generator().manual_seed(23)
    q = torch.randn(T, H, D, generator=g).bfloat16().cuda()
    k = torch.randn(T, H, D, generator=g).bfloat16().cuda()
    v = torch.randn(T, H, D, generator=g).bfloat16().cuda()
    do = torch.randn(T, H, D, generator=g).bfloat16().cuda()
    r1 = (torch.randn(H, 2 * R + 1, generator=g) * 0.5).cuda()
    leaves = [t.clone().requires_grad_() for t in (q, k, v, r1)]
    cu_t = torch.tensor(cu, dtype=torch.int32).cuda()
    o = flash_attn_varlen_func(leaves[0], leaves[1], leaves[2], cu_t, cu_t, 512, 512, causal, 0.125, leaves[3], R)
    dq, dk, dv, d1 = torch.autograd.grad(o, leaves, do)

    ref_leaves = [t.float().clone().requires_grad_() for t in (q, k, v, r1)]
    outs = []
    for i in range(len(cu) - 1):
        s, e = cu[i], cu[i + 1]
        if e == s:
            continue
        n = e - s
        idx = torch.clamp(torch.arange(n)[None, :] - torch.arange(n)[:, None], -R, R).cuda() + R
        bias = ref_leaves[3][:, idx].unsqueeze(0)                       # (1, H, n, n)
        qi, ki, vi = (ref_leaves[j][s:e].permute(1, 0, 2).unsqueeze(0) for j in range(3))
        outs.append(oracle.attn_ref(qi, ki, vi, bias, 0.125, causal=causal, upcast=True)[0].permute(1, 0, 2))
    ref_o = torch.zeros(T, H, D, device="cuda")
    pos = [(cu[i], cu[i + 1]) for i in range(len(cu) - 1) if cu[i + 1] > cu[i]]
    for (s, e), oi in zip(pos, outs):
        ref_o[s:e] = oi
    rdq, rdk, rdv, rd1 = torch.autograd.grad(sum((oi * do[s:e].float()).sum() for (s, e), oi in zip(pos, outs)), ref_leaves)
    assert maxdiff(o, ref_o) <= bound(ref_o, torch.bfloat16)
    for got, ref, key in ((dq, rdq, "dq"), (dk, rdk, "dk"), (dv, rdv, "dv")):
        assert maxdiff(got, ref) <= gbound(ref, torch.bfloat16), key
    assert maxdiff(d1, rd1) <= 1e-2 * max(1.0, rd1.abs().max().item()) + 3e-2


# ---- round-2 edge cases ------------------------------------------------------------------------------------------
def test_masked_bias_finfo_min_fully_masked_row():
    """`use_masking` folds the attention mask into the bias as finfo(dtype).min (reference modeling_flash_t5.py:266-270).
    finfo(bf16).min * log2e overflows fp32: the kernels keep the scaled bias finite, so a row whose keys are ALL masked is a
    uniform softmax like in the reference (`attn_ref` / its Triton kernel scale (s - m), not s), and partially masked rows
    ignore the masked keys.  Truth = autograd through the eager fp32 `attn_ref` restatement (softmax, not exp(s - L): with
    |L| ~ 3e38 that formula has no digits left -- which is also why the BACKWARD of a fully masked row is defined here as
    zero: FA2 recomputes p = exp(s - L); such rows are padding and carry do = 0 in the model)."""
    for dtype in (torch.bfloat16, torch.float16):
        B, H, M, N, D = 2, 2, 96, 160, 64
        q, k, v, _, do = make_inputs(B, H, M, N, D, dtype, None, seed=21)
        bias = (torch.randn(B, H, M, N, generator=torch.Generator().manual_seed(2)) * 0.5).to(dtype).cuda()
        fmin = torch.finfo(dtype).min
        bias[:, :, :, 100:] = fmin          # padded keys
        bias[0, :, 5, :] = fmin             # query rows with every key masked
        bias[1, 1, 40:44, :] = fmin
        do[0, :, 5] = 0
        do[1, 1, 40:44] = 0
        leaves = [t.detach().float().requires_grad_() for t in (q, k, v)]
        o_ref = oracle.attn_ref(leaves[0], leaves[1], leaves[2], bias.float(), 0.125, causal=False, upcast=True)
        g_ref = torch.autograd.grad(o_ref, leaves, do.float())
        ref = {"o": o_ref.detach(), "dq": g_ref[0], "dk": g_ref[1], "dv": g_ref[2]}
        assert torch.isfinite(ref["o"]).all()
        got = run_dense(q, k, v, bias, do, 0.125, False)
        for key in ("o", "dq", "dk", "dv"):
            assert torch.isfinite(got[key].float()).all(), (dtype, key)
            assert maxdiff(got[key], ref[key]) <= (bound if key == "o" else gbound)(ref[key], dtype), (dtype, key)
        assert torch.isfinite(got["db"].float()).all()
        if dtype == torch.bfloat16:  # -3.4e38 absorbs the scores entirely: the fully masked row is the plain mean of v
            assert maxdiff(got["o"][0, :, 5], v[0].float().mean(1)) <= bound(ref["o"], dtype)  # (fp16: -65504 does not)


@pytest.mark.parametrize("M,N,md", [(600, 600, 1024), (2304, 2304, 1024), (700, 900, 512)])
def test_rpe_large_radius(M, N, md):
    """max_distance up to the limit of the linear-memory mode (R = 1024: 98 KiB of per-wave accumulators in the dK/dV
    body, 49 KiB of dynamic LDS in the reduction launch)."""
    from flasht5_amd import flash_attention_v2_rpe
    dtype, B, H, causal, bidir = torch.bfloat16, 1, 2, False, True
    q, k, v, do, table, bias = _rpe_case(B, H, M, N, dtype, causal, bidir, md, seed=M + N + md)
    ref = oracle_all(q, k, v, bias, do, 0.125, causal)
    leaves = [t.detach().clone().requires_grad_() for t in (q, k, v)]
    tb = table.cuda().requires_grad_()
    o = flash_attention_v2_rpe(leaves[0], leaves[1], leaves[2], tb, bidir, 32, md, causal, 0.125)
    dq, dk, dv, dt = torch.autograd.grad(o, leaves + [tb], do)
    _, _, _, _, db_alg = oracle.attn_bwd_oracle(q, k, v, bias, o.detach(), ref["L"], do, 0.125, causal)
    tl = table.clone().requires_grad_()
    oracle.compute_bias(tl, M, N, bidir, 32, md).backward(db_alg.cpu())
    for got, key in ((o, "o"), (dq, "dq"), (dk, "dk"), (dv, "dv")):
        assert maxdiff(got, ref[key]) <= (bound if key == "o" else gbound)(ref[key], dtype), key
    assert maxdiff(dt.cpu(), tl.grad) <= 1e-2 * max(1.0, tl.grad.abs().max().item()) + 3e-2
    with pytest.raises(ValueError):
        flash_attention_v2_rpe(leaves[0], leaves[1], leaves[2], tb, bidir, 32, 2048, causal, 0.125)  # beyond the mode's limit


def test_head_dim_16_rpe_entry_points():
    """head_dim 16 (reference :234 accepts it) through the linear-memory entry points as well"""
    from flasht5_amd import flash_attention_v2_rpe, flash_attention_v2_rpe1d
    from flasht5_amd import positional_encoding as pe
    B, H, M, N, D = 2, 3, 130, 200, 16
    q, k, v, _, do = make_inputs(B, H, M, N, D, torch.bfloat16, None, seed=5)
    table = (torch.randn(32, H, generator=torch.Generator().manual_seed(6)) * 0.5)
    bias = oracle.compute_bias(table, M, N, True, 32, 128).contiguous().cuda()
    ref = oracle_all(q, k, v, bias, do, 0.25, False)
    leaves = [t.detach().clone().requires_grad_() for t in (q, k, v)]
    tb = table.cuda().requires_grad_()
    o = flash_attention_v2_rpe(leaves[0], leaves[1], leaves[2], tb, True, 32, 128, False, 0.25)
    dq, dk, dv, dt = torch.autograd.grad(o, leaves + [tb], do)
    assert o.shape == q.shape and dq.shape == q.shape and dk.shape == k.shape
    for got, key in ((o, "o"), (dq, "dq"), (dk, "dk"), (dv, "dv")):
        assert maxdiff(got, ref[key]) <= (bound if key == "o" else gbound)(ref[key], torch.bfloat16), key
    r1 = pe.rpe1d_from_table(tb.detach(), True, 32, 128)
    o1 = flash_attention_v2_rpe1d(q, k, v, r1, 128, False, 0.25)
    assert torch.equal(o1, o.detach())


def test_bad_arguments_raise_round2():
    """shape / dtype / device mistakes are refused by the host mirror instead of reaching the kernels as wild pointers"""
    from flasht5_amd import flash_attention_v2_bias, flash_attention_v2_rpe1d, flash_attn_varlen_func
    q, k, v, b, _ = make_inputs(2, 2, 64, 96, 64, torch.bfloat16, "1h")
    with pytest.raises(ValueError):
        flash_attention_v2_bias(q, k, v, b[:, :, :, :64].contiguous())       # bias N too short
    with pytest.raises(ValueError):
        flash_attention_v2_bias(q, k, v, b[:, :1].expand(3, 2, 64, 96))       # batch 3 is neither 1 nor B
    with pytest.raises(TypeError):
        flash_attention_v2_bias(q, k, v, b.float())
    with pytest.raises(ValueError):
        flash_attention_v2_bias(q, k[:, :1], v[:, :1], None)                  # head count mismatch
    r1 = torch.zeros(2, 257, device="cuda")
    with pytest.raises(ValueError):
        flash_attention_v2_rpe1d(q, k, v, r1[:, :200], 128)
    qp, kp = torch.zeros(64, 2, 64, device="cuda", dtype=torch.bfloat16), torch.zeros(96, 2, 64, device="cuda", dtype=torch.bfloat16)
    cu_q = torch.tensor([0, 32, 64], dtype=torch.int64)          # CPU int64: converted, not dereferenced on the host
    cu_k = torch.tensor([0, 40, 96], dtype=torch.int32, device="cuda")
    o = flash_attn_varlen_func(qp, kp, kp, cu_q, cu_k, 32, 56)
    assert o.shape == qp.shape
    with pytest.raises(ValueError):
        flash_attn_varlen_func(qp, kp, kp, cu_q[:2], cu_k, 32, 56)
    with pytest.raises(ValueError):
        flash_attn_varlen_func(qp, kp, kp, cu_q, cu_k, 32, 56, rpe1d=r1[:, :100], radius=128)


@pytest.mark.parametrize("S,mode,world", [(512, "rpe", 8), (512, "rpe", 5), (1024, "none", 4), (2048, "rpe", 8), (512, "dense", 8)])
def test_unit_range_shards_match_unsharded(S, mode, world):
    """the B*H (batch, head) units dealt to `world` ranks in head-major chunks, every chunk as ONE forward and ONE
    backward call on this device (`AttentionPlan(units=...)` -> fat5_attn_params.unit_begin / unit_count): o, dq, dk, dv of
    the assembled shards equal the unsharded run BIT FOR BIT, and the shards' partial bias-table gradients add up (fp32 order)
    to the unsharded table gradient -- what the ranks' single all-reduce computes."""
    from flasht5_amd.flash_attention_v2_bias import AttentionPlan
    from flasht5_amd import positional_encoding as pe
    from flasht5_amd.sharding import unit_range, shard_units
    B, H, D = 4, 12, 64
    q, k, v, _, do = make_inputs(B, H, S, S, D, torch.bfloat16, None, seed=S + world, strided=True)
    table = (torch.randn(32, H, generator=torch.Generator().manual_seed(3)) * 0.5).cuda()
    kw = {}
    if mode == "rpe":
        kw = dict(rpe1d=pe.rpe1d_from_table(table), radius=128, rpe_bucket=pe.bucket_index32(128, True, 32, 128, "cuda"), num_buckets=32)
    elif mode == "dense":
        kw = dict(bias=pe.compute_bias(table, S, S).expand(B, H, S, S).to(torch.bfloat16).contiguous())
    full = AttentionPlan(q, k, v, do, sm_scale=0.125, **kw)
    o = full.forward().clone()
    dq, dk, dv, db = (t.clone() if t is not None else None for t in full.backward())
    torch.cuda.synchronize()
    poison = lambda t: torch.full_like(t, float("nan"))  # noqa: E731  (a unit outside every range would stay NaN)
    so, sdq, sdk, sdv = poison(o), poison(dq), poison(dk), poison(dv)
    sdb = torch.zeros_like(db) if mode == "rpe" else (poison(db) if db is not None else None)
    for r in range(world):
        ub, uc = unit_range(B, H, world, r)
        assert [(u % B, u // B) for u in range(ub, ub + uc)] == shard_units(B, H, world, r)
        plan = AttentionPlan(q, k, v, do, sm_scale=0.125, units=(ub, uc), **kw)
        for t in (plan.o, plan.dq, plan.dk, plan.dv):
            t.fill_(float("nan"))
        if mode == "dense":
            plan.dbias.fill_(float("nan"))
        po = plan.forward()
        pdq, pdk, pdv, pdb = plan.backward()
        torch.cuda.synchronize()
        for (b, h) in shard_units(B, H, world, r):
            so[b, h], sdq[b, h], sdk[b, h], sdv[b, h] = po[b, h], pdq[b, h], pdk[b, h], pdv[b, h]
            if mode == "dense":
                sdb[b, h] = pdb[b, h]
        # nothing outside the range was written
        mask = torch.ones(B, H, dtype=torch.bool)
        for (b, h) in shard_units(B, H, world, r):
            mask[b, h] = False
        assert torch.isnan(po.float()[mask.cuda()]).all() and torch.isnan(pdk.float()[mask.cuda()]).all()
        if mode == "rpe":
            heads = sorted({h for _, h in shard_units(B, H, world, r)})
            other = [h for h in range(H) if h not in heads]
            assert (pdb[:, other] == 0).all()  # heads without a unit in the range: zero partial gradient
            sdb += pdb                          # the all-reduce
    assert torch.equal(so, o) and torch.equal(sdq, dq) and torch.equal(sdk, dk) and torch.equal(sdv, dv)
    if mode == "rpe":
        assert maxdiff(sdb, db) <= 1e-5 * max(1.0, db.abs().max().item())
    elif mode == "dense":
        assert torch.equal(sdb, db)


def test_rpe_generator_cache_follows_table_updates():
    """flash_attention_v2_rpe remembers the (H, 2R+1) generator of a table until the table is modified in place"""
    from flasht5_amd import flash_attention_v2_rpe
    q, k, v, _, _ = make_inputs(1, 2, 128, 128, 64, torch.bfloat16, None, seed=3)
    table = (torch.randn(32, 2, generator=torch.Generator().manual_seed(1)) * 0.5).cuda()
    o1 = flash_attention_v2_rpe(q, k, v, table)
    assert torch.equal(flash_attention_v2_rpe(q, k, v, table), o1)
    with torch.no_grad():
        table.add_(1.0)[:, 0].mul_(-3.0)     # what an optimizer step does
    o2 = flash_attention_v2_rpe(q, k, v, table)
    o2_ref = flash_attention_v2_rpe(q, k, v, table.clone())
    assert torch.equal(o2, o2_ref) and not torch.equal(o2, o1)


@pytest.mark.parametrize("B,H,M,N,D,causal,dtype", [
    (4, 3, 256, 256, 64, False, torch.bfloat16), (2, 2, 200, 333, 64, False, torch.bfloat16), (3, 2, 300, 300, 64, True, torch.bfloat16),
    (6, 2, 192, 256, 64, False, torch.bfloat16),   # batch beyond the kernel's 4-element register chunk: fp32 pass-through scratch
    (9, 1, 130, 70, 32, True, torch.float16), (4, 2, 128, 192, 128, False, torch.bfloat16), (5, 2, 520, 77, 64, True, torch.bfloat16),
    (3, 2, 256, 320, 64, True, torch.bfloat16), (2, 2, 384, 256, 64, True, torch.bfloat16)])   # causal, N % 8 == 0, M != N: whole-chunk skips of the reduction
@pytest.mark.parametrize("split", [True, False])
def test_dense_dbias_batch_inner_kernel(B, H, M, N, D, causal, dtype, split, monkeypatch):
    """(1, H, M, N) bias shared by the batch: the bias gradient comes from the batch-inner dBias kernel (attn_bwd_dbias.h) --
    no (B, H, M, N) staging (workspace O(B*H*M)), each term rounded to the bias dtype before the sum like the reference
    (flash_attention_v2_bias.py:720, :214).  Against the oracle, and against the staged + reduced path it replaces.
    split: two wave groups sharing the batch (round 4, D <= 64; the default) / the one-group form."""
    import ctypes
    from flasht5_amd.flash_attention_v2_bias import AttentionPlan
    from flasht5_amd import _lib
    q, k, v, b, do = make_inputs(B, H, M, N, D, dtype, "1h", seed=B * M + N)
    ref = oracle_all(q, k, v, b, do, 0.25, causal)
    res = {}
    for mode in ("2", "0"):  # 2: the batch-inner kernel also below its size threshold; 0: the staged path
        plan = AttentionPlan(q, k, v, do, bias=b, causal=causal, sm_scale=0.25,
                             variant=(_lib.V_DBIAS_INKERNEL | (0 if split else _lib.V_DBIAS_NOSPLIT)) if mode == "2" else _lib.V_DBIAS_STAGED)
        plan.forward()
        plan.dbias.fill_(float("nan"))
        plan.ws.view(torch.uint8).fill_(255)  # (NaN patterns: with a causal mask the staged path must not read what the dQ kernel never wrote)
        dq, dk, dv, db = (t.clone() for t in plan.backward())
        torch.cuda.synchronize()
        res[mode] = (dq, dk, dv, db, plan.ws.numel())
    dq, dk, dv, db, ws = res["2"]
    assert torch.isfinite(db.float()).all()
    assert maxdiff(db, ref["db"]) <= gbound(ref["db"], dtype) * (1 + B)
    for got, key in ((dq, "dq"), (dk, "dk"), (dv, "dv")):
        assert maxdiff(got, ref[key]) <= gbound(ref[key], dtype), key
    # workspace: delta (+ the fp32 (H, M, N) pass-through for B > 4), nothing of size B*H*M*N*2
    assert ws <= B * H * M * 4 + (H * M * N * 4 if B > 4 else 0) + 4096
    assert res["0"][4] >= B * H * M * N * 2
    # same terms, same rounding, same order as the staged path: equal up to one rounding of the sum
    assert maxdiff(db, res["0"][3]) <= 2.0 ** -7 * max(1.0, ref["db"].abs().max().item())
    assert torch.equal(dq, res["0"][0]) and torch.equal(dk, res["0"][1]) and torch.equal(dv, res["0"][2])


def test_native_host_path_equals_ctypes_path():
    """the C++ host path (csrc/torch_binding.cpp: C++ autograd functions on the C ABI) and the Python / ctypes path run the same
    launches: outputs and gradients are bit-identical, for the dense-bias, table and 1-D generator entry points"""
    from flasht5_amd import _lib, flash_attention_v2_bias, flash_attention_v2_rpe, flash_attention_v2_rpe1d
    from flasht5_amd.flash_attention_v2_bias import FlashAttentionAdditiveBias, FlashAttentionRPE, FlashAttentionRPE1D
    from flasht5_amd import positional_encoding as pe
    assert _lib.native() is not None, "lib/_fat5_torch.so missing: python flasht5_amd/build.py"
    q, k, v, b, do = make_inputs(2, 3, 200, 264, 64, torch.bfloat16, "1h", seed=21, strided=True)
    table = (torch.randn(32, 3, generator=torch.Generator().manual_seed(3)) * 0.5).cuda()

    def grads(fn, extra):
        leaves = [t.detach().clone().requires_grad_() for t in (q, k, v, extra)]
        o = fn(*leaves)
        return [o.detach()] + list(torch.autograd.grad(o, leaves, do))

    pairs = [
        (lambda q_, k_, v_, b_: flash_attention_v2_bias(q_, k_, v_, b_, True, 0.125),
         lambda q_, k_, v_, b_: FlashAttentionAdditiveBias.apply(q_, k_, v_, b_, True, 0.125), b),
        (lambda q_, k_, v_, t_: flash_attention_v2_rpe(q_, k_, v_, t_, True, 32, 128, False, 0.125),
         lambda q_, k_, v_, t_: FlashAttentionRPE.apply(q_, k_, v_, t_, True, 32, 128, False, 0.125), table),
        (lambda q_, k_, v_, r_: flash_attention_v2_rpe1d(q_, k_, v_, r_, 128, False, 0.125),
         lambda q_, k_, v_, r_: FlashAttentionRPE1D.apply(q_, k_, v_, r_, 128, False, 0.125), pe.rpe1d_from_table(table)),
    ]
    for native_fn, ctypes_fn, extra in pairs:
        for x, y in zip(grads(native_fn, extra), grads(ctypes_fn, extra)):
            assert x.dtype == y.dtype and x.shape == y.shape and torch.equal(x, y)
    # no-bias call, D = 16
    q16, k16, v16, _, do16 = make_inputs(1, 2, 96, 96, 16, torch.bfloat16, None, seed=4)
    ln = [t.detach().clone().requires_grad_() for t in (q16, k16, v16)]
    lc = [t.detach().clone().requires_grad_() for t in (q16, k16, v16)]
    on = flash_attention_v2_bias(*ln, None, False, None)
    oc = FlashAttentionAdditiveBias.apply(*lc, None, False, None)
    assert on.shape == (1, 2, 96, 16) and torch.equal(on, oc)
    for x, y in zip(torch.autograd.grad(on, ln, do16), torch.autograd.grad(oc, lc, do16)):
        assert torch.equal(x, y)


@pytest.mark.parametrize("mode", ["self", "cross", "self_rpe", "broken"])
def test_packed_projection_gradients_land_in_one_buffer(mode, monkeypatch):
    """q, k, v as slices of ONE projection output (attention_module.unpack_heads): the attention backward writes dq / dk / dv into
    the slices of one buffer and unpack's backward returns that buffer -- no stack / cat / zero-fill -- with the same values as
    autograd's own select path.  'broken': an op between unpack and attention breaks the packing -> the stack fallback, same values."""
    from flasht5_amd import flash_attention_v2_bias, flash_attention_v2_rpe
    from flasht5_amd.attention_module import unpack_heads
    B, S, N, H, D = 2, 384, 256, 6, 64
    g = torch.Generator().manual_seed(12)
    n = 2 if mode == "cross" else 3
    x = (torch.randn(B, N if mode == "cross" else S, n * H * D, generator=g) * 0.5).cuda().bfloat16()
    qx = (torch.randn(B, S, H * D, generator=g) * 0.5).cuda().bfloat16()
    table = (torch.randn(32, H, generator=g) * 0.5).cuda()
    do = torch.randn(B, H, S, D, generator=g).cuda().bfloat16()

    def run(unpack):
        xs, qs = x.clone().requires_grad_(), qx.clone().requires_grad_()
        if unpack:
            parts = unpack_heads(xs, n, H)
        else:
            p5 = xs.view(B, xs.shape[1], n, H, D)
            parts = tuple(p5[:, :, i].permute(0, 2, 1, 3) for i in range(n))
        if mode == "cross":
            q, (k, v) = qs.view(B, S, H, D).permute(0, 2, 1, 3), parts
        else:
            q, k, v = parts
        if mode == "broken":
            k = k * 1.0
        if mode == "self_rpe":
            o = flash_attention_v2_rpe(q, k, v, table, True, 32, 128, False, 0.125)
        else:
            o = flash_attention_v2_bias(q, k, v, None, mode != "cross", 0.125)
        o.backward(do)
        return o.detach(), xs.grad, qs.grad

    o0, gx0, gq0 = run(False)
    calls = {"n": 0}
    real_stack = torch.stack
    monkeypatch.setattr(torch, "stack", lambda *a, **k: (calls.__setitem__("n", calls["n"] + 1), real_stack(*a, **k))[1])
    o1, gx1, gq1 = run(True)
    assert torch.equal(o0, o1) and torch.equal(gx0, gx1)
    if mode == "cross":
        assert torch.equal(gq0, gq1)
    # (the Python form of unpack_heads stacks through torch.stack; the C++ form -- the default -- through at::stack, which the patch does not see)
    from flasht5_amd import _lib
    assert calls["n"] == (1 if mode == "broken" and _lib.native() is None else 0)
