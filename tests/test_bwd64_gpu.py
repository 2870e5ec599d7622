"""GPU parity tests of the long-sequence backward bodies (csrc/attn_bwd64.h: dK/dV with 64 keys per wave, dQ with 64 query rows
per wave; software-pipelined step loops, 4-slot LDS rings, AGPR accumulators) -- forced per call with the variant bits FAT5_V_KV64_ON / FAT5_V_Q64_ON (include/fat5.h) at sizes
the oracle finishes in seconds; at (4,12,8192,64) the default dispatch picks them by itself
(test_attention_gpu.py::test_cfg3_properties_s8192)."""
import pytest
import torch

import oracle
from attn_helpers import make_inputs, oracle_all, maxdiff
from test_attention_gpu import bound, gbound, _rpe_case

pytestmark = pytest.mark.gpu


HALF = {"on": False}  # set per test by the module fixture: the half-length (128-key workgroup) variant of the 64-key dK/dV body


def _bits(kv64, q64, fwd64=None):
    from flasht5_amd import _lib
    b = _lib.V_KV64_HALF_ON if HALF["on"] else _lib.V_KV64_HALF_OFF
    if kv64 is not None:
        b |= _lib.V_KV64_ON if kv64 else _lib.V_KV64_OFF
    if q64 is not None:
        b |= _lib.V_Q64_ON if q64 else _lib.V_Q64_OFF
    if fwd64 is not None:
        b |= _lib.V_FWD64_ON if fwd64 else _lib.V_FWD64_OFF
    return b


@pytest.fixture(autouse=True, params=["wg256", "half"])
def force_bwd64(request):
    """every test of this module runs twice: 256-key / 256-row workgroups, and the half-length variants (128-key workgroups whose
    two wave pairs each walk half of the query steps and merge dK^T / dV^T through LDS; likewise for dQ)"""
    from flasht5_amd import _lib
    HALF["on"] = request.param == "half"
    with _lib.variant(_bits(True, True)):
        yield
    HALF["on"] = False


def _grads(q, k, v, do, causal, scale, table=None, bidir=True, md=128):
    from flasht5_amd import flash_attention_v2_bias, flash_attention_v2_rpe
    leaves = [t.detach().clone().requires_grad_() for t in (q, k, v)]
    if table is None:
        o = flash_attention_v2_bias(leaves[0], leaves[1], leaves[2], None, causal, scale)
        dq, dk, dv = torch.autograd.grad(o, leaves, do)
        return {"o": o.detach(), "dq": dq, "dk": dk, "dv": dv}
    tb = table.cuda().requires_grad_()
    o = flash_attention_v2_rpe(leaves[0], leaves[1], leaves[2], tb, bidir, 32, md, causal, scale)
    dq, dk, dv, dt = torch.autograd.grad(o, leaves + [tb], do)
    return {"o": o.detach(), "dq": dq, "dk": dk, "dv": dv, "dtable": dt}


def _table_truth(q, k, v, bias, o, L, do, scale, causal, table, M, N, bidir, md):
    """(truth, rounding allowance) per bucket.  delta from the STORED o (see test_attention_gpu.py::
    test_rpe_mode_matches_dense_oracle).  The pipelined steps sum the dS they rounded to bf16 for the dK GEMM -- what the
    reference's bias gradient is made of, too (`ds.to(dtype)`, flash_attention_v2_bias.py:720) -- so a bucket of n terms carries
    n independent roundings of relative size <= 2^-9: allowance = 4 sigma = 4 * 2^-9 / sqrt(3) * sqrt(sum ds^2)."""
    _, _, _, _, db = oracle.attn_bwd_oracle(q, k, v, bias, o, L, do, scale, causal)
    db = db.cpu()
    tl = table.clone().requires_grad_()
    oracle.compute_bias(tl, M, N, bidir, 32, md).backward(db)
    t2 = table.clone().requires_grad_()
    oracle.compute_bias(t2, M, N, bidir, 32, md).backward(db * db)
    return tl.grad, 4.0 * 2.0 ** -9 / 3 ** 0.5 * t2.grad.sqrt()


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("B,H,M,N,causal,mode,md", [
    (1, 2, 256, 256, False, "none", 128),     # one workgroup: 8 pipelined steps (two trips of the 4-step loop)
    (2, 3, 1024, 1024, False, "none", 128),   # four key blocks per (b, h)
    (2, 3, 1024, 1024, False, "rpe", 128),    # far-positive range, band (general steps), far-negative range
    (1, 2, 1024, 1024, False, "rpe", 32),     # narrow band: pipelined ranges on both sides of every workgroup
    (1, 2, 2048, 2048, True, "rpe", 128),     # causal: the mask rides in the bias table (P = 0), steps above the diagonal skipped
    (1, 2, 1000, 1100, True, "rpe", 128),     # ... 0 < P = N - M < R: the cut at a non-zero table offset, ragged rows / keys
    (1, 2, 1100, 1000, True, "rpe", 128),     # ... -R <= P < 0: dead rows (L = -inf) meet -inf table entries
    (1, 2, 2048, 1952, True, "rpe", 128),     # ... P = -96, whole steps
    (1, 2, 2048, 2048, True, "none", 128),    # causal without bias (round 5): the mask rides in the score MFMAs' C operand, diagonal steps pipelined
    (1, 2, 1000, 1100, True, "none", 128),    # ... bottom-right with N - M = 100, ragged rows / keys (a key-tail wave stays general)
    (1, 2, 300, 2500, True, "none", 128),     # ... M << N
    (1, 2, 1100, 1000, True, "none", 128),    # ... N < M: dead rows meet masked scores
    (1, 2, 1000, 1100, False, "rpe", 128),    # ragged: last step padded (rows past M), key tail workgroup (all general)
    (1, 2, 300, 2500, True, "rpe", 128),      # M << N, bottom-right causal
    (1, 2, 2500, 300, True, "none", 128),     # M >> N: dead rows (lse = -inf) in the pipelined range
    (1, 1, 3000, 520, False, "rpe", 64),      # remainder iterations after the 4-step loop, 1-row-short last step
    (1, 2, 90, 70, False, "rpe", 128),        # fewer steps than ring slots
    (1, 2, 1536, 1536, False, "rpe", 512),    # wide band (radius 512): most steps general, 117 KB of LDS
])
def test_bwd64_matches_oracle(B, H, M, N, causal, mode, md, dtype):
    scale = 0.125
    if mode == "rpe":
        q, k, v, do, table, bias = _rpe_case(B, H, M, N, dtype, causal, True, md, seed=M + 5 * N)
    else:
        q, k, v, _, do = make_inputs(B, H, M, N, 64, dtype, None, seed=M + 5 * N, strided=True)
        table, bias = None, None
    ref = oracle_all(q, k, v, bias, do, scale, causal)
    got = _grads(q, k, v, do, causal, scale, table, True, md)
    for key in ("dq", "dk", "dv"):
        assert torch.isfinite(got[key].float()).all(), key
        assert maxdiff(got[key], ref[key]) <= gbound(ref[key], dtype), key
    if table is not None:
        want, allow = _table_truth(q, k, v, bias, got["o"], ref["L"], do, scale, causal, table, M, N, True, md)
        err = (got["dtable"].cpu() - want).abs()
        assert bool((err <= allow + 2e-3 * max(1.0, want.abs().max().item()) + 1e-2).all()), (err.max().item(), allow.max().item())


def test_bwd64_agrees_with_32key_body(monkeypatch):
    """Both dK/dV bodies on the same inputs: dk / dv equal to output rounding, the diagonal sums to fp32 summation order."""
    from flasht5_amd.flash_attention_v2_bias import AttentionPlan
    from flasht5_amd import positional_encoding as pe
    q, k, v, _, do = make_inputs(2, 4, 2048, 2048, 64, torch.bfloat16, None, seed=3, strided=True)
    table = (torch.randn(32, 4, generator=torch.Generator().manual_seed(1)) * 0.5).cuda()
    outs = []
    from flasht5_amd import _lib
    for f in (False, True):
        _lib.set_variant(_bits(f, f))
        plan = AttentionPlan(q, k, v, do, sm_scale=0.125, need_dbias=True, rpe1d=pe.rpe1d_from_table(table, True, 32, 128), radius=128)
        plan.forward()
        plan.backward()
        torch.cuda.synchronize()
        outs.append((plan.dk.float().clone(), plan.dv.float().clone(), plan.dbias.clone(), plan.dq.float().clone()))
    for i in (0, 1, 3):
        assert (outs[0][i] - outs[1][i]).abs().max().item() <= 2.0 ** -6 * max(1.0, outs[0][i].abs().max().item()), i
    # (far bins: this body sums the bf16-rounded dS, the 32-key body the unrounded ones; ~1e6 terms each)
    assert (outs[0][2] - outs[1][2]).abs().max().item() <= 1e-2 * max(1.0, outs[0][2].abs().max().item())


def test_bwd64_unit_range_shards_are_bit_identical(monkeypatch):
    from flasht5_amd.flash_attention_v2_bias import AttentionPlan
    from flasht5_amd import positional_encoding as pe
    from flasht5_amd.sharding import unit_range
    B, H = 2, 3
    q, k, v, _, do = make_inputs(B, H, 1024, 1024, 64, torch.bfloat16, None, seed=5)
    table = (torch.randn(32, H, generator=torch.Generator().manual_seed(2)) * 0.5).cuda()
    rpe1d = pe.rpe1d_from_table(table, True, 32, 128)
    full = AttentionPlan(q, k, v, do, sm_scale=0.125, need_dbias=True, rpe1d=rpe1d, radius=128)
    full.forward(); full.backward()
    torch.cuda.synchronize()
    dk = torch.zeros_like(full.dk)
    dq = torch.zeros_like(full.dq)
    acc = torch.zeros_like(full.dbias)
    for r in range(2):
        part = AttentionPlan(q, k, v, do, sm_scale=0.125, need_dbias=True, rpe1d=rpe1d, radius=128, units=unit_range(B, H, 2, r))
        part.dk.zero_()
        part.dq.zero_()
        part.forward(); part.backward()
        torch.cuda.synchronize()
        dk += part.dk
        dq += part.dq
        acc += part.dbias
    assert torch.equal(dk, full.dk)
    assert torch.equal(dq, full.dq)
    assert (acc - full.dbias).abs().max().item() <= 1e-4 * max(1.0, full.dbias.abs().max().item())


def test_bwd64_deterministic():
    q, k, v, do, table, _ = _rpe_case(1, 4, 2048, 2048, torch.bfloat16, False, True, 128, seed=9)
    a = _grads(q, k, v, do, False, 0.125, table)
    b = _grads(q, k, v, do, False, 0.125, table)
    for key in ("dq", "dk", "dv", "dtable"):
        assert torch.equal(a[key], b[key]), key


@pytest.mark.parametrize("kv64,q64", [("1", "0"), ("0", "1")])
def test_bwd64_mixed_with_32wide_bodies(monkeypatch, kv64, q64):
    """one 64-wide body next to the other 32-wide one (what the dispatch picks when only M or only N is long): the row statistics
    cross between them in both directions (delta for the 32-key dK/dV body, the pre-scaled pair for the 64-key one)"""
    from flasht5_amd import _lib
    _lib.set_variant(_bits(kv64 == "1", q64 == "1"))
    q, k, v, do, table, bias = _rpe_case(2, 2, 600, 456, torch.bfloat16, True, True, 128, seed=77)
    ref = oracle_all(q, k, v, bias, do, 0.125, True)
    got = _grads(q, k, v, do, True, 0.125, table)
    for key in ("dq", "dk", "dv"):
        assert maxdiff(got[key], ref[key]) <= gbound(ref[key], torch.bfloat16), key
    want, allow = _table_truth(q, k, v, bias, got["o"], ref["L"], do, 0.125, True, table, 600, 456, True, 128)
    err = (got["dtable"].cpu() - want).abs()
    assert bool((err <= allow + 2e-3 * max(1.0, want.abs().max().item()) + 1e-2).all())


@pytest.mark.parametrize("dtype,causal", [(torch.float16, False), (torch.bfloat16, True)])
def test_default_dispatch_long_sequence_slice_vs_oracle(monkeypatch, dtype, causal):
    """(4, 12, 4096, 64): the dispatch picks the 64-wide bodies by itself (fp16 too; causal too).  Two heads of batch 0 against the
    fp32 oracle run on the device, table gradient of the whole batch through the zero-sum property of the softmax Jacobian."""
    from flasht5_amd.flash_attention_v2_bias import AttentionPlan
    from flasht5_amd import positional_encoding as pe
    from flasht5_amd import _lib
    _lib.set_variant(0)  # the library's own choice
    B, H, S = 4, 12, 4096
    q, k, v, _, do = make_inputs(B, H, S, S, 64, dtype, None, seed=12, strided=True)
    table = (torch.randn(32, H, generator=torch.Generator().manual_seed(4)) * 0.5).cuda()
    plan = AttentionPlan(q, k, v, do, rpe1d=pe.rpe1d_from_table(table), radius=128, sm_scale=0.125, causal=causal)
    o = plan.forward().clone()
    dq, dk, dv, d1 = (t.clone() for t in plan.backward())
    torch.cuda.synchronize()
    bias = pe.compute_bias(table[:, :2], S, S).contiguous()
    sl = (slice(0, 1), slice(0, 2))
    qs, ks, vs, dos = (t[sl] for t in (q, k, v, do))
    ref_o, ref_L = oracle.attn_fwd_oracle(qs, ks, vs, bias, 0.125, causal)
    rdq, rdk, rdv, _, _ = oracle.attn_bwd_oracle(qs, ks, vs, bias, ref_o, ref_L, dos, 0.125, causal)
    assert maxdiff(o[sl], ref_o) <= bound(ref_o, dtype)
    for got, ref, name in ((dq[sl], rdq, "dq"), (dk[sl], rdk, "dk"), (dv[sl], rdv, "dv")):
        assert torch.isfinite(got.float()).all(), name
        assert maxdiff(got, ref) <= gbound(ref, dtype), name
    assert torch.isfinite(d1).all()


def test_bwd64_stage_split_equals_one_call():
    """FAT5_BWD_DQ, FAT5_BWD_DKDV, FAT5_BWD_REDUCE as three calls (the dQ stage leaves the row statistics of the dK/dV stage in the
    workspace) = one fat5_attn_bwd call, bit for bit; causal, unit range included"""
    from flasht5_amd.flash_attention_v2_bias import AttentionPlan
    from flasht5_amd import positional_encoding as pe
    q, k, v, _, do = make_inputs(2, 3, 1100, 1300, 64, torch.bfloat16, None, seed=31, strided=True)
    table = (torch.randn(32, 3, generator=torch.Generator().manual_seed(6)) * 0.5).cuda()
    rpe1d = pe.rpe1d_from_table(table)
    for units in (None, (1, 4)):
        a = AttentionPlan(q, k, v, do, sm_scale=0.125, causal=True, need_dbias=True, rpe1d=rpe1d, radius=128, units=units)
        b = AttentionPlan(q, k, v, do, sm_scale=0.125, causal=True, need_dbias=True, rpe1d=rpe1d, radius=128, units=units)
        for p_ in (a, b):
            for t in (p_.dq, p_.dk, p_.dv):
                t.zero_()
            p_.forward()
        a.backward()
        for stage in (1, 2, 4):
            b.backward(stage)
        torch.cuda.synchronize()
        for x, y in ((a.dq, b.dq), (a.dk, b.dk), (a.dv, b.dv), (a.dbias, b.dbias)):
            assert torch.equal(x, y)


@pytest.mark.parametrize("B,H,M,N,causal,mode", [
    (4, 12, 2048, 2048, False, "rpe"),    # the shape the mixed launch is for: 48 (b, h) pairs, 384 256-key workgroups = 1.5 rounds
    (4, 12, 2048, 2048, True, "none"),
    (2, 8, 1100, 1300, False, "rpe"),     # ragged: an odd number of 128-key rows (the last 256-key workgroup has no second row to zero)
    (4, 4, 512, 768, True, "rpe"),        # M < N, bottom-right causal
])
def test_bwd64_mixed_launch_matches_the_256_key_launch(B, H, M, N, causal, mode):
    """256-key and half-length workgroups in ONE launch (attn_bwd_kv64_mixed_kernel: the first pairs of every XCD full, the others
    half-length) against the plain 256-key launch: dq identical (the dQ kernel is the same), dk / dv equal to output rounding of
    the same fp32 sums in another order, the table gradient to fp32 summation order; and against the oracle's bound."""
    from flasht5_amd import _lib
    from flasht5_amd.flash_attention_v2_bias import AttentionPlan
    from flasht5_amd import positional_encoding as pe
    q, k, v, _, do = make_inputs(B, H, M, N, 64, torch.bfloat16, None, seed=M + N, strided=True)
    table = (torch.randn(32, H, generator=torch.Generator().manual_seed(1)) * 0.5).cuda()
    kw = dict(causal=causal, sm_scale=0.125)
    if mode == "rpe":
        kw.update(rpe1d=pe.rpe1d_from_table(table), radius=128, rpe_bucket=pe.bucket_index32(128, True, 32, 128, "cuda"), num_buckets=32)
    outs = []
    for bits in (_lib.V_KV64_ON | _lib.V_Q64_ON | _lib.V_KV64_HALF_OFF | _lib.V_KV64_MIX_OFF, _lib.V_KV64_ON | _lib.V_Q64_ON | _lib.V_KV64_MIX_ON):
        plan = AttentionPlan(q, k, v, do, variant=bits, **kw)
        plan.forward(); plan.backward(); torch.cuda.synchronize()
        outs.append([t.clone() for t in (plan.dq, plan.dk, plan.dv)] + ([plan.dbias.clone()] if plan.dbias is not None else []))
    a, b = outs
    assert torch.equal(a[0], b[0])
    for x, y in zip(a[1:3], b[1:3]):
        assert maxdiff(x, y) <= 2.0 ** -7 * float(x.float().abs().max())
    if mode == "rpe":
        assert maxdiff(a[3], b[3]) <= 1e-3 * max(1.0, float(a[3].abs().max()))
    ref = oracle_all(q, k, v, pe.compute_bias(table, M, N).to(torch.bfloat16) if mode == "rpe" else None, do, 0.125, causal)
    for key, got in zip(("dq", "dk", "dv"), b[:3]):
        assert maxdiff(got, ref[key]) <= gbound(ref[key], torch.bfloat16), key


@pytest.mark.parametrize("B,H,M,N,causal,mode,md", [
    (4, 12, 512, 512, False, "rpe", 128),     # cfg2: 96 + 96 workgroups side by side
    (4, 12, 1536, 1536, False, "rpe", 128),   # 576 workgroups on 256 CUs: dQ workgroups queue behind the dK/dV ones
    (2, 3, 1024, 1024, False, "none", 128),
    (2, 3, 512, 512, True, "none", 128),      # causal without bias in the one-launch form (round 5: the library's choice up to 512 keys): mask in the dK/dV half's C operand
    (1, 2, 1000, 1100, True, "none", 128),
    (1, 2, 2048, 2048, True, "rpe", 128),     # causal: masked (general) steps produce their statistics the same way
    (1, 2, 1000, 1100, True, "rpe", 128),     # causal with the mask in the table at a non-zero offset (0 < N - M < R)
    (1, 2, 1100, 1000, True, "rpe", 128),     # ... N - M < 0: dead rows whose statistics the dK/dV half forms itself
    (1, 2, 2048, 1952, True, "rpe", 128),
    (1, 2, 1000, 1100, False, "rpe", 128),    # ragged: the last step's rows past M (raw L reads as zero there), a key tail
    (1, 2, 300, 2500, True, "rpe", 32),       # M << N
    (1, 2, 2500, 300, True, "none", 128),     # M >> N: dead rows
    (1, 2, 90, 70, False, "rpe", 128),        # fewer steps than ring slots
    (1, 1, 40, 600, False, "rpe", 64),        # two steps only: everything from the prologue
])
def test_bwd64_fused_launch(B, H, M, N, causal, mode, md):
    """attn_bwd_fused64_kernel (variant bit FAT5_V_FUSED64_ON): the 256-key dK/dV body in its self-sufficient form (row statistics
    -L/scale, -delta formed from the step's own O / dO rows two steps ahead of their use) and the dQ body in ONE launch, against the
    oracle and against the two-launch form of the same bodies (dq identical; dk / dv to the last bits of delta's summation order)."""
    from flasht5_amd import _lib
    from flasht5_amd.flash_attention_v2_bias import AttentionPlan
    from flasht5_amd import positional_encoding as pe
    dtype = torch.bfloat16
    q, k, v, _, do = make_inputs(B, H, M, N, 64, dtype, None, seed=3 * M + N, strided=True)
    table = (torch.randn(32, H, generator=torch.Generator().manual_seed(1)) * 0.5).cuda()
    kw = dict(causal=causal, sm_scale=0.125)
    bias = None
    if mode == "rpe":
        kw.update(rpe1d=pe.rpe1d_from_table(table, True, 32, md), radius=md, rpe_bucket=pe.bucket_index32(md, True, 32, md, "cuda"), num_buckets=32)
        bias = pe.compute_bias(table, M, N, True, 32, md).to(dtype)
    outs = []
    for bits in (_lib.V_KV64_ON | _lib.V_Q64_ON | _lib.V_KV64_HALF_OFF | _lib.V_KV64_MIX_OFF | _lib.V_FUSED64_OFF, _lib.V_FUSED64_ON):
        plan = AttentionPlan(q, k, v, do, variant=bits, **kw)
        assert plan.bwd_launches() == (1 if bits == _lib.V_FUSED64_ON else 2)
        plan.forward(); plan.backward(); torch.cuda.synchronize()
        outs.append([t.clone() for t in (plan.dq, plan.dk, plan.dv)] + ([plan.dbias.clone()] if plan.dbias is not None else []))
    a, b = outs
    assert torch.equal(a[0], b[0])
    for x, y in zip(a[1:3], b[1:3]):
        assert torch.isfinite(y.float()).all()
        assert maxdiff(x, y) <= 2.0 ** -7 * max(1e-6, float(x.float().abs().max()))
    ref = oracle_all(q, k, v, bias, do, 0.125, causal)
    for key, got in zip(("dq", "dk", "dv"), b[:3]):
        assert maxdiff(got, ref[key]) <= gbound(ref[key], dtype), key
    if mode == "rpe":
        # Round 6: where the whole launch is resident at once the one-launch form has its dQ workgroups form the per-diagonal sums (tests/test_qdiag_gpu.py);
        # both sides sum the same dS -- unrounded in band / general steps, rounded to the input dtype in far trips -- but which tiles are far trips follows the
        # wave's 64 rows there and its 64 keys here: the two tables differ by the rounding noise of the far bins (inside twice the allowance of _table_truth),
        # and each is checked against the oracle
        want, allow = _table_truth(q, k, v, bias.float(), plan.o, ref["L"], do, 0.125, causal, table.cpu(), M, N, True, md)
        for t in (a[3], b[3]):
            err = (t.cpu() - want).abs()
            assert bool((err <= allow + 2e-3 * max(1.0, want.abs().max().item()) + 1e-2).all()), (err.max().item(), allow.max().item())
        diff = (a[3] - b[3]).abs().cpu()
        assert bool((diff <= 2 * allow + 1e-3 * max(1.0, float(a[3].abs().max()))).all()), diff.max().item()
