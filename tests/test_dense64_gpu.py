"""GPU parity tests of the dense-bias bodies of round 5 -- the reference's own operator, one (1, H, M, N) bias shared by the batch
(flash_attention_v2_bias.py:228-288; caller modeling_flash_t5.py:280-285):

 * csrc/attn_bwd_qdb64.h: dQ and the batch-reduced dbias in ONE kernel (four batch elements per workgroup, the batch sum of the rounded dS
   through LDS on the matrix pipe), forced / forbidden per call with FAT5_V_QDB64_ON / _OFF;
 * the dense instantiations of the 64-key dK/dV body (csrc/attn_bwd64.h, FAT5_V_KV64_ON) and of the 64-row forward (csrc/attn_fwd64.h, FAT5_V_FWD64_ON).

Against the fp32 oracle at sizes it finishes in seconds, and against the bodies they replace (same terms, same roundings)."""
import pytest
import torch

from attn_helpers import make_inputs, oracle_all, maxdiff
from test_attention_gpu import bound, gbound

pytestmark = pytest.mark.gpu


def _plan(q, k, v, do, b, causal, scale, bits):
    from flasht5_amd.flash_attention_v2_bias import AttentionPlan
    plan = AttentionPlan(q, k, v, do, bias=b, causal=causal, sm_scale=scale, variant=bits)
    plan.forward()
    plan.dbias.fill_(float("nan"))
    plan.ws.view(torch.uint8).fill_(255)  # (NaN patterns in the workspace: nothing may be read that was not written)
    plan.dq.fill_(float("nan"))
    out = [t.clone() for t in plan.backward()]
    torch.cuda.synchronize()
    return plan, out


@pytest.mark.parametrize("B,H,M,N,causal", [
    (4, 3, 256, 256, False),     # one trip of the 3-step loop and change, four waves = four batch elements
    (4, 2, 512, 512, True),      # causal: masked (general) steps on the diagonal, tiles above it never visited (dbias zero there)
    (2, 2, 200, 336, False),     # two idle waves, ragged rows, a key tail (N % 32 != 0)
    (3, 2, 300, 304, True),      # three batch elements, bottom-right causal with N - M = 4
    (1, 2, 256, 320, False),     # forced at B = 1: three idle waves
    (6, 2, 192, 256, False),     # B > 4: two groups -> fp32 slabs + partial reduction
    (16, 2, 256, 256, True),     # the reference benchmark's batch: four groups, causal (the reduction knows the mask)
    (5, 2, 520, 72, True),       # M >> N: dead rows (lse = -inf), a single partly filled key step
    (4, 1, 64, 2048, False),     # one row block, many trips
    (8, 12, 128, 1024, False),   # grid over heads x groups: the XCD-aware decode with 24 (head, group) pairs
    (4, 2, 1000, 1096, True),    # ragged rows and keys with the mask
])
def test_qdb64_matches_oracle_and_the_staged_path(B, H, M, N, causal):
    from flasht5_amd import _lib
    dtype = torch.bfloat16
    q, k, v, b, do = make_inputs(B, H, M, N, 64, dtype, "1h", seed=B * M + N, strided=True)
    ref = oracle_all(q, k, v, b, do, 0.25, causal)
    pn, new = _plan(q, k, v, do, b, causal, 0.25, _lib.V_QDB64_ON)
    po, old = _plan(q, k, v, do, b, causal, 0.25, _lib.V_QDB64_OFF | _lib.V_DBIAS_STAGED)
    assert pn.describe()["dq"] == "64row-batch4" and po.describe()["dq"] != "64row-batch4"
    dq, dk, dv, db = new
    for got, key in ((dq, "dq"), (dk, "dk"), (dv, "dv")):
        assert torch.isfinite(got.float()).all(), key
        assert maxdiff(got, ref[key]) <= gbound(ref[key], dtype), key
    assert torch.isfinite(db.float()).all()
    assert maxdiff(db, ref["db"]) <= gbound(ref["db"], dtype) * (1 + B)
    # same terms (dS rounded to bf16 per batch element), fp32 sum, one rounding -- as the staged path: equal up to one rounding of the sum
    assert maxdiff(db, old[3]) <= 2.0 ** -7 * max(1.0, ref["db"].abs().max().item())
    assert maxdiff(dq, old[0]) <= 2.0 ** -7 * max(1.0, ref["dq"].abs().max().item())
    # workspace: no (B, H, M, N) tensor
    assert pn.ws.numel() <= B * H * (M + 64) * 4 * 3 + ((B + 3) // 4 > 1) * ((B + 3) // 4) * H * M * N * 4 + 8192
    if causal:  # above the diagonal: exact zeros
        keep = torch.arange(N, device="cuda")[None, :] <= torch.arange(M, device="cuda")[:, None] + (N - M)
        assert bool((db[0][:, ~keep] == 0).all())


def test_qdb64_deterministic_and_default_dispatch():
    from flasht5_amd import _lib
    q, k, v, b, do = make_inputs(4, 12, 1024, 1024, 64, torch.bfloat16, "1h", seed=9, strided=True)
    p0, a = _plan(q, k, v, do, b, False, 0.125, 0)
    assert p0.describe()["dq"] == "64row-batch4" and p0.describe()["dkdv"] == "64key"  # the library's own choice for the model's case (from 2^25 scores per call on)
    _, c = _plan(q, k, v, do, b, False, 0.125, 0)
    for x, y in zip(a, c):
        assert torch.equal(x, y)
    # not legal -> the older paths, same answers to rounding: a per-batch bias, fp16, keys not a multiple of 8
    qb, kb, vb, bb, dob = make_inputs(2, 2, 256, 256, 64, torch.bfloat16, "bh", seed=3)
    pb, _ = _plan(qb, kb, vb, dob, bb, False, 0.125, 0)
    assert pb.describe()["dq"] != "64row-batch4"
    qh, kh, vh, bh_, doh = make_inputs(2, 2, 256, 252, 64, torch.bfloat16, "1h", seed=4)
    ph, got = _plan(qh, kh, vh, doh, bh_, False, 0.125, 0)
    assert ph.describe()["dq"] != "64row-batch4"
    ref = oracle_all(qh, kh, vh, bh_, doh, 0.125, False)
    assert maxdiff(got[3], ref["db"]) <= gbound(ref["db"], torch.bfloat16) * 3


def test_qdb64_strided_bias_view_and_masked_bias():
    """a bias that is a (1, H, M, N) view of a larger tensor (row pitch > N), with finfo.min entries (`use_masking`, modeling_flash_t5.py:266-270)"""
    from flasht5_amd import _lib
    B, H, M, N = 4, 2, 256, 256
    q, k, v, _, do = make_inputs(B, H, M, N, 64, torch.bfloat16, None, seed=12, strided=True)
    big = torch.randn(1, H, M, N + 64, generator=torch.Generator().manual_seed(2)).bfloat16().cuda()
    big[..., 200:256] = torch.finfo(torch.bfloat16).min  # a padded tail of keys masked for every row
    b = big[..., :N]
    ref = oracle_all(q, k, v, b, do, 0.125, False)
    _, got = _plan(q, k, v, do, b, False, 0.125, _lib.V_QDB64_ON)
    for x, key in zip(got, ("dq", "dk", "dv")):
        assert maxdiff(x, ref[key]) <= gbound(ref[key], torch.bfloat16), key
    assert maxdiff(got[3], ref["db"]) <= gbound(ref["db"], torch.bfloat16) * (1 + B)


@pytest.mark.parametrize("B,H,M,N,causal,kind", [
    (2, 3, 1024, 1024, False, "1h"),   # four key blocks per (b, h): trips of the pipelined dense iteration
    (1, 2, 256, 256, False, "1h"),     # one workgroup, two trips
    (2, 2, 1024, 1024, True, "1h"),    # causal: diagonal steps general, steps above the diagonal skipped
    (1, 2, 1000, 1096, False, "1h"),   # ragged: last step padded (rows past M), a key-tail workgroup (all general)
    (1, 2, 300, 2504, True, "1h"),     # M << N, bottom-right causal
    (1, 2, 2500, 296, True, "1h"),     # M >> N: dead rows (lse = -inf) in the pipelined range
    (1, 1, 3000, 520, False, "1h"),    # remainder iterations after the 4-step loop
    (1, 2, 90, 72, False, "1h"),       # fewer steps than ring slots
    (2, 2, 512, 768, False, "bh"),     # a per-batch bias (B, H, M, N): dbias = dS, no reduction
    (3, 2, 384, 512, False, "11"),     # one bias for all heads: head + batch reduction by the older dQ path, dK/dV by the 64-key body
])
def test_kv64_dense_matches_oracle_and_the_32key_body(B, H, M, N, causal, kind):
    """the dense instantiation of the 64-key dK/dV body (FAT5_V_KV64_ON): dk / dv against the oracle and against the 32-key body (same roundings)"""
    from flasht5_amd import _lib
    dtype = torch.bfloat16
    q, k, v, b, do = make_inputs(B, H, M, N, 64, dtype, kind, seed=3 * M + N, strided=True)
    ref = oracle_all(q, k, v, b, do, 0.125, causal)
    pn, new = _plan(q, k, v, do, b, causal, 0.125, _lib.V_KV64_ON)
    po, old = _plan(q, k, v, do, b, causal, 0.125, _lib.V_KV64_OFF)
    assert pn.describe()["dkdv"] == "64key" and po.describe()["dkdv"] == "32key"
    for i, key in enumerate(("dq", "dk", "dv")):
        assert torch.isfinite(new[i].float()).all(), key
        assert maxdiff(new[i], ref[key]) <= gbound(ref[key], dtype), key
    for i in (1, 2):
        assert maxdiff(new[i], old[i]) <= 2.0 ** -6 * max(1.0, float(old[i].float().abs().max()))
    nred = (B if b.shape[0] == 1 else 1) * (H if b.shape[1] == 1 else 1)
    assert maxdiff(new[3], ref["db"]) <= gbound(ref["db"], dtype) * (1 + nred)


@pytest.mark.parametrize("scale", [1.3, 0.0884, -0.5, 1.0])
@pytest.mark.parametrize("B,H,M,N,causal", [(4, 2, 512, 512, True), (3, 2, 300, 560, False)])
def test_dense_bodies_at_scales_that_are_not_powers_of_two(B, H, M, N, causal, scale):
    """The bias rides into the scores as bias * (1 / scale) on the matrix pipe, 1 / scale as two 16-bit terms: the reference benchmarks at sm_scale 1.3
    (benchmarks/bench_fa2_bias.py), 1 / sqrt(128) = 0.0884, a negative scale, and T5's 1.0 -- forward, dQ + dbias (qdb64) and the 64-key dK/dV body."""
    from flasht5_amd import _lib
    dtype = torch.bfloat16
    q, k, v, b, do = make_inputs(B, H, M, N, 64, dtype, "1h", seed=B * M + N + 1, strided=True)
    if abs(scale) > 1.0:  # (keep the logits in a sane range, as the reference's benchmark inputs do)
        q = (q.float() * 0.5).to(dtype)
    ref = oracle_all(q, k, v, b, do, scale, causal)
    pn, new = _plan(q, k, v, do, b, causal, scale, _lib.V_QDB64_ON | _lib.V_KV64_ON)
    assert pn.describe()["dq"] == "64row-batch4" and pn.describe()["dkdv"] == "64key"
    assert maxdiff(pn.o, ref["o"]) <= bound(ref["o"], dtype)
    fin = torch.isfinite(ref["L"])
    assert maxdiff(pn.lse[fin], ref["L"][fin]) <= 1e-3 * max(1.0, float(ref["L"][fin].abs().max()))
    for got, key in zip(new[:3], ("dq", "dk", "dv")):
        assert torch.isfinite(got.float()).all(), key
        assert maxdiff(got, ref[key]) <= gbound(ref[key], dtype), key
    assert maxdiff(new[3], ref["db"]) <= gbound(ref["db"], dtype) * (1 + B)


@pytest.mark.parametrize("B,H,M,N,causal,scale", [
    (4, 12, 512, 512, False, 0.125),    # the metric's smallest size with the reference's own operator: 96 + 96 workgroups side by side
    (4, 3, 512, 512, True, 0.125),      # causal: workgroups of unequal length in both halves, 24 + 24 (the dK/dV count a multiple of eight)
    (2, 3, 300, 560, False, 1.3),       # ragged rows / keys, two idle waves per dQ workgroup, 18 dK/dV workgroups: padded to 24, the padding exits
    (6, 2, 256, 264, True, 1.0),        # B > 4: fp32 slabs + the ordered reduction behind the one launch; 1 / scale a 16-bit value (one selector term)
    (1, 5, 1000, 1096, False, 0.0884),  # forced at B = 1; rows past M inside the last 32-row step of the statistics kernel
    (5, 2, 520, 72, True, 0.125),       # M >> N: dead rows (lse = -inf) -- their statistics are the zero-probability value in both forms
])
def test_dfused64_is_bit_identical_to_the_separate_launches(B, H, M, N, causal, scale):
    """Both dense 64-wide bodies in ONE launch behind bwd_stat2_kernel (attn_bwd_dfused64_kernel, FAT5_V_FUSED64_ON) against the same bodies as separate launches
    with the dQ body's own statistics (FAT5_V_FUSED64_OFF): every output bit for bit (the statistics kernel sums a row in the dQ prologue's order), and against the oracle."""
    from flasht5_amd import _lib
    dtype = torch.bfloat16
    q, k, v, b, do = make_inputs(B, H, M, N, 64, dtype, "1h", seed=7 * M + N + B, strided=True)
    if abs(scale) > 1.0:
        q = (q.float() * 0.5).to(dtype)
    ref = oracle_all(q, k, v, b, do, scale, causal)
    base = _lib.V_QDB64_ON | _lib.V_KV64_ON
    pn, new = _plan(q, k, v, do, b, causal, scale, base | _lib.V_FUSED64_ON)
    po, old = _plan(q, k, v, do, b, causal, scale, base | _lib.V_FUSED64_OFF)
    assert pn.describe()["fused"] == "1" and po.describe()["fused"] == "0"
    assert pn.describe()["dq"] == "64row-batch4" and pn.describe()["dkdv"] == "64key" and pn.bwd_launches() == 1 and po.bwd_launches() == 2
    for x, y, key in zip(new, old, ("dq", "dk", "dv", "db")):
        assert torch.isfinite(x.float()).all(), key
        assert torch.equal(x, y), key
    for got, key in zip(new[:3], ("dq", "dk", "dv")):
        assert maxdiff(got, ref[key]) <= gbound(ref[key], dtype), key
    assert maxdiff(new[3], ref["db"]) <= gbound(ref["db"], dtype) * (1 + B)
    # stage-by-stage calls of the same plan (dQ + dbias, then dK/dV: separate launches whatever the layout says) give the same bits again
    pn.dq.fill_(float("nan")); pn.dk.fill_(float("nan")); pn.dv.fill_(float("nan")); pn.dbias.fill_(float("nan"))
    pn.backward(5); pn.backward(2)
    torch.cuda.synchronize()
    for x, y, key in zip((pn.dq, pn.dk, pn.dv, pn.dbias), old, ("dq", "dk", "dv", "db")):
        assert torch.equal(x, y), key


@pytest.mark.parametrize("B,H,M,N,causal,scale", [
    (4, 3, 512, 512, False, 0.125),     # 1 / scale an fp16 value: the one-term selector kernels
    (4, 2, 512, 512, True, 1.3),        # the reference benchmark's scale: 1 / 1.3 as two fp16 terms
    (6, 2, 300, 560, False, 0.0884),    # B > 4: fp32 slabs; ragged
    (2, 3, 1000, 1096, True, -0.5),     # a negative scale, ragged rows and keys with the mask
    (16, 2, 256, 256, True, 1.0),       # T5's scale, four groups
])
def test_dense_64wide_bodies_in_fp16(B, H, M, N, causal, scale):
    """fp16 through the dense 64-wide backward bodies (second half of round 5): dQ + dBias (qdb64), the 64-key dK/dV body, both in one launch -- against the oracle at the
    fp16 bounds, bit-identical between the one-launch and the separate-launch form, and within a rounding of the older fp16 paths (same terms: dS rounded to fp16 per batch element)."""
    from flasht5_amd import _lib
    dtype = torch.float16
    q, k, v, b, do = make_inputs(B, H, M, N, 64, dtype, "1h", seed=5 * M + N + B, strided=True)
    if abs(scale) > 1.0:
        q = (q.float() * 0.5).to(dtype)
    ref = oracle_all(q, k, v, b, do, scale, causal)
    base = _lib.V_QDB64_ON | _lib.V_KV64_ON
    pn, new = _plan(q, k, v, do, b, causal, scale, base | _lib.V_FUSED64_ON)
    ps, sep = _plan(q, k, v, do, b, causal, scale, base | _lib.V_FUSED64_OFF)
    po, old = _plan(q, k, v, do, b, causal, scale, _lib.V_QDB64_OFF | _lib.V_KV64_OFF)
    assert pn.describe()["dq"] == "64row-batch4" and pn.describe()["dkdv"] == "64key" and pn.describe()["fused"] == "1"
    assert ps.describe()["dq"] == "64row-batch4" and ps.describe()["fused"] == "0" and po.describe()["dq"] != "64row-batch4" and po.describe()["dkdv"] == "32key"
    for x, y, key in zip(new, sep, ("dq", "dk", "dv", "db")):
        assert torch.isfinite(x.float()).all(), key
        assert torch.equal(x, y), key
    for got, key in zip(new[:3], ("dq", "dk", "dv")):
        assert maxdiff(got, ref[key]) <= gbound(ref[key], dtype), key
    assert maxdiff(new[3], ref["db"]) <= gbound(ref["db"], dtype) * (1 + B)
    for i, key in enumerate(("dq", "dk", "dv", "db")):
        assert maxdiff(new[i], old[i]) <= 2.0 ** -8 * max(1.0, float(old[i].float().abs().max())) * (1 + (B if key == "db" else 0)), key
    if causal:
        keep = torch.arange(N, device="cuda")[None, :] <= torch.arange(M, device="cuda")[:, None] + (N - M)
        assert bool((new[3][0][:, ~keep] == 0).all())
