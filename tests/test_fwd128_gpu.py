"""GPU parity tests of the head_dim-128 instantiation of the long-sequence forward body (csrc/attn_fwd64.h, round 5: ONE wave per SIMD, O^T in AGPRs behind
asm MFMAs, 32 MFMA gaps per pipelined block, Q / O as whole rows through LDS) -- forced per call with FAT5_V_FWD64_ON at sizes the oracle finishes in seconds.
Reference operator: src/model/ops/flash_attention_v2_bias.py:327-483 (`_fwd_kernel`), benchmarked at d_head 128 by benchmarks/bench_fa2_bias.py."""
import pytest
import torch

import oracle
from attn_helpers import make_inputs, oracle_all, maxdiff, eager_lowprec_errors
from test_attention_gpu import bound, gbound

pytestmark = pytest.mark.gpu
D = 128


def _case(B, H, M, N, dtype, mode, seed):
    q, k, v, _, do = make_inputs(B, H, M, N, D, dtype, None, seed=seed, strided=True)
    if mode != "rpe":
        return q, k, v, do, None, None
    table = torch.randn(32, H, generator=torch.Generator().manual_seed(seed + 100)) * 0.5
    bias = oracle.compute_bias(table, M, N, True, 32, 128).contiguous().cuda()
    return q, k, v, do, table, bias


def _run(q, k, v, do, causal, scale, table, bits):
    from flasht5_amd import flash_attention_v2_bias, flash_attention_v2_rpe, _lib
    from flasht5_amd.flash_attention_v2_bias import _attn_fwd
    from flasht5_amd import positional_encoding as pe
    with _lib.variant(bits):
        leaves = [t.detach().clone().requires_grad_() for t in (q, k, v)]
        if table is None:
            o = flash_attention_v2_bias(leaves[0], leaves[1], leaves[2], None, causal, scale)
        else:
            o = flash_attention_v2_rpe(leaves[0], leaves[1], leaves[2], table.cuda(), True, 32, 128, causal, scale)
        dq, dk, dv = torch.autograd.grad(o, leaves, do)
        rp = pe.rpe1d_from_table(table.cuda(), True, 32, 128) if table is not None else None
        _, L = _attn_fwd(q, k, v, None, rp, 128 if table is not None else 0, causal, scale)
    return {"o": o.detach(), "dq": dq, "dk": dk, "dv": dv, "L": L}


@pytest.mark.parametrize("B,H,M,N,causal,mode,dtype", [
    (1, 2, 256, 256, False, "none", torch.bfloat16),      # one workgroup, one steady-state trip
    (1, 2, 100, 90, False, "rpe", torch.bfloat16),        # shorter than a tile: masked tiles only
    (2, 3, 1024, 1024, False, "none", torch.bfloat16),    # pipelined range + remainder tiles
    (2, 3, 1024, 1024, False, "rpe", torch.bfloat16),     # far-negative range, band, far-positive range
    (1, 2, 1280, 1280, False, "rpe", torch.bfloat16),     # band range cut short by the end of the keys
    (1, 2, 2048, 2048, True, "rpe", torch.bfloat16),      # causal: the mask rides in the bias table
    (1, 2, 1000, 1100, True, "rpe", torch.bfloat16),      # ... cut at a non-zero table offset, ragged rows and keys
    (1, 2, 1100, 1000, True, "rpe", torch.bfloat16),      # ... P < 0: the first 100 rows see no key
    (1, 2, 2048, 2048, True, "none", torch.bfloat16),     # plain causal: diagonal tiles masked (unpipelined), the rest pipelined
    (1, 2, 1000, 1100, False, "rpe", torch.bfloat16),     # ragged M and N (rows past M arrive as zeros, N tail)
    (1, 2, 300, 2500, True, "rpe", torch.bfloat16),       # M << N, bottom-right causal
    (1, 2, 2500, 300, True, "none", torch.bfloat16),      # M >> N: fully masked rows (o = 0, lse = -inf)
    (1, 2, 1536, 1536, False, "rpe", torch.float16),      # fp16: the first tile's row maxima as reference point
    (2, 3, 1024, 1024, False, "none", torch.float16),
    (1, 2, 1000, 1100, True, "rpe", torch.float16),
    (1, 1, 3072, 3072, False, "none", torch.bfloat16),    # several trips of the 4-tile steady-state loop
])
def test_fwd128_matches_oracle(B, H, M, N, causal, mode, dtype):
    from flasht5_amd import _lib
    scale = D ** -0.5
    q, k, v, do, table, bias = _case(B, H, M, N, dtype, mode, seed=M + 5 * N)
    ref = oracle_all(q, k, v, bias, do, scale, causal)
    got = _run(q, k, v, do, causal, scale, table, _lib.V_FWD64_ON)
    old = _run(q, k, v, do, causal, scale, table, _lib.V_FWD64_OFF)
    assert torch.isfinite(got["o"].float()).all()
    assert maxdiff(got["o"], ref["o"]) <= bound(ref["o"], dtype)
    # against the 32-row body: o to one output rounding
    assert maxdiff(got["o"], old["o"]) <= 2.0 ** (-7 if dtype == torch.bfloat16 else -10) * max(1.0, float(ref["o"].abs().max()))
    for key in ("dq", "dk", "dv"):  # the backward consumes the lse this body wrote
        assert torch.isfinite(got[key].float()).all(), key
        assert maxdiff(got[key], ref[key]) <= gbound(ref[key], dtype), key
    # lse: the pipelined blocks sum the probabilities as rounded to 16 bits (see test_fwd64_gpu.py for the model of this bound)
    sc = torch.einsum("bhmd,bhnd->bhmn", q.float(), k.float()) * scale
    if bias is not None:
        sc = sc + bias.float()
    if causal:
        keep = torch.arange(N, device=sc.device)[None, :] <= torch.arange(M, device=sc.device)[:, None] + (N - M)
        sc = sc.masked_fill(~keep, float("-inf"))
    fin = torch.isfinite(ref["L"])
    pr = torch.exp(sc - torch.where(fin, ref["L"], torch.zeros_like(ref["L"]))[..., None])
    sigma = 0.425 * (2.0 ** -8 if dtype == torch.bfloat16 else 2.0 ** -11) * pr.square().sum(-1).sqrt()
    L = got["L"]
    assert torch.equal(torch.isfinite(L), fin) and bool((L[~fin] == float("-inf")).all())
    dl = (L - ref["L"]).abs()[fin]
    allow = (5 * sigma + 1e-4 * ref["L"].abs().clamp(min=1.0))[fin]
    assert bool((dl <= allow).all()), (dl.max().item(), allow.max().item())


@pytest.mark.parametrize("boost,rows,at", [(0.0, "all", 512), (40.0, "all", 1111), (400.0, "all", 512), (1000.0, "even", 768),
                                           (72.0, "all", 1024), (-60.0, "all", 0), (-60.0, "even", 0), (-300.0, "all", 0)])
def test_fwd128_optimistic_softmax_edge_cases(boost, rows, at):
    """no running maximum in the pipelined sweep: scores shifted by `boost` nats from key `at` on -- inside the sweep's range, or beyond it -> the exact second
    pass of the workgroup (rescaling the AGPR accumulators behind hand-placed wait states)"""
    from flasht5_amd import _lib
    B, H, S = 1, 2, 2048
    g = torch.Generator().manual_seed(11)
    q = torch.randn(B, H, S, D, generator=g).bfloat16()
    k = torch.randn(B, H, S, D, generator=g).bfloat16()
    v = torch.randn(B, H, S, D, generator=g).bfloat16()
    q[..., 0] = 4.0
    if rows == "even":
        q[..., 1::2, 0] = 0.0
    k[..., at:, 0] = boost / 4.0
    q, k, v = q.cuda(), k.cuda(), v.cuda()
    do = torch.randn(B, H, S, D, generator=g).bfloat16().cuda()
    sc = 0.25  # (|q.k| over 128 dims has sigma ~ 11: keep the unboosted logits inside the sweep's range)
    got = _run(q, k, v, do, False, sc, None, _lib.V_FWD64_ON)
    ref = oracle_all(q, k, v, None, do, sc, False)
    assert torch.isfinite(got["o"].float()).all()
    assert maxdiff(got["o"], ref["o"]) <= bound(ref["o"], torch.bfloat16)
    lp = eager_lowprec_errors(q, k, v, None, do, sc, False, ref)
    for key in ("dq", "dk", "dv"):
        e = maxdiff(got[key], ref[key])
        assert torch.isfinite(got[key].float()).all(), key
        assert e <= max(gbound(ref[key], torch.bfloat16), 3 * lp[key]), (key, e, lp[key])


def test_fwd128_default_dispatch_and_determinism():
    from flasht5_amd import _lib
    from flasht5_amd.flash_attention_v2_bias import AttentionPlan
    q, k, v, _, do = make_inputs(8, 16, 1024, 1024, D, torch.bfloat16, None, seed=5, strided=True)
    plan = AttentionPlan(q, k, v, do, sm_scale=D ** -0.5, need_dbias=False)
    assert plan.describe()["fwd"] == "64row"  # 2048 waves of 64 rows: two rounds of the chip's 1024 SIMDs
    plan.forward(); a = plan.o.clone(); la = plan.lse.clone()
    plan.forward()
    torch.cuda.synchronize()
    assert torch.equal(a, plan.o) and torch.equal(la, plan.lse)
    small = AttentionPlan(q[:1, :2], k[:1, :2], v[:1, :2], do[:1, :2], sm_scale=D ** -0.5, need_dbias=False)
    assert small.describe()["fwd"] != "64row"


@pytest.mark.parametrize("B,H,M,N,causal,scale", [
    (2, 3, 1024, 1024, False, None),    # trips of three tiles (three ring slots), the bias slot by tile parity
    (1, 2, 256, 256, False, None),      # one workgroup
    (2, 2, 1280, 1280, True, None),     # causal: diagonal tiles through the masked, unpipelined tile (bias read from the staged tile)
    (1, 2, 1000, 1096, False, None),    # ragged rows and a key tail
    (16, 2, 512, 576, False, 1.3),      # the reference benchmark's batch and scale (benchmarks/bench_fa2_bias.py): 16 workgroups share every bias tile
    (1, 2, 300, 2504, True, None),      # M << N, bottom-right causal
    (1, 1, 1664, 1664, False, None),    # 26 tiles: eight trips and a remainder of two
])
def test_fwd128_dense_bias_matches_oracle(B, H, M, N, causal, scale):
    """the reference's own operator -- one (1, H, M, N) bias for the batch (flash_attention_v2_bias.py:228-288) -- at head_dim 128: three ring slots per operand beside
    the two-tile bias ring, the exact-pass flag inside the bias ring"""
    from flasht5_amd import _lib
    from flasht5_amd.flash_attention_v2_bias import AttentionPlan
    dtype = torch.bfloat16
    scale = scale or D ** -0.5
    q, k, v, b, do = make_inputs(B, H, M, N, D, dtype, "1h", seed=M + 3 * N, strided=True)
    if scale > 1.0:
        q = (q.float() * 0.25).to(dtype)
    ref = oracle_all(q, k, v, b, do, scale, causal)
    outs = {}
    for name, bits in (("new", _lib.V_FWD64_ON), ("old", _lib.V_FWD64_OFF)):
        plan = AttentionPlan(q, k, v, do, bias=b, causal=causal, sm_scale=scale, variant=bits)
        assert (plan.describe()["fwd"] == "64row") == (name == "new")
        plan.o.fill_(float("nan")); plan.lse.fill_(float("nan"))
        plan.forward()
        outs[name] = (plan.o.clone(), plan.lse.clone(), [t.clone() for t in plan.backward()])
    o, L, grads = outs["new"]
    assert torch.isfinite(o.float()).all()
    assert maxdiff(o, ref["o"]) <= bound(ref["o"], dtype)
    assert maxdiff(o, outs["old"][0]) <= 2.0 ** -7 * max(1.0, float(ref["o"].abs().max()))
    fin = torch.isfinite(ref["L"])
    assert torch.equal(torch.isfinite(L), fin)
    assert maxdiff(L[fin], ref["L"][fin]) <= 2e-3 * max(1.0, float(ref["L"][fin].abs().max()))
    for got, key in zip(grads[:3], ("dq", "dk", "dv")):  # the backward consumes this lse
        assert maxdiff(got, ref[key]) <= gbound(ref[key], dtype), key


def test_fwd128_dense_exact_second_pass_and_masked_bias():
    """logits beyond the sweep's range (the workgroup's exact second pass: flag in the bias ring, accumulators rescaled in AGPRs) and a bias with finfo.min columns
    (`use_masking`, modeling_flash_t5.py:266-270)"""
    from flasht5_amd import _lib
    from flasht5_amd.flash_attention_v2_bias import AttentionPlan
    B, H, S = 2, 2, 1024
    q, k, v, b, do = make_inputs(B, H, S, S, D, torch.bfloat16, "1h", seed=21, strided=True)
    b = b.clone()
    b[..., 900:] = torch.finfo(torch.bfloat16).min
    b[:, 0, :512, 300:400] = 150.0   # 150 nats: 2^216 -- rows 0..511 of head 0 leave the sweep's range, the others stay inside
    ref = oracle_all(q, k, v, b, do, 0.05, False)
    plan = AttentionPlan(q, k, v, do, bias=b, causal=False, sm_scale=0.05, variant=_lib.V_FWD64_ON)
    plan.forward()
    torch.cuda.synchronize()
    assert torch.isfinite(plan.o.float()).all()
    assert maxdiff(plan.o, ref["o"]) <= bound(ref["o"], torch.bfloat16)
    assert maxdiff(plan.lse, ref["L"]) <= 2e-3 * float(ref["L"].abs().max())


@pytest.mark.parametrize("causal", [True, False])
@pytest.mark.parametrize("Dh,mode", [(128, "dense"), (128, "none"), (64, "dense"), (64, "none")])
def test_large_scale_takes_the_first_tile_reference_point(Dh, mode, causal):
    """sm_scale 1.3 on unit-variance inputs (the reference's benchmark, benchmarks/bench_fa2_bias.py): logits of sigma 10 .. 15 nats.  From sm_scale * sqrt(D) >= 8 on the
    bf16 sweep takes the rows' first-tile maxima as reference point instead of 0 (no second pass for rows beyond 69 nats).  A padded bias (finfo.min columns): part of
    the first tile (causal case), or all of it and half of the second -- left padding: the rows keep reference point 0 (non-causal case).  (No row is left without
    an unmasked key: the fp32 oracle does not normalise such rows -- its exp(x - L) absorbs ln(l) beside 3e38 -- while the kernels, like the reference, return the mean of V.)"""
    from flasht5_amd import _lib
    from flasht5_amd.flash_attention_v2_bias import AttentionPlan
    B, H, S = 2, 2, 1024
    q, k, v, b, do = make_inputs(B, H, S, S, Dh, torch.bfloat16, "1h" if mode == "dense" else None, seed=31, strided=True)
    if b is not None:
        b = b.clone()
        if causal:
            b[:, 1, :, 32:96] = torch.finfo(torch.bfloat16).min
        else:
            b[:, 1, :, :96] = torch.finfo(torch.bfloat16).min
    ref = oracle_all(q, k, v, b, do, 1.3, causal)
    plan = AttentionPlan(q, k, v, do, bias=b, causal=causal, sm_scale=1.3, variant=_lib.V_FWD64_ON)
    assert plan.describe()["fwd"].startswith("64row")
    plan.forward()
    torch.cuda.synchronize()
    assert torch.isfinite(plan.o.float()).all()
    assert maxdiff(plan.o, ref["o"]) <= bound(ref["o"], torch.bfloat16)
    assert maxdiff(plan.lse, ref["L"]) <= 2e-3 * float(ref["L"].abs().max())


def test_fwd128_golden_fixture_through_the_pipelined_dense_body():
    """The reference-generated head_dim-128 fixture (tests/golden/attn_d128_nc_bf16.npz: the reference's eager path on (1,2,64,100,128) with a (1,H,M,N) bias) through the
    pipelined 64-row dense body, forced: its 100 keys per row are no multiple of 8, so the bias goes in as a view of a buffer with a 104-key row pitch
    (16-byte aligned rows: the LDS-DMA condition) -- the same values.  o and lse against the stored reference outputs; the backward consumes this o / lse."""
    import golden_io
    from flasht5_amd import _lib
    from flasht5_amd.flash_attention_v2_bias import AttentionPlan
    g = golden_io.load_attn("attn_d128_nc_bf16")
    assert g["D"] == 128 and g["dtype"] == torch.bfloat16 and not g["causal"]
    q, k, v, do = (g[n].cuda() for n in ("q", "k", "v", "do"))
    M, N = g["M"], g["N"]
    big = torch.zeros(1, g["H"], M, (N + 7) // 8 * 8, dtype=torch.bfloat16, device="cuda")
    big[..., :N] = g["bias"].cuda()
    bias = big[..., :N]
    plan = AttentionPlan(q, k, v, do, bias=bias, causal=False, sm_scale=g["sm_scale"], variant=_lib.V_FWD64_ON)
    assert plan.describe()["fwd"] == "64row"
    plan.o.fill_(float("nan")); plan.lse.fill_(float("nan"))
    plan.forward()
    grads = plan.backward()
    torch.cuda.synchronize()
    o_ref, L_ref = g["o"].cuda(), g["L"].cuda()
    assert torch.isfinite(plan.o.float()).all()
    assert maxdiff(plan.o, o_ref) <= bound(o_ref, torch.bfloat16)
    assert maxdiff(plan.lse, L_ref) <= 2e-3 * max(1.0, float(L_ref.abs().max()))
    for got, key in zip(grads[:3], ("dq", "dk", "dv")):
        ref = g[key].cuda()
        assert maxdiff(got, ref) <= gbound(ref, torch.bfloat16), key
    assert maxdiff(grads[3], g["dbias"].cuda()) <= gbound(g["dbias"].cuda(), torch.bfloat16) * 2
