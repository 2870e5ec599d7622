"""GPU parity tests of the long-sequence forward body (csrc/attn_fwd64.h: 64 query rows per wave, software-pipelined tile
loop, 4-slot LDS rings) -- forced per call with the variant bit FAT5_V_FWD64_ON (include/fat5.h) at sizes the oracle finishes in seconds; at (4,12,8192,64) the default
dispatch picks it by itself (test_attention_gpu.py::test_cfg3_properties_s8192)."""
import pytest
import torch

import oracle
from attn_helpers import make_inputs, oracle_all, maxdiff, eager_lowprec_errors
from test_attention_gpu import bound, gbound, _rpe_case

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True, params=["rows256", "ksplit"])
def force_fwd64(request):
    """every test of this module runs twice: 256-row workgroups (one wave per 64 rows), and the key-split variant (128-row
    workgroups, two waves per 64 rows merging through LDS)"""
    from flasht5_amd import _lib
    with _lib.variant(_lib.V_FWD64_ON | (_lib.V_FWD64_KSPLIT_ON if request.param == "ksplit" else _lib.V_FWD64_KSPLIT_OFF)):
        yield request.param


def _run(q, k, v, do, causal, scale, table=None, bidir=True, md=128):
    from flasht5_amd import flash_attention_v2_bias, flash_attention_v2_rpe
    leaves = [t.detach().clone().requires_grad_() for t in (q, k, v)]
    if table is None:
        o = flash_attention_v2_bias(leaves[0], leaves[1], leaves[2], None, causal, scale)
    else:
        o = flash_attention_v2_rpe(leaves[0], leaves[1], leaves[2], table.cuda(), bidir, 32, md, causal, scale)
    dq, dk, dv = torch.autograd.grad(o, leaves, do)
    return {"o": o.detach(), "dq": dq, "dk": dk, "dv": dv}


@pytest.mark.parametrize("B,H,M,N,causal,mode,dtype", [
    (1, 2, 256, 256, False, "none", torch.bfloat16),      # one workgroup, one steady-state trip
    (1, 2, 100, 90, False, "rpe", torch.bfloat16),        # shorter than a tile: masked tiles only
    (2, 3, 1024, 1024, False, "none", torch.bfloat16),    # pipelined range + remainder tiles
    (2, 3, 1024, 1024, False, "rpe", torch.bfloat16),     # far-negative range, band, far-positive range
    (1, 2, 1280, 1280, False, "rpe", torch.bfloat16),     # band range cut short by the end of the keys (tail tiles in band mode)
    (1, 2, 2048, 2048, True, "rpe", torch.bfloat16),      # causal: the mask rides in the bias table (P = N - M = 0 inside the band): diagonal tiles run band blocks
    (1, 2, 1000, 1100, True, "rpe", torch.bfloat16),      # ... with the cut at a non-zero table offset (0 < P < R), ragged rows and keys
    (1, 2, 1100, 1000, True, "rpe", torch.bfloat16),      # ... P < 0: the first 100 rows see no key (L = -inf together with a -inf table)
    (1, 2, 2048, 1952, True, "rpe", torch.bfloat16),      # ... P = -96 on whole tiles
    (1, 2, 1100, 1000, True, "rpe", torch.float16),       # ... fp16: dead rows and partly masked first tiles under the first-tile reference point
    (1, 2, 1000, 1100, True, "rpe", torch.float16),
    (1, 2, 2048, 2048, True, "none", torch.bfloat16),
    (1, 2, 1000, 1100, False, "rpe", torch.bfloat16),     # ragged M and N (row clamp, N tail)
    (1, 2, 300, 2500, True, "rpe", torch.bfloat16),       # M << N, bottom-right causal
    (1, 2, 2500, 300, True, "none", torch.bfloat16),      # M >> N: fully masked rows (o = 0, lse = -inf)
    (1, 2, 1536, 1536, False, "rpe", torch.float16),      # fp16 (round 4): the pipelined sweep with the first tile's row maxima as reference point
    (2, 3, 1024, 1024, False, "none", torch.float16),
    (1, 2, 2048, 2048, True, "rpe", torch.float16),       # fp16 causal: rows whose first tile is partly masked
    (1, 2, 1000, 1100, False, "rpe", torch.float16),
    (1, 2, 2500, 300, True, "none", torch.float16),       # fp16, M >> N: rows without any visible key (reference point 0, l = 0)
    (1, 1, 3072, 3072, False, "none", torch.bfloat16),    # several trips of the 4-tile steady-state loop
])
def test_fwd64_matches_oracle(B, H, M, N, causal, mode, dtype):
    scale = 0.125
    if mode == "rpe":
        q, k, v, do, table, bias = _rpe_case(B, H, M, N, dtype, causal, True, 128, seed=M + 5 * N)
    else:
        q, k, v, _, do = make_inputs(B, H, M, N, 64, dtype, None, seed=M + 5 * N, strided=True)
        table, bias = None, None
    ref = oracle_all(q, k, v, bias, do, scale, causal)
    got = _run(q, k, v, do, causal, scale, table)
    assert torch.isfinite(got["o"].float()).all()
    assert maxdiff(got["o"], ref["o"]) <= bound(ref["o"], dtype)
    # the backward consumes the lse this body wrote
    for key in ("dq", "dk", "dv"):
        assert torch.isfinite(got[key].float()).all(), key
        assert maxdiff(got[key], ref[key]) <= gbound(ref[key], dtype), key
    # the lse itself against the oracle's: the pipelined blocks sum the probabilities as rounded to 16 bits (attn_fwd64.h), i.e.
    # ln l is off by sum_i eps_i p_i / sum p with independent relative roundings eps_i: round-to-nearest to 8 significant bits (11 in
    # fp16) is uniform within half a spacing, 2^-8 / mantissa relative; over log-uniform mantissas its rms is 2^-8 / sqrt(3) * 0.736
    # = 0.425 * 2^-8 (round 3: the former 2^-9 / sqrt(3) understated it by 1.47x and only held while half the tiles -- baseline and band
    # tiles -- summed unrounded probabilities).  5 sigma per row + fp32 evaluation noise
    from flasht5_amd.flash_attention_v2_bias import _attn_fwd
    from flasht5_amd import positional_encoding as pe
    rp = pe.rpe1d_from_table(table.cuda(), True, 32, 128) if table is not None else None
    _, L = _attn_fwd(q, k, v, None, rp, 128 if table is not None else 0, causal, scale)
    sc = torch.einsum("bhmd,bhnd->bhmn", q.float(), k.float()) * scale
    if bias is not None:
        sc = sc + bias.float()
    if causal:
        keep = torch.arange(N, device=sc.device)[None, :] <= torch.arange(M, device=sc.device)[:, None] + (N - M)
        sc = sc.masked_fill(~keep, float("-inf"))
    fin = torch.isfinite(ref["L"])
    pr = torch.exp(sc - torch.where(fin, ref["L"], torch.zeros_like(ref["L"]))[..., None])
    sigma = 0.425 * (2.0 ** -8 if dtype == torch.bfloat16 else 2.0 ** -11) * pr.square().sum(-1).sqrt()
    assert torch.equal(torch.isfinite(L), fin) and bool((L[~fin] == float("-inf")).all())
    dl = (L - ref["L"]).abs()[fin]
    allow = (5 * sigma + 1e-4 * ref["L"].abs().clamp(min=1.0))[fin]
    assert bool((dl <= allow).all()), (dl.max().item(), allow.max().item())


@pytest.mark.parametrize("boost,rows,at", [(0.0, "all", 512), (40.0, "all", 512), (40.0, "all", 1111), (400.0, "all", 512),
                                           (1000.0, "even", 768), (90.0, "all", 320), (66.0, "all", 1024), (72.0, "all", 1024),
                                           (-20.0, "all", 0), (-60.0, "all", 0), (-60.0, "even", 0), (-300.0, "all", 0)])
def test_fwd64_optimistic_softmax_edge_cases(boost, rows, at):
    """The pipelined sweep keeps NO running maximum (reference point 0 for every row: attn_fwd64.h).  Scores shifted by `boost`
    nats from key `at` on: inside the sweep's range (0, 40, -20 nats; 66 nats = 2^95: just inside), beyond it -> exact second
    pass of the workgroup: row sums at or above 2^100 (72 nats = 2^104, 90, 400 and 1000 nats -- the last two overflow fp32; "even": only
    every other row leaves the range) or below 2^-40 (-60 / -300 nats on every key: flushed probabilities)."""
    B, H, S, D = 1, 2, 2048, 64
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
    got = _run(q, k, v, do, False, 1.0)
    ref = oracle_all(q, k, v, None, do, 1.0, False)
    assert torch.isfinite(got["o"].float()).all()
    assert maxdiff(got["o"], ref["o"]) <= bound(ref["o"], torch.bfloat16)
    lp = eager_lowprec_errors(q, k, v, None, do, 1.0, False, ref)
    for key in ("dq", "dk", "dv"):
        e = maxdiff(got[key], ref[key])
        assert torch.isfinite(got[key].float()).all(), key
        assert e <= max(gbound(ref[key], torch.bfloat16), 3 * lp[key]), (key, e, lp[key])


def test_fwd64_agrees_with_32row_body(monkeypatch):
    """Both forward bodies on the same inputs: o equal to one output rounding, lse to the rounding of the summed probabilities."""
    from flasht5_amd.flash_attention_v2_bias import AttentionPlan
    from flasht5_amd import positional_encoding as pe
    q, k, v, _, do = make_inputs(2, 4, 2048, 2048, 64, torch.bfloat16, None, seed=3, strided=True)
    table = (torch.randn(32, 4, generator=torch.Generator().manual_seed(1)) * 0.5).cuda()
    plan = AttentionPlan(q, k, v, do, sm_scale=0.125, need_dbias=False, rpe1d=pe.rpe1d_from_table(table, True, 32, 128), radius=128)
    outs = []
    from flasht5_amd import _lib
    for f in (_lib.V_FWD64_OFF, _lib.V_FWD64_ON):
        plan.set_variant(f)
        plan.forward()
        torch.cuda.synchronize()
        outs.append((plan.o.float().clone(), plan.lse.clone()))
    assert (outs[0][0] - outs[1][0]).abs().max().item() <= 2.0 ** -7 * max(1.0, outs[0][0].abs().max().item())
    # lse: the pipelined blocks of the 64-row body sum the probabilities AS ROUNDED for the P.V product (row sums on the matrix pipe,
    # attn_fwd64.h), the 32-row body the unrounded ones: the sums differ by sum_i eps_i p_i with |eps_i| <= 2^-9 (independent
    # roundings; rms 0.425 * 2^-8, see test_fwd64_matches_oracle), i.e. ln l by about 0.425 * 2^-8 * sqrt(sum p^2) / sum p per row;
    # allow 5 sigma on top of the fp32 ordering noise
    bias = oracle.compute_bias(table.cpu(), 2048, 2048, True, 32, 128).cuda().float()
    s = torch.einsum("bhmd,bhnd->bhmn", q.float(), k.float()) * 0.125 + bias
    p = torch.softmax(s, dim=-1)
    sigma = 0.425 * 2.0 ** -8 * p.square().sum(-1).sqrt()
    dl = (outs[0][1] - outs[1][1]).abs()
    assert bool((dl <= 5 * sigma + 2e-5).all()), (dl.max().item(), sigma.max().item())


def test_fwd64_deterministic():
    q, k, v, _, do = make_inputs(1, 4, 2048, 2048, 64, torch.bfloat16, None, seed=9, strided=True)
    a = _run(q, k, v, do, False, 0.125)
    b = _run(q, k, v, do, False, 0.125)
    assert torch.equal(a["o"], b["o"])


@pytest.mark.parametrize("B,H,M,N,causal,kind", [
    (2, 3, 1024, 1024, False, "1h"),     # the model's bias: one tile ring shared by the batch
    (2, 2, 512, 768, False, "bh"),       # per-batch bias
    (1, 2, 1280, 1280, True, "1h"),      # causal: diagonal tiles masked (exact, unpipelined), the rest pipelined
    (2, 2, 1000, 1096, False, "11"),     # ragged M (rows past M read as zeros), N a multiple of 8 but not of 64: a masked tail tile
    (1, 2, 300, 2048, True, "b1"),       # M << N, bottom-right causal, head-broadcast bias
    (1, 2, 2048, 256, True, "1h"),       # M >> N: fully masked rows
    (1, 1, 3072, 3072, False, "1h"),     # several trips of the 4-tile loop: the two-tile bias ring turns over many times
])
def test_fwd64_dense_bias_matches_oracle(B, H, M, N, causal, kind, force_fwd64):
    """Round 4: the dense-bias instantiation of the 64-row body (bias tiles by LDS-DMA into a two-tile ring, the bias words as the
    addend of the exponent FMA inside the pipelined stream; reference kernel bias path flash_attention_v2_bias.py:440-443) forced on
    shapes the oracle finishes in seconds: o, lse and -- through the backward that consumes this lse -- all four gradients."""
    if force_fwd64 == "ksplit":
        pytest.skip("dense bias: 256-row workgroups only")
    from flasht5_amd import _lib
    assert _lib.describe(B=B, H=H, M=M, N=N, bias_mode=_lib.BIAS_DENSE, variant=_lib.V_FWD64_ON)["fwd"] == "64row"
    dtype = torch.bfloat16
    q, k, v, bias, do = make_inputs(B, H, M, N, 64, dtype, kind, seed=7 * M + N, strided=True)
    from attn_helpers import run_dense
    ref = oracle_all(q, k, v, bias, do, 0.125, causal)
    got = run_dense(q, k, v, bias, do, 0.125, causal)
    assert torch.isfinite(got["o"].float()).all()
    assert maxdiff(got["o"], ref["o"]) <= bound(ref["o"], dtype)
    fin = torch.isfinite(ref["L"])
    assert torch.equal(torch.isfinite(got["L"]), fin)
    assert maxdiff(got["L"][fin], ref["L"][fin]) <= 2e-3 * max(1.0, float(ref["L"][fin].abs().max()))
    for key in ("dq", "dk", "dv", "db"):
        assert torch.isfinite(got[key].float()).all(), key
        assert maxdiff(got[key], ref[key]) <= gbound(ref[key], dtype), key


def test_fwd64_dense_bias_masking_values_and_edge_rows(force_fwd64):
    """a bias holding finfo.min (the reference's `use_masking`, modeling_flash_t5.py:266-270) on half of the keys of every row and on ALL
    keys of some rows: masked keys get p = 0 inside the pipelined sweep (the product overflows to -inf), a fully masked row has
    l = 0 there and sends its workgroup through the exact pass, which clamps like the 32-row body: same results as that body"""
    if force_fwd64 == "ksplit":
        pytest.skip("dense bias: 256-row workgroups only")
    from flasht5_amd import _lib
    from flasht5_amd.flash_attention_v2_bias import _attn_fwd
    B, H, S = 1, 2, 1024
    q, k, v, bias, _ = make_inputs(B, H, S, S, 64, torch.bfloat16, "1h", seed=5, strided=True)
    bias = bias.clone()
    bias[..., S // 2:] = torch.finfo(torch.bfloat16).min
    bias[:, :, 7::64, :] = torch.finfo(torch.bfloat16).min
    outs = []
    for bits in (_lib.V_FWD64_ON | _lib.V_FWD64_KSPLIT_OFF, _lib.V_FWD64_OFF):
        with _lib.variant(bits):
            o, L = _attn_fwd(q, k, v, bias, None, 0, False, 0.125)
        outs.append((o.float(), L))
    assert torch.isfinite(outs[0][0]).all()
    assert maxdiff(outs[0][0], outs[1][0]) <= 2.0 ** -7 * max(1.0, float(outs[1][0].abs().max()))
    assert maxdiff(outs[0][1], outs[1][1]) <= 1e-3 * max(1.0, float(outs[1][1].abs().max()))


@pytest.mark.parametrize("boost,at", [(0.0, 512), (6.0, 512), (9.0, 1111), (14.0, 512), (30.0, 768), (200.0, 320), (-20.0, 64), (-60.0, 64)])
def test_fwd64_fp16_reference_point_edge_cases(boost, at):
    """fp16 sweep (round 4): reference point = the row's maximum over its first tile.  Scores shifted by `boost` nats from key `at`
    on: within fp16's range above the reference (6, 9 nats), beyond it (14 nats ~ 2^20, 30, 200: probabilities overflow -> the
    workgroup's exact second pass), far below it (-20, -60 nats from the second tile on: flushed, negligible beside the first tile)."""
    B, H, S, D = 1, 2, 2048, 64
    g = torch.Generator().manual_seed(13)
    q = torch.randn(B, H, S, D, generator=g).half()
    k = torch.randn(B, H, S, D, generator=g).half()
    v = torch.randn(B, H, S, D, generator=g).half()
    q[..., 0] = 4.0
    k[..., at:, 0] = boost / 4.0
    q, k, v = q.cuda(), k.cuda(), v.cuda()
    do = torch.randn(B, H, S, D, generator=g).half().cuda()
    got = _run(q, k, v, do, False, 1.0)
    ref = oracle_all(q, k, v, None, do, 1.0, False)
    assert torch.isfinite(got["o"].float()).all()
    assert maxdiff(got["o"], ref["o"]) <= bound(ref["o"], torch.float16)
    lp = eager_lowprec_errors(q, k, v, None, do, 1.0, False, ref)
    for key in ("dq", "dk", "dv"):  # (the reference's rule for the gradients: keys of size 50 in fp16 -- the eager low-precision path is the yardstick)
        assert torch.isfinite(got[key].float()).all(), key
        e = maxdiff(got[key], ref[key])
        assert e <= max(gbound(ref[key], torch.float16), 3 * lp[key]), (key, e, lp[key])
