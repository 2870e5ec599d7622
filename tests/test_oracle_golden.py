"""CPU: the oracle/ restatement reproduces every committed golden vector.

The goldens were produced from the imported reference by tests/golden/make_golden.py
(eager `attn_ref` + autograd, `RelativePositionalEncoding`, the eager LayerNorm / CE modules, and the
reference's Triton kernels run under the Triton CPU interpreter)."""
import numpy as np
import pytest
import torch

import oracle
from golden_io import load, load_attn, ATTN_CASES, TRITON_CASES


def md(a, b):
    return (a.float() - b.float()).abs().max().item()


def test_bucket_known_answers():
    # SURVEY 8(a8) known answers (probed from reference positional_encoding.py:25-71)
    d = np.arange(-200, 200, 25)
    assert oracle.relative_position_bucket(d, True, 32, 128).tolist() == \
        [15, 15, 15, 15, 15, 14, 13, 11, 0, 27, 29, 30, 31, 31, 31, 31]
    assert oracle.relative_position_bucket(d, False, 32, 128).tolist() == \
        [31, 31, 31, 31, 30, 27, 24, 19, 0, 0, 0, 0, 0, 0, 0, 0]


def test_bucket_golden():
    z = load("rpe_buckets")
    deltas = z["deltas"]
    n = 0
    for key, val in z.items():
        if key.startswith("bucket_"):
            _, bidir, nb, mdist = key.split("_")
            mine = oracle.relative_position_bucket(deltas, bool(int(bidir)), int(nb), int(mdist))
            assert np.array_equal(mine.astype(np.int32), val), key
            n += 1
    assert n == 8


@pytest.mark.parametrize("bidir,M,N", [(1, 256, 256), (1, 96, 160), (0, 128, 128)])
def test_bias1d_golden(bidir, M, N):
    z = load("rpe_buckets")
    table = torch.from_numpy(z[f"table_{bidir}_{M}_{N}"])
    b1 = oracle.bias1d_from_table(table, M, N, bool(bidir), 32, 128)
    assert torch.equal(b1, torch.from_numpy(z[f"bias1d_{bidir}_{M}_{N}"]))
    dense = oracle.compute_bias(table, M, N, bool(bidir), 32, 128)
    assert torch.equal(dense[0, :, 0, :], torch.from_numpy(z[f"bias_{bidir}_{M}_{N}_row0"]))
    assert torch.equal(dense[0, :, M - 1, :], torch.from_numpy(z[f"bias_{bidir}_{M}_{N}_rowlast"]))
    assert torch.equal(oracle.toeplitz_from_bias1d(b1, M, N), dense)
    # scatter of diagonal sums == autograd through the dense bias
    g = torch.Generator().manual_seed(5)
    dbias = torch.randn(1, table.shape[1], M, N, generator=g)
    tl = table.clone().requires_grad_()
    oracle.compute_bias(tl, M, N, bool(bidir), 32, 128).backward(dbias)
    d1 = torch.zeros(table.shape[1], M + N - 1)
    idx = (torch.arange(N)[None, :] - torch.arange(M)[:, None]) + (M - 1)
    d1.index_add_(1, idx.reshape(-1), dbias[0].reshape(table.shape[1], -1))
    tg = oracle.table_grad_from_dbias1d(d1, M, N, bool(bidir), 32, 128)
    assert md(tg, tl.grad) < 1e-3


@pytest.mark.parametrize("bidir", [True, False])
def test_randomized_position_bias_golden(bidir):
    """reference `randomized_position` branch (positional_encoding.py:79-89): positions drawn BY the reference (fixture),
    bias rebuilt by the oracle bit for bit"""
    z = load("rpe_buckets")
    b = int(bidir)
    table, ctx, mem, want = (torch.from_numpy(z[f"rand_{k}_{b}"]) for k in ("table", "ctx", "mem", "bias"))
    got = oracle.compute_bias(table, len(ctx), len(mem), bidir, 32, 128, ctx.numpy(), mem.numpy())
    assert torch.equal(got[0], want)
    assert ctx[0] == 0 and mem[0] == 0 and bool((ctx[1:] > ctx[:-1]).all()) and bool((mem[1:] > mem[:-1]).all())


def test_attn_cfg1_golden():
    c = load_attn("attn_cfg1_fp32")
    o, L = oracle.attn_fwd_oracle(c["q"], c["k"], c["v"], c["bias"], c["sm_scale"], c["causal"])
    assert md(o, c["o"]) < 1e-6 and md(L, c["L"]) < 1e-6
    o2 = oracle.attn_ref(c["q"], c["k"], c["v"], c["bias"], c["sm_scale"], causal=c["causal"], upcast=True)
    assert md(o2, c["o"]) < 2e-5


@pytest.mark.parametrize("name", ATTN_CASES)
def test_attn_golden(name):
    c = load_attn(name)
    q, k, v, b, do = c["q"], c["k"], c["v"], c["bias"], c["do"]
    o, L = oracle.attn_fwd_oracle(q, k, v, b, c["sm_scale"], c["causal"])
    assert md(o, c["o"]) < 1e-6 and (L - c["L"]).nan_to_num(0, 0, 0).abs().max() < 1e-6
    dq, dk, dv, ds, dbias = oracle.attn_bwd_oracle(q, k, v, b, o, L, do, c["sm_scale"], c["causal"])
    tol = 4e-4  # fp32 summation order vs autograd
    assert md(dq, c["dq"]) < tol and md(dk, c["dk"]) < tol and md(dv, c["dv"]) < tol
    if b is not None:
        assert dbias.shape == b.shape
        assert md(dbias, c["dbias"]) < 4 * tol
    # row sums of dS vanish (softmax Jacobian) -- property used by the kernels' tests at full size
    assert ds.sum(-1).abs().max() < 2e-3


@pytest.mark.parametrize("name", TRITON_CASES)
def test_oracle_vs_reference_triton_kernels(name):
    """The oracle agrees with the reference's own Triton kernels (interpreter, fp16) within fp16 rounding."""
    c = load_attn(name)
    q, k, v, b, do = c["q"], c["k"], c["v"], c["bias"], c["do"]
    o, L = oracle.attn_fwd_oracle(q, k, v, b, c["sm_scale"], c["causal"])
    dq, dk, dv, _, dbias = oracle.attn_bwd_oracle(q, k, v, b, o, L, do, c["sm_scale"], c["causal"])
    for mine, key in ((o, "o_triton"), (dq, "dq_triton"), (dk, "dk_triton"), (dv, "dv_triton"), (dbias, "dbias_triton")):
        assert md(mine, c[key]) < 2e-3 * max(1.0, mine.abs().max().item()), key
    assert md(L, c["L_triton"]) < 1e-4


@pytest.mark.parametrize("tag", ["a", "b", "c"])
def test_rmsnorm_golden(tag):
    z = load("rmsnorm")
    x, w, dy = (torch.from_numpy(z[f"{n}_{tag}"]) for n in ("x", "w", "dy"))
    y, rstd = oracle.rmsnorm_fwd_oracle(x, w, 1e-6)
    dx, dw = oracle.rmsnorm_bwd_oracle(dy, x, w, rstd)
    assert md(y, torch.from_numpy(z[f"y_{tag}"])) < 1e-5
    assert md(rstd, torch.from_numpy(z[f"rstd_{tag}"])) < 1e-5
    assert md(dx, torch.from_numpy(z[f"dx_{tag}"])) < 1e-5
    assert md(dw, torch.from_numpy(z[f"dw_{tag}"])) < 1e-4
    assert md(oracle.rmsnorm_eager(x, w, 1e-6), torch.from_numpy(z[f"y_eager_{tag}"])) == 0.0
    assert md(dx, torch.from_numpy(z[f"dx_eager_{tag}"])) < 1e-5


@pytest.mark.parametrize("tag", ["a", "b", "c", "d"])
def test_ce_golden(tag):
    z = load("cross_entropy")
    logits = torch.from_numpy(z[f"logits_{tag}"])
    labels = torch.from_numpy(z[f"labels_{tag}"])
    dloss = torch.from_numpy(z[f"dloss_{tag}"])
    smooth, zl = (float(x) for x in z[f"cfg_{tag}"])
    loss, zz, lse = oracle.ce_fwd_oracle(logits, labels, smooth, 1.0, zl, -100)
    dl = oracle.ce_bwd_oracle(dloss, logits, lse, labels, smooth, 1.0, zl, -100)
    s = max(1.0, 50 * zl)
    assert md(loss, torch.from_numpy(z[f"loss_{tag}"])) < 2e-4 * s
    assert md(zz, torch.from_numpy(z[f"z_{tag}"])) < 2e-4 * s
    assert md(lse, torch.from_numpy(z[f"lse_{tag}"])) < 1e-5
    assert md(dl, torch.from_numpy(z[f"dlogits_{tag}"])) < 1e-5 * s
    assert loss[1] == 0 and zz[1] == 0 and dl[1].abs().max() == 0  # ignore_index rows


def test_varlen_oracle_matches_dense():
    g = torch.Generator().manual_seed(0)
    H, D = 2, 64
    cu_q, cu_k = [0, 5, 5, 17], [0, 9, 12, 40]
    q = torch.randn(cu_q[-1], H, D, generator=g)
    k = torch.randn(cu_k[-1], H, D, generator=g)
    v = torch.randn(cu_k[-1], H, D, generator=g)
    out = oracle.attn_varlen_oracle(q, k, v, cu_q, cu_k, 0.125)
    o0, _ = oracle.attn_fwd_oracle(q[0:5].permute(1, 0, 2)[None], k[0:9].permute(1, 0, 2)[None],
                                   v[0:9].permute(1, 0, 2)[None], None, 0.125)
    assert md(out[0:5], o0[0].permute(1, 0, 2)) == 0


def test_varlen_bwd_oracle_matches_autograd():
    """the packed backward restatement equals autograd through the per-sequence eager attention (`attn_ref`)"""
    g = torch.Generator().manual_seed(1)
    H, D = 2, 32
    cu_q, cu_k = [0, 5, 5, 17, 18], [0, 9, 12, 40, 41]
    q = torch.randn(cu_q[-1], H, D, generator=g)
    k = torch.randn(cu_k[-1], H, D, generator=g)
    v = torch.randn(cu_k[-1], H, D, generator=g)
    do = torch.randn(cu_q[-1], H, D, generator=g)
    for causal in (False, True):
        leaves = [t.clone().requires_grad_() for t in (q, k, v)]
        outs = []
        for i in range(len(cu_q) - 1):
            qs, qe, ks, ke = cu_q[i], cu_q[i + 1], cu_k[i], cu_k[i + 1]
            if qe == qs or ke == ks:
                continue
            oi = oracle.attn_ref(leaves[0][qs:qe].permute(1, 0, 2)[None], leaves[1][ks:ke].permute(1, 0, 2)[None],
                                 leaves[2][ks:ke].permute(1, 0, 2)[None], None, 0.25, causal=causal, upcast=True)
            outs.append((oi[0].permute(1, 0, 2) * do[qs:qe]).sum())
        grads = torch.autograd.grad(sum(outs), leaves)
        got = oracle.attn_varlen_bwd_oracle(q, k, v, do, cu_q, cu_k, 0.25, causal)
        for a, b in zip(got, grads):
            assert md(a, b) < 2e-5


ADAMW_CASES = ["fp32", "fp32_wd", "bf16", "bf16_kahan_wd", "fp16_kahan", "fp32_sbf16_wd", "bf16_kahan_sfp16", "bf16_plain", "fp32_plain_wd", "fp16_kahan_plain_sbf16"]


@pytest.mark.parametrize("name", ADAMW_CASES)
def test_adamw_scale_golden(name):
    """the optimizer-step oracle reproduces, bit for bit, what the reference class `AdamWScale` produced on the same seeded tensors
    (three steps; fixture written by tests/golden/make_golden.py gen_adamw, which also asserted it against the live class)"""
    from golden_io import _t
    z = load("adamw_scale")
    cfg = z[f"{name}__cfg"]
    dtype = {0: torch.float32, 1: torch.float16, 2: torch.bfloat16}[int(cfg[0])]
    kahan, wd, lr, b1, b2, eps = bool(cfg[1]), float(cfg[2]), float(cfg[3]), float(cfg[4]), float(cfg[5]), float(cfg[6])
    # round 3: `use_state_dtype` (exp_avg / exp_avg_sq in another 16-bit dtype, reference :101-103) and `correct_bias` (:177)
    sdtype = {-1: dtype, 0: torch.float32, 1: torch.float16, 2: torch.bfloat16}[int(cfg[7])]
    correct = bool(cfg[8])
    for i in range(4):
        p = _t(z[f"{name}__p0_{i}"]).clone()
        assert p.dtype == dtype
        m, v = torch.zeros_like(p, dtype=sdtype), torch.zeros_like(p, dtype=sdtype)
        k = torch.zeros_like(p) if (kahan and dtype != torch.float32) else None
        for step in range(3):
            oracle.adamw_scale_step(p, _t(z[f"{name}__g{step}_{i}"]).clone(), m, v, k, step + 1, lr, b1, b2, wd, eps, correct)
        assert torch.equal(p, _t(z[f"{name}__p_{i}"])) and torch.equal(m, _t(z[f"{name}__m_{i}"])) and torch.equal(v, _t(z[f"{name}__v_{i}"]))
        if k is not None:
            assert torch.equal(k, _t(z[f"{name}__k_{i}"]))


@pytest.mark.parametrize("act", ["gelu_tanh", "relu"])
@pytest.mark.parametrize("tag", ["fp32", "bf16"])
def test_gated_act_golden(act, tag):
    """oracle/gated_act.py against the reference's own FlashT5DenseGatedAct (fixture: tests/golden/make_golden.py::gen_gated_act,
    projections captured by hooks, gradients by autograd) and against torch's tanh GELU"""
    z = load("gated_act")
    g = {k: torch.from_numpy(z[f"{act}_{tag}_{k}"]) for k in ("h0", "h1", "out", "dout", "dh0", "dh1")}
    ulp = 2.0 ** -8 if tag == "bf16" else 2.0 ** -23
    out = oracle.gated_act_oracle(g["h0"], g["h1"], act)
    dh0, dh1 = oracle.gated_act_bwd_oracle(g["dout"], g["h0"], g["h1"], act)
    # the reference rounds act(h0) and the product (and, backward, autograd's intermediate products) to the tensor dtype: a few ulps
    for got, ref, n in ((out, g["out"], 2), (dh0, g["dh0"], 4), (dh1, g["dh1"], 2)):
        err = (got - ref.double()).abs()
        tol = n * ulp * ref.double().abs().clamp_min(1e-3) + (3e-6 if tag == "fp32" else 1e-3)  # (fp32: tanhf's own error where 1 + tanh cancels)
        assert bool((err <= tol).all()), float((err / tol).max())
    if act == "gelu_tanh":
        x = torch.linspace(-12, 12, 4001, dtype=torch.float64)
        a = oracle.gated_act_oracle(x, torch.ones_like(x), act)
        assert float((a - torch.nn.functional.gelu(x, approximate="tanh")).abs().max()) < 1e-12


@pytest.mark.parametrize("tag", ["fp32", "bf16"])
def test_norm_linear_golden(tag):
    """oracle/fused_linear.py as a unit against the reference's own `layer_norm -> Wq / Wk / Wv` and `hidden + o(attn)` sequences
    (fixture: tests/golden/make_golden.py::gen_norm_linear: FlashT5LayerNorm + nn.Linear modules, gradients by autograd).
    fp32: the restatement and its autograd gradients to fp32 rounding; bf16: within the reference's own intermediate roundings
    (normed activation, GEMM output: the oracle keeps fp32 in between, which is what the GPU comparison's tolerance covers)."""
    z = load("norm_linear")
    g = {k[len(tag) + 1:]: torch.from_numpy(z[k]) for k in z if k.startswith(tag + "_")}
    x, gw, W = (g[k].clone().requires_grad_() for k in ("x", "g", "W"))
    out, _ = oracle.rmsnorm_linear_oracle(x, gw, W, 1e-6)
    dx, dg, dW = torch.autograd.grad(out, (x, gw, W), g["dqkv"])
    a, wo, res = (g[k].clone().requires_grad_() for k in ("a", "wo", "res"))
    y = oracle.linear_residual_oracle(a, wo, res)
    da, dwo, dres = torch.autograd.grad(y, (a, wo, res), g["dy"])
    rel = 2e-5 if tag == "fp32" else 3.0 * 2.0 ** -8   # bf16: normed rounded, GEMM output rounded (+ the gradient GEMMs' operands)
    for got, key in ((out, "qkv"), (dx, "dx"), (dg, "dg"), (dW, "dW"), (y, "y"), (da, "da"), (dwo, "dwo"), (dres, "dres")):
        ref = g[key]
        assert (got - ref).abs().max().item() <= rel * max(1.0, ref.abs().max().item()), key
    if tag == "bf16":  # with the reference's intermediate rounding restated: one output rounding
        r2 = oracle.rmsnorm_linear_reference_rounding(g["x"].bfloat16(), g["g"].bfloat16(), g["W"].bfloat16(), 1e-6)
        assert (r2 - g["qkv"]).abs().max().item() <= 2.0 ** -7 * g["qkv"].abs().max().item()
