"""GPU parity of SURVEY 8(f) n3's fusions against the ORACLE (fp32 restatements of the reference's op pairs), forward and gradients:

  * `rmsnorm_linear`  -- pre-norm + the stacked projections as ONE library GEMM (round 6: the hand-written GEMM is gone): reference `layer_norm` -> Wq / Wk / Wv
    (modeling_flash_t5.py:304-318, :95-112), `layer_norm` -> wi_0 / wi_1 (:159-160)
  * `linear_residual` -- residual add as the GEMM's epilogue: `hidden_states + self.o(...)` / `+ self.wo(...)` (:316, :162-163)
  * `fused_add_rms_layernorm` (residual add inside the next pre-norm) and `lm_head_cross_entropy` (chunked lm_head -> loss,
    :725-730): VERDICT r2 asked for these two against the oracle rather than against the unfused HIP operators
  * the config-5 step with `fuse_norm_linear=True` against the same step built from the separate operators.

Tolerance of the GEMM outputs: one output rounding plus the 1e-3 of the north star, relative to the tensor's largest entry --
`(1e-3 + u * half_ulp) * max(1, max|ref|)`; u = 2 where the fused kernel rounds differently from the reference (it folds the
norm weight into the projection instead of rounding the normalised activation)."""
import pytest
import torch

import oracle
from attn_helpers import maxdiff

pytestmark = pytest.mark.gpu
HU = {torch.bfloat16: 2.0 ** -8, torch.float16: 2.0 ** -11}


def _b(ref, dtype, u=1.0):
    return (1e-3 + u * HU[dtype]) * max(1.0, ref.float().abs().max().item())


def _mk(M, N, K, dtype, seed):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(M, K, generator=g).to(dtype).cuda()
    gw = (1.0 + 0.2 * torch.randn(K, generator=g)).to(dtype).cuda()
    W = (torch.randn(N, K, generator=g) / K ** 0.5).to(dtype).cuda()
    res = torch.randn(M, N, generator=g).to(dtype).cuda()
    dout = torch.randn(M, N, generator=g).to(dtype).cuda()
    return x, gw, W, res, dout


@pytest.mark.parametrize("M,N,K", [(4096, 2304, 768), (200, 776, 128), (1, 8, 64), (300, 4096, 768), (513, 768, 2048)])
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_rmsnorm_linear_forward_and_gradients_vs_oracle(M, N, K, dtype):
    from flasht5_amd import rmsnorm_linear
    x, gw, W, _, dout = _mk(M, N, K, dtype, M + N + K)
    xs, gs, Ws = (t.detach().clone().requires_grad_() for t in (x, gw, W))
    out = rmsnorm_linear(xs, gs, Ws, 1e-6)
    assert out.shape == (M, N) and out.dtype == dtype
    ref, _ = oracle.rmsnorm_linear_oracle(x, gw, W, 1e-6)
    assert maxdiff(out, ref) <= _b(ref, dtype, 2.0), maxdiff(out, ref)
    # ... and against the reference's own rounding order (normalised activation rounded to the activation dtype first)
    ref2 = oracle.rmsnorm_linear_reference_rounding(x, gw, W, 1e-6)
    assert maxdiff(out, ref2) <= _b(ref2, dtype, 2.0) + HU[dtype] * ref2.abs().max().item()
    dx, dg, dW = torch.autograd.grad(out, (xs, gs, Ws), dout)
    xf, gf, Wf = (t.detach().float().requires_grad_() for t in (x, gw, W))
    rout, _ = oracle.rmsnorm_linear_oracle(xf, gf, Wf, 1e-6)
    rdx, rdg, rdW = torch.autograd.grad(rout, (xf, gf, Wf), dout.float())
    for name, got, want in (("dx", dx, rdx), ("dg", dg, rdg), ("dW", dW, rdW)):
        assert got.dtype == dtype and torch.isfinite(got.float()).all(), name
        assert maxdiff(got, want) <= _b(want, dtype, 3.0), (name, maxdiff(got, want), want.abs().max().item())


@pytest.mark.parametrize("M,N,K", [(4096, 768, 768), (130, 72, 192), (4096, 768, 2048)])
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_linear_residual_forward_and_gradients_vs_oracle(M, N, K, dtype):
    from flasht5_amd import linear_residual
    x, _, W, res, dout = _mk(M, N, K, dtype, 7 * M + N)
    xs, Ws, rs = (t.detach().clone().requires_grad_() for t in (x, W, res))
    out = linear_residual(xs, Ws, rs)
    ref = oracle.linear_residual_oracle(x, W, res)
    assert maxdiff(out, ref) <= _b(ref, dtype, 2.0)  # (two roundings: the product, then the sum -- like the two separate ops)
    # bit-identical to the two separate ops whenever the library GEMM accumulates the same way; at least within one rounding
    sep = res + torch.nn.functional.linear(x, W)
    assert maxdiff(out, sep) <= 2 * HU[dtype] * max(1.0, sep.float().abs().max().item())
    da, dW, dr = torch.autograd.grad(out, (xs, Ws, rs), dout)
    assert torch.equal(dr, dout)
    assert maxdiff(da, dout.float() @ W.float()) <= _b(dout.float() @ W.float(), dtype)
    assert maxdiff(dW, dout.float().t() @ x.float()) <= _b(dout.float().t() @ x.float(), dtype)


def test_rmsnorm_linear_views():
    """(B, S, K) inputs and strided rows are taken as they are"""
    from flasht5_amd import rmsnorm_linear
    x, gw, W, _, _ = _mk(777, 264, 256, torch.bfloat16, 3)
    big = torch.randn(4, 50, 512, device="cuda").bfloat16()
    xv = big[:, :, :256]  # row stride 512, K = 256
    out = rmsnorm_linear(xv, gw, W, 1e-6)
    ref, _ = oracle.rmsnorm_linear_oracle(xv.reshape(-1, 256), gw, W, 1e-6)
    assert out.shape == (4, 50, 264) and maxdiff(out.reshape(-1, 264), ref) <= _b(ref, torch.bfloat16, 2.0)


def test_unsupported_shapes_run_the_separate_ops():
    from flasht5_amd import rmsnorm_linear, linear_residual
    x = torch.randn(10, 100, device="cuda").bfloat16()      # K = 100: not a multiple of 64
    gw = torch.ones(100, device="cuda").bfloat16()
    W = torch.randn(24, 100, device="cuda").bfloat16()
    ref, _ = oracle.rmsnorm_linear_oracle(x, gw, W, 1e-6)
    assert maxdiff(rmsnorm_linear(x, gw, W, 1e-6), ref) <= _b(ref, torch.bfloat16, 2.0)
    r = torch.randn(10, 24, device="cuda").bfloat16()
    ref2 = oracle.linear_residual_oracle(x, W, r)
    assert maxdiff(linear_residual(x, W, r), ref2) <= _b(ref2, torch.bfloat16, 2.0)


def test_rmsnorm_linear_three_weights_one_gemm():
    """(Wq, Wk, Wv) in one call: outputs concatenated, one gradient per weight"""
    from flasht5_amd import rmsnorm_linear
    x, gw, W, _, _ = _mk(384, 3 * 128, 256, torch.bfloat16, 17)
    ws = [w.detach().clone().requires_grad_() for w in W.split(128, 0)]
    xs, gs = x.detach().clone().requires_grad_(), gw.detach().clone().requires_grad_()
    out = rmsnorm_linear(xs, gs, tuple(ws), 1e-6)
    dout = torch.randn_like(out)
    grads = torch.autograd.grad(out, [xs, gs] + ws, dout)
    xf, gf, Wf = (t.detach().float().requires_grad_() for t in (x, gw, W))
    rout, _ = oracle.rmsnorm_linear_oracle(xf, gf, Wf, 1e-6)
    rdx, rdg, rdW = torch.autograd.grad(rout, (xf, gf, Wf), dout.float())
    assert maxdiff(out, rout) <= _b(rout, torch.bfloat16, 2.0)
    assert maxdiff(grads[0], rdx) <= _b(rdx, torch.bfloat16, 3.0) and maxdiff(grads[1], rdg) <= _b(rdg, torch.bfloat16, 3.0)
    assert maxdiff(torch.cat(grads[2:], 0), rdW) <= _b(rdW, torch.bfloat16, 3.0)


@pytest.mark.parametrize("rows,n", [(48, 768), (7, 1024), (33, 200)])
def test_fused_add_rms_layernorm_vs_oracle_fp32(rows, n):
    """(h, y) = (x + r, rmsnorm(x + r) w) and ALL its gradients against oracle.rmsnorm_fwd_oracle / rmsnorm_bwd_oracle on fp32 inputs"""
    from flasht5_amd import fused_add_rms_layernorm
    g = torch.Generator().manual_seed(rows + n)
    x, r = torch.randn(rows, n, generator=g), torch.randn(rows, n, generator=g)
    w = 1.0 + 0.1 * torch.randn(n, generator=g)
    gh, gy = torch.randn(rows, n, generator=g), torch.randn(rows, n, generator=g)
    xs, rs, ws = (t.cuda().requires_grad_() for t in (x, r, w))
    h, y = fused_add_rms_layernorm(xs, rs, ws, 1e-6)
    dx, dr, dw = torch.autograd.grad((h, y), (xs, rs, ws), (gh.cuda(), gy.cuda()))
    h_ref = x + r
    y_ref, rstd = oracle.rmsnorm_fwd_oracle(h_ref, w, 1e-6)
    dh_ref, dw_ref = oracle.rmsnorm_bwd_oracle(gy, h_ref, w, rstd)
    dh_ref = dh_ref + gh  # h feeds the norm AND the residual stream
    h, y, dx, dr, dw = (t.cpu() for t in (h, y, dx, dr, dw))
    assert maxdiff(h, h_ref) == 0.0
    assert maxdiff(y, y_ref) <= 1e-5 * max(1.0, y_ref.abs().max().item())
    assert maxdiff(dx, dh_ref) <= 1e-5 * max(1.0, dh_ref.abs().max().item()) and torch.equal(dx, dr)
    assert maxdiff(dw, dw_ref) <= 1e-5 * max(1.0, dw_ref.abs().max().item())


@pytest.mark.parametrize("rows,V,d,chunk", [(96, 1000, 128, 32), (50, 32128, 64, 16), (8, 520, 256, 256)])
def test_lm_head_cross_entropy_vs_oracle_fp32(rows, V, d, chunk):
    """chunked lm_head -> loss: losses, z-losses and the gradients of hidden / weight against oracle.ce_fwd_oracle / ce_bwd_oracle
    applied to the fp32 logits `hidden @ weight.T` (label smoothing 0.1, z-loss 1e-4, ignored rows)"""
    from flasht5_amd import lm_head_cross_entropy
    g = torch.Generator().manual_seed(rows + V)
    hid = torch.randn(rows, d, generator=g)
    W = torch.randn(V, d, generator=g) / d ** 0.5
    lab = torch.randint(0, V, (rows,), generator=g)
    lab[::7] = -100
    gl = torch.randn(rows, generator=g)
    hs, Ws = hid.cuda().requires_grad_(), W.cuda().requires_grad_()
    losses, z = lm_head_cross_entropy(hs, Ws, lab.cuda(), label_smoothing=0.1, lse_square_scale=1e-4, chunk_rows=chunk)
    dh, dW = torch.autograd.grad(losses, (hs, Ws), gl.cuda())
    losses, z, dh, dW = (t.cpu() for t in (losses, z, dh, dW))
    logits = hid @ W.t()
    l_ref, z_ref, lse = oracle.ce_fwd_oracle(logits, lab, 0.1, 1.0, 1e-4, -100)
    dlog = oracle.ce_bwd_oracle(gl, logits, lse, lab, 0.1, 1.0, 1e-4, -100)
    tol = lambda t: 2e-4 * max(1.0, t.abs().max().item())  # noqa: E731  (fp32 GEMMs and exp / log on both sides)
    assert maxdiff(losses, l_ref) <= tol(l_ref) and maxdiff(z, z_ref) <= tol(z_ref)
    assert maxdiff(dh, dlog @ W) <= tol(dlog @ W)
    assert maxdiff(dW, dlog.t() @ hid) <= tol(dlog.t() @ hid)


def test_cfg5_fused_norm_linear_matches_separate_ops():
    """the config-5 step with the pre-norms inside the projection GEMMs and the residual adds as GEMM epilogues
    (`fuse_norm_linear`): loss within bf16 noise of the step built from the separate operators, gradients likewise"""
    from flasht5_amd import FAT5Config, FAT5ForConditionalGeneration
    import copy
    cfg = FAT5Config(num_layers=2, num_decoder_layers=2)
    torch.manual_seed(13)
    m0 = FAT5ForConditionalGeneration(cfg).cuda().bfloat16()
    cfg1 = copy.copy(cfg)
    cfg1.fuse_norm_linear = True
    m1 = FAT5ForConditionalGeneration(cfg1).cuda().bfloat16()
    m1.load_state_dict(m0.state_dict())
    g = torch.Generator().manual_seed(4)
    ids = torch.randint(0, cfg.vocab_size, (2, 512), generator=g).cuda()
    labels = torch.randint(0, cfg.vocab_size, (2, 256), generator=g).cuda()
    l0, l1 = m0(ids, labels), m1(ids, labels)
    assert abs(l0.item() - l1.item()) <= 2e-3 * abs(l0.item()), (l0.item(), l1.item())
    l0.backward()
    l1.backward()
    for (n, p0), (_, p1) in zip(m0.named_parameters(), m1.named_parameters()):
        assert p1.grad is not None and torch.isfinite(p1.grad.float()).all(), n
        rel = 2.0 ** -3 if "relative_attention_bias" in n else 2.0 ** -4  # (every activation differs in its last bf16 bit between the two formulations)
        assert maxdiff(p1.grad, p0.grad) <= rel * max(p0.grad.float().abs().max().item(), 1e-6), (n, maxdiff(p1.grad, p0.grad))


def test_rmsnorm_linear_return_residual_joins_the_gradients_in_the_kernel():
    """`h + f(norm(h))` with the residual alias handed back by rmsnorm_linear: same forward bits, and the input gradient equals
    dx + dres (the fused kernel adds in fp32 and rounds once: within one ulp of autograd's rounded sum of two rounded terms)"""
    from flasht5_amd import rmsnorm_linear, linear_residual
    g = torch.Generator().manual_seed(21)
    x = torch.randn(512, 768, generator=g).cuda().bfloat16()
    gw = (1 + 0.1 * torch.randn(768, generator=g)).cuda().bfloat16()
    W = (torch.randn(1536, 768, generator=g) / 768 ** 0.5).cuda().bfloat16()
    Wo = (torch.randn(768, 1536, generator=g) / 1536 ** 0.5).cuda().bfloat16()
    dout = torch.randn(512, 768, generator=g).cuda().bfloat16()
    outs = []
    for fused in (False, True):
        xs, gs, Ws, Wos = (t.clone().requires_grad_() for t in (x, gw, W, Wo))
        if fused:
            y, res = rmsnorm_linear(xs, gs, Ws, 1e-6, return_residual=True)
        else:
            y, res = rmsnorm_linear(xs, gs, Ws, 1e-6), xs
        out = linear_residual(torch.nn.functional.gelu(y, approximate="tanh"), Wos, res)
        out.backward(dout)
        outs.append((out.detach(), xs.grad, gs.grad, Ws.grad, Wos.grad))
    a, b = outs
    assert torch.equal(a[0], b[0])
    for u, v in zip(a[2:], b[2:]):
        assert torch.equal(u, v)
    # (autograd rounds dx, then rounds dx + dres; the kernel rounds the fp32 sum once: they differ by the rounding of the larger TERM)
    d = (a[1].float() - b[1].float()).abs()
    term = torch.maximum(torch.maximum(a[1].float().abs(), dout.float().abs()), (a[1].float() - dout.float()).abs())
    assert bool((d <= 2.0 ** -7 * term.clamp_min(1e-2)).all()), float((d / term.clamp_min(1e-2)).max())
    # the alias alone (projection output unused): the gradient passes through
    xs = x.clone().requires_grad_()
    _, res = rmsnorm_linear(xs, gw, W, 1e-6, return_residual=True)
    res.backward(dout)
    assert torch.equal(xs.grad, dout)


def test_native_host_path_equals_python_functions():
    """rmsnorm_linear / linear_residual / gated_act_packed / unpack_heads through the C++ autograd functions (csrc/torch_binding.cpp, the
    eager default) and through the Python classes: the same C-ABI calls in the same order -> identical outputs and gradients"""
    from flasht5_amd import _lib, rmsnorm_linear, linear_residual, gated_act_packed
    from flasht5_amd.fused_linear import RMSNormLinear, LinearResidual
    from flasht5_amd.gated_act import GatedActPacked
    if _lib.native() is None:
        pytest.skip("lib/_fat5_torch.so not built")
    g = torch.Generator().manual_seed(31)
    x = torch.randn(2, 192, 768, generator=g).cuda().bfloat16()
    gw = (1 + 0.1 * torch.randn(768, generator=g)).cuda().bfloat16()
    W0, W1 = ((torch.randn(1024, 768, generator=g) / 768 ** 0.5).cuda().bfloat16() for _ in range(2))
    Wo = (torch.randn(768, 1024, generator=g) / 32).cuda().bfloat16()
    dout = torch.randn(2, 192, 768, generator=g).cuda().bfloat16()

    def run(native):
        leaves = [t.clone().requires_grad_() for t in (x, gw, W0, W1, Wo)]
        xs, gs, w0, w1, wo = leaves
        if native:
            h, res = rmsnorm_linear(xs, gs, (w0, w1), 1e-6, return_residual=True)
            t = gated_act_packed(h, "gelu_tanh")
            out = linear_residual(t, wo, res)
        else:
            h, res = RMSNormLinear.apply(xs, gs, 1e-6, True, w0, w1)
            t = GatedActPacked.apply(h, 0)
            out = LinearResidual.apply(t, wo, res)
        out.backward(dout)
        return [out.detach()] + [t_.grad for t_ in leaves]
    for a, b in zip(run(True), run(False)):
        assert torch.equal(a, b)


def test_fused_block_traces_under_fake_tensors():
    """VERDICT r3 missing #5: the Python autograd functions of the fused projections call `fat5::` custom ops with fake
    implementations (the reference registers every kernel with a fake, flash_attention_v2_bias.py:83-89, :219-226), so a block
    built from them traces: torch.compile(backend="aot_eager") of norm -> three projections -> output projection + residual,
    forward values and every gradient against the eager run of the same functions."""
    from flasht5_amd import rmsnorm_linear, linear_residual
    torch.manual_seed(0)
    dev, dt = "cuda", torch.bfloat16
    x = torch.randn(2, 96, 256, device=dev, dtype=dt)
    g = (1.0 + 0.1 * torch.randn(256, device=dev)).to(dt)
    Ws = [(torch.randn(128, 256, device=dev) * 0.05).to(dt) for _ in range(3)]
    Wo = (torch.randn(256, 384, device=dev) * 0.05).to(dt)

    def block(x, g, w0, w1, w2, wo):
        qkv, res = rmsnorm_linear(x, g, (w0, w1, w2), 1e-6, return_residual=True)
        return linear_residual(torch.tanh(qkv), wo, res)

    def run(fn):
        leaves = [t.detach().clone().requires_grad_() for t in (x, g, *Ws, Wo)]
        out = fn(*leaves)
        grads = torch.autograd.grad(out.float().square().sum(), leaves)
        return [out.detach()] + list(grads)

    ref = run(block)
    got = run(torch.compile(block, backend="aot_eager", fullgraph=True))
    for a, b in zip(ref, got):
        assert torch.isfinite(b.float()).all()
        assert (a.float() - b.float()).abs().max().item() <= 2.0 ** -6 * max(1.0, a.float().abs().max().item())
    # shape-only: the ops under FakeTensorMode
    from torch._subclasses.fake_tensor import FakeTensorMode
    with FakeTensorMode():
        fx = torch.empty(4, 64, 256, device=dev, dtype=dt)
        out = rmsnorm_linear(fx, torch.empty(256, device=dev, dtype=dt), torch.empty(512, 256, device=dev, dtype=dt), 1e-6)
        assert out.shape == (4, 64, 512)


def test_fused_projections_match_the_reference_fixture():
    """rmsnorm_linear / linear_residual against tests/golden/norm_linear.npz -- the reference's own FlashT5LayerNorm -> three nn.Linear and
    `hidden + o(attn)` in bf16 with autograd's gradients: forward within the reference's two intermediate roundings + ours,
    gradients within 2e-2 of their largest entry (bf16 operands in every gradient GEMM)."""
    from golden_io import load
    from flasht5_amd import rmsnorm_linear, linear_residual
    z = load("norm_linear")
    g = {k[5:]: torch.from_numpy(z[k]) for k in z if k.startswith("bf16_")}
    x, gw, W = (g[k].cuda().bfloat16().requires_grad_() for k in ("x", "g", "W"))
    out = rmsnorm_linear(x, gw, tuple(W.split(W.shape[0] // 3, 0)), 1e-6)
    dx, dg, dW = torch.autograd.grad(out, (x, gw, W), g["dqkv"].cuda().bfloat16())
    a, wo, res = (g[k].cuda().bfloat16().requires_grad_() for k in ("a", "wo", "res"))
    y = linear_residual(a, wo, res)
    da, dwo, dres = torch.autograd.grad(y, (a, wo, res), g["dy"].cuda().bfloat16())
    for got, key, rel in ((out, "qkv", 3 * 2.0 ** -8), (y, "y", 3 * 2.0 ** -8), (dx, "dx", 2e-2), (dg, "dg", 2e-2), (dW, "dW", 2e-2),
                          (da, "da", 2e-2), (dwo, "dwo", 2e-2), (dres, "dres", 2.0 ** -8)):
        ref = g[key].cuda()
        assert (got.float() - ref).abs().max().item() <= rel * max(1.0, ref.abs().max().item()), key


@pytest.mark.parametrize("rows,V,d,chunk,dtype", [(96, 1000, 128, 32, torch.float32), (50, 32128, 64, 16, torch.float32), (8, 520, 256, 256, torch.float32),
                                                  (200, 4096, 256, 64, torch.bfloat16)])
def test_lm_head_cross_entropy_mean_vs_oracle(rows, V, d, chunk, dtype):
    """reduction="mean" (round 4: the gradients are formed in the forward pass, no recomputation): the mean loss, the mean z-loss and the
    gradients of hidden / weight -- scaled by a non-trivial upstream gradient -- against oracle.ce_fwd_oracle / ce_bwd_oracle on the
    fp32 logits; bf16: within the rounding of the logits / dlogits that the unfused bf16 sequence has as well."""
    from flasht5_amd import lm_head_cross_entropy
    g = torch.Generator().manual_seed(rows + V)
    hid = torch.randn(rows, d, generator=g).to(dtype)
    W = (torch.randn(V, d, generator=g) / d ** 0.5).to(dtype)
    lab = torch.randint(0, V, (rows,), generator=g)
    lab[::7] = -100
    hs, Ws = hid.cuda().requires_grad_(), W.cuda().requires_grad_()
    loss, z = lm_head_cross_entropy(hs, Ws, lab.cuda(), label_smoothing=0.1, lse_square_scale=1e-4, chunk_rows=chunk, reduction="mean")
    assert loss.dim() == 0 and not z.requires_grad
    dh, dW = torch.autograd.grad(loss * 1.75, (hs, Ws))
    logits = hid.float() @ W.float().t()
    l_ref, z_ref, lse = oracle.ce_fwd_oracle(logits, lab, 0.1, 1.0, 1e-4, -100)
    dlog = oracle.ce_bwd_oracle(torch.full((rows,), 1.75 / rows), logits, lse, lab, 0.1, 1.0, 1e-4, -100)
    rel = 2e-4 if dtype == torch.float32 else 2e-2
    tol = lambda t: rel * max(1e-3 if dtype == torch.float32 else 0.0, t.abs().max().item())  # noqa: E731
    assert abs(loss.item() - l_ref.mean().item()) <= rel * abs(l_ref.mean().item()) and abs(z.item() - z_ref.mean().item()) <= rel * abs(z_ref.mean().item()) + 1e-7
    assert maxdiff(dh.cpu(), dlog @ W.float()) <= tol(dlog @ W.float())
    assert maxdiff(dW.cpu(), dlog.t() @ hid.float()) <= tol(dlog.t() @ hid.float())
