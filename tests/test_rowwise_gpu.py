"""GPU parity tests of the two bandwidth-bound siblings (RMSNorm, cross-entropy + z-loss) vs the CPU oracle,
the golden fixtures and the reference tests' tolerance (atol 1e-2, test_layer_norm.py:32,42-43,
test_cross_entropy.py:48-49)."""
import pytest
import torch

import oracle
from golden_io import load

pytestmark = pytest.mark.gpu


def md(a, b):
    return (a.float().cpu() - b.float().cpu()).abs().max().item()


@pytest.mark.parametrize("tag", ["a", "b", "c"])
def test_rmsnorm_golden(tag):
    from flasht5_amd import fast_rms_layernorm
    z = load("rmsnorm")
    x, w, dy = (torch.from_numpy(z[f"{n}_{tag}"]).cuda() for n in ("x", "w", "dy"))
    xx, ww = x.clone().requires_grad_(), w.clone().requires_grad_()
    y = fast_rms_layernorm(xx, ww, 1e-6)
    y.backward(dy)
    assert md(y, torch.from_numpy(z[f"y_{tag}"])) < 1e-5
    assert md(xx.grad, torch.from_numpy(z[f"dx_{tag}"])) < 1e-5
    assert md(ww.grad, torch.from_numpy(z[f"dw_{tag}"])) < 1e-4


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16, torch.float16])
@pytest.mark.parametrize("batch,seqlen", [(4, 64), (7, 128), (6, 512)])
@pytest.mark.parametrize("d", [768, 1024])
def test_rmsnorm_reference_shapes(batch, seqlen, d, dtype):
    """test_layer_norm.py: fwd, dx, dw with out.backward(out), atol 1e-2."""
    from flasht5_amd import fast_rms_layernorm
    g = torch.Generator().manual_seed(batch * seqlen + d)
    x = torch.randn(batch, seqlen, d, generator=g).to(dtype)
    w = (1 + 0.1 * torch.randn(d, generator=g)).to(dtype)
    y_ref, rstd = oracle.rmsnorm_fwd_oracle(x, w, 1e-6)
    dx_ref, dw_ref = oracle.rmsnorm_bwd_oracle(y_ref, x, w, rstd)
    xx, ww = x.cuda().requires_grad_(), w.cuda().requires_grad_()
    y = fast_rms_layernorm(xx, ww, 1e-6)
    assert y.shape == x.shape and y.dtype == dtype
    y.backward(y.detach())
    tol = 1e-2 if dtype != torch.float32 else 1e-4
    assert md(y, y_ref) <= tol * max(1.0, y_ref.abs().max().item())
    assert md(xx.grad, dx_ref) <= tol * max(1.0, dx_ref.float().abs().max().item())
    # dw sums rows*1 terms: relative tolerance on the column sums
    assert md(ww.grad, dw_ref) <= (1e-2 if dtype != torch.float32 else 1e-3) * max(1.0, dw_ref.float().abs().max().item())


def test_rmsnorm_mixed_dtype_and_odd_n():
    from flasht5_amd import fast_rms_layernorm
    g = torch.Generator().manual_seed(1)
    x = torch.randn(33, 2048, generator=g).bfloat16()
    w = 1 + 0.1 * torch.randn(2048, generator=g)  # fp32 weight with bf16 activations
    y_ref, rstd = oracle.rmsnorm_fwd_oracle(x, w, 1e-6)
    xx, ww = x.cuda().requires_grad_(), w.cuda().requires_grad_()
    y = fast_rms_layernorm(xx, ww, 1e-6)
    dy = torch.randn(33, 2048, generator=g).bfloat16()
    y.backward(dy.cuda())
    dx_ref, dw_ref = oracle.rmsnorm_bwd_oracle(dy, x, w, rstd)
    assert md(y, y_ref) <= 3e-2 and md(xx.grad, dx_ref) <= 3e-2 and md(ww.grad, dw_ref) <= 1e-3 * dw_ref.abs().max().item() + 1e-3
    # n not a multiple of the vector width: scalar path (forward only is supported for such n)
    x2 = torch.randn(5, 100, generator=g)
    w2 = torch.randn(100, generator=g)
    from flasht5_amd.rms_norm import rmsnorm_fwd
    y2, r2 = rmsnorm_fwd(x2.cuda()[:, :99].contiguous(), w2.cuda()[:99].contiguous(), 1e-6)
    y2_ref, r2_ref = oracle.rmsnorm_fwd_oracle(x2[:, :99], w2[:99], 1e-6)
    assert md(y2, y2_ref) < 1e-5 and md(r2, r2_ref) < 1e-5


@pytest.mark.parametrize("tag", ["a", "b", "c", "d"])
def test_ce_golden(tag):
    from flasht5_amd import cross_entropy_loss
    z = load("cross_entropy")
    logits = torch.from_numpy(z[f"logits_{tag}"]).cuda().requires_grad_()
    labels = torch.from_numpy(z[f"labels_{tag}"]).cuda()
    dloss = torch.from_numpy(z[f"dloss_{tag}"]).cuda()
    smooth, zl = (float(x) for x in z[f"cfg_{tag}"])
    loss, zz = cross_entropy_loss(logits, labels, label_smoothing=smooth, lse_square_scale=zl)
    loss.backward(dloss)
    s = max(1.0, 50 * zl)
    assert md(loss, torch.from_numpy(z[f"loss_{tag}"])) < 2e-4 * s
    assert md(zz, torch.from_numpy(z[f"z_{tag}"])) < 2e-4 * s
    assert md(logits.grad, torch.from_numpy(z[f"dlogits_{tag}"])) < 1e-5 * s
    assert loss[1] == 0 and zz[1] == 0 and logits.grad[1].abs().max() == 0  # ignore_index rows
    assert not zz.requires_grad


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16, torch.float16])
@pytest.mark.parametrize("V", [32768, 32128, 32102])
@pytest.mark.parametrize("zl,smooth", [(0.0, 0.0), (1.0, 0.1), (2.0, 0.0)])
def test_ce_reference_shapes(V, dtype, zl, smooth):
    """test_cross_entropy.py: loss (mean) and dlogits, atol 1e-2."""
    from flasht5_amd import cross_entropy_loss
    g = torch.Generator().manual_seed(V)
    rows = 4 * 64
    logits = torch.randn(rows, V, generator=g).to(dtype)
    labels = torch.randint(0, 4, (rows,), generator=g)
    l_ref, z_ref, lse = oracle.ce_fwd_oracle(logits, labels, smooth, 1.0, zl, -100)
    mean_ref = l_ref.mean()
    dl_ref = oracle.ce_bwd_oracle(torch.full((rows,), float(mean_ref) / rows), logits, lse, labels, smooth, 1.0, zl, -100)
    lg = logits.cuda().requires_grad_()
    out = cross_entropy_loss(lg, labels.cuda(), lse_square_scale=zl, label_smoothing=smooth)[0].mean()
    out.backward(out.detach())
    assert abs(out.item() - mean_ref.item()) <= 1e-2 * max(1.0, abs(mean_ref.item()))
    assert md(lg.grad, dl_ref) <= 1e-2


def test_ce_inplace_backward_and_logit_scale():
    from flasht5_amd import cross_entropy_loss
    g = torch.Generator().manual_seed(3)
    logits = torch.randn(64, 5000, generator=g).bfloat16()
    labels = torch.randint(0, 5000, (64,), generator=g)
    labels[3] = -100
    dloss = torch.randn(64, generator=g)
    l_ref, z_ref, lse = oracle.ce_fwd_oracle(logits, labels, 0.05, 0.5, 1e-3, -100)
    dl_ref = oracle.ce_bwd_oracle(dloss, logits, lse, labels, 0.05, 0.5, 1e-3, -100)
    base = logits.cuda()
    lg = base.clone().requires_grad_()
    work = lg * 1.0  # non-leaf so that the in-place gradient write is legal
    loss, z = cross_entropy_loss(work, labels.cuda(), label_smoothing=0.05, logit_scale=0.5, lse_square_scale=1e-3,
                                 inplace_backward=True)
    assert md(loss, l_ref) < 2e-3 and md(z, z_ref) < 1e-4
    loss.backward(dloss.cuda())
    assert md(lg.grad, dl_ref) <= 1e-2
    assert md(work, dl_ref) <= 1e-2  # gradient landed in the logits buffer itself


def test_ce_precomputed_lse_and_out_of_range_label():
    from flasht5_amd import cross_entropy_loss
    g = torch.Generator().manual_seed(4)
    logits = torch.randn(16, 1000, generator=g)
    labels = torch.randint(0, 1000, (16,), generator=g)
    _, _, lse = oracle.ce_fwd_oracle(logits, labels)
    l1, _ = cross_entropy_loss(logits.cuda(), labels.cuda(), precomputed_lse=lse.cuda())
    l2, _ = cross_entropy_loss(logits.cuda(), labels.cuda())
    assert md(l1, l2) < 1e-5
    labels[0] = 1000  # out of bounds: CE term dropped, smoothing term kept (reference :98-103)
    l_ref, _, _ = oracle.ce_fwd_oracle(logits, labels, 0.1)
    l3, _ = cross_entropy_loss(logits.cuda(), labels.cuda(), label_smoothing=0.1)
    assert md(l3, l_ref) < 1e-4
    with pytest.raises(NotImplementedError):
        cross_entropy_loss(logits.cuda(), labels.cuda(), process_group=object())


def test_module_mirrors_match_the_reference_modules_eager_branches():
    """`FlashT5LayerNorm` / `FlashT5CrossEntropyLoss` (reference modeling_flash_t5.py:40-112) on the HIP operators vs the
    eager branches of the reference modules restated in fp32: forward values and gradients."""
    from flasht5_amd import FlashT5LayerNorm, FlashT5CrossEntropyLoss
    torch.manual_seed(5)
    x = torch.randn(2, 37, 768, device="cuda").bfloat16().requires_grad_()
    ln = FlashT5LayerNorm(768, eps=1e-6).cuda().bfloat16()
    with torch.no_grad():
        ln.weight.copy_(torch.randn(768) * 0.1 + 1.0)
    gy = torch.randn(2, 37, 768, device="cuda").bfloat16()
    y = ln(x)
    gx, gw = torch.autograd.grad(y, (x, ln.weight), gy)
    xf, wf = x.detach().float().requires_grad_(), ln.weight.detach().float().requires_grad_()
    var = xf.pow(2).mean(-1, keepdim=True)
    yr = wf * (xf * torch.rsqrt(var + 1e-6))
    rgx, rgw = torch.autograd.grad(yr, (xf, wf), gy.float())
    assert (y.float() - yr).abs().max() <= 2e-2 * max(1.0, yr.abs().max().item())
    assert (gx.float() - rgx).abs().max() <= 2e-2 * max(1.0, rgx.abs().max().item())
    assert (gw.float() - rgw).abs().max() <= 2e-2 * max(1.0, rgw.abs().max().item())

    V = 1000
    logits = torch.randn(2, 19, V, device="cuda").bfloat16().requires_grad_()
    labels = torch.randint(0, V, (2, 19), device="cuda")
    labels[0, :3] = -100
    ce = FlashT5CrossEntropyLoss(z_loss_factor=1e-4, label_smoothing=0.1)
    loss = ce(logits, labels)
    (gl,) = torch.autograd.grad(loss, logits)
    lf = logits.detach().float().requires_grad_()
    flat, lab = lf.view(-1, V), labels.view(-1)
    per = torch.nn.functional.cross_entropy(flat, lab, label_smoothing=0.1, reduction="none", ignore_index=-100)
    lse = torch.logsumexp(flat, dim=-1)
    per = per + 1e-4 * lse.square() * (lab != -100)
    ref = per.mean()  # the operator path averages over ALL rows (reference :64-68)
    (rgl,) = torch.autograd.grad(ref, lf)
    assert abs(loss.item() - ref.item()) <= 2e-3 * max(1.0, abs(ref.item()))
    assert (gl.float() - rgl).abs().max() <= 1e-2 * max(1e-3, rgl.abs().max().item()) + 1e-6


@pytest.mark.parametrize("rows,d,V,dtype,chunk", [(1000, 256, 4096, torch.bfloat16, 256), (777, 128, 32128, torch.bfloat16, 300),
                                                  (512, 64, 1000, torch.float32, 128), (8192, 768, 32768, torch.bfloat16, None)])
def test_lm_head_cross_entropy_chunked(rows, d, V, dtype, chunk):
    """lm_head GEMM + cross-entropy (+ smoothing, z-loss) in row chunks -- the (rows, V) logits are never
    materialised -- equals the unfused `cross_entropy_loss(hidden @ W^T)` (same kernels, same GEMM inputs): losses, z-losses,
    d hidden, d weight; peak memory stays below one full logits tensor."""
    from flasht5_amd import cross_entropy_loss, lm_head_cross_entropy
    g = torch.Generator().manual_seed(rows + V)
    h = (torch.randn(rows, d, generator=g) * 0.5).to(dtype).cuda().requires_grad_()
    w = (torch.randn(V, d, generator=g) * d ** -0.5).to(dtype).cuda().requires_grad_()
    labels = torch.randint(0, V, (rows,), generator=g)
    labels[::17] = -100
    labels = labels.cuda()
    dl = torch.randn(rows, generator=g).cuda()
    kw = dict(label_smoothing=0.1, lse_square_scale=1e-4)
    torch.cuda.synchronize()
    torch.cuda.reset_peak_memory_stats()
    base0 = torch.cuda.memory_allocated()
    l0, z0 = cross_entropy_loss(h @ w.t(), labels, inplace_backward=True, **kw)  # the reference's leanest form (:247)
    gh0, gw0 = torch.autograd.grad(l0, (h, w), dl)
    torch.cuda.synchronize()
    peak0 = torch.cuda.max_memory_allocated() - base0
    torch.cuda.reset_peak_memory_stats()
    base = torch.cuda.memory_allocated()
    l1, z1 = lm_head_cross_entropy(h, w, labels, chunk_rows=chunk, **kw)
    gh1, gw1 = torch.autograd.grad(l1, (h, w), dl)
    torch.cuda.synchronize()
    peak = torch.cuda.max_memory_allocated() - base
    assert torch.equal(l1, l0) and torch.equal(z1, z0)              # same GEMM rows -> same logits -> same kernel results
    assert md(gh1, gh0) <= 1e-6 + 2.0 ** -8 * gh0.float().abs().max().item() * (dtype != torch.float32)
    # d weight: chunk partial products accumulate in fp32 (the unfused GEMM accumulates the whole K = rows in one pass)
    assert md(gw1, gw0) <= (2.0 ** -7 if dtype != torch.float32 else 1e-5) * max(1.0, gw0.float().abs().max().item())
    assert not z1.requires_grad
    if chunk is None:  # FAT5-base head, 8 x 1024 target tokens: 537 MB of logits never exist; what remains is weight-sized
        assert peak <= peak0 - 0.5 * rows * V * h.element_size(), (peak, peak0)  # (the fp32 dW accumulator + one 64 MB chunk)


@pytest.mark.parametrize("rows,n,dtype,wdtype", [(4096, 768, torch.bfloat16, torch.bfloat16), (37, 1024, torch.float16, torch.float32),
                                                 (5, 100, torch.bfloat16, torch.bfloat16), (64, 768, torch.float32, torch.float32)])
def test_fused_add_rmsnorm_equals_add_then_norm(rows, n, dtype, wdtype):
    """residual add + RMSNorm in one pass: h, y and every gradient bit-identical to `h = x + r;
    y = fast_rms_layernorm(h)` (the residual stream and the normalised tensor both feed later work), and within the
    RMSNorm tolerance of the fp32 oracle composition"""
    from flasht5_amd import fast_rms_layernorm, fused_add_rms_layernorm
    g = torch.Generator().manual_seed(rows + n)
    x = torch.randn(rows, n, generator=g).to(dtype).cuda()
    r = (torch.randn(rows, n, generator=g) * 0.5).to(dtype).cuda()
    w = (1 + 0.1 * torch.randn(n, generator=g)).to(wdtype).cuda()
    gh = torch.randn(rows, n, generator=g).to(dtype).cuda()
    gy = torch.randn(rows, n, generator=g).to(dtype).cuda()

    def run(fused):
        xs, rs, ws = (t.detach().clone().requires_grad_() for t in (x, r, w))
        if fused:
            h, y = fused_add_rms_layernorm(xs, rs, ws, 1e-6)
        else:
            h = xs + rs
            y = fast_rms_layernorm(h, ws, 1e-6)
        return [h.detach(), y.detach()] + list(torch.autograd.grad([h, y], [xs, rs, ws], [gh, gy]))

    for i, (a, b) in enumerate(zip(run(True), run(False))):
        assert a.dtype == b.dtype
        if i == 4 and n % 8:  # dw of the scalar fallback kernel (n not a multiple of 8): LDS float atomics, order not fixed
            assert md(a, b) <= 1e-2 * max(1.0, b.float().abs().max().item())
        else:
            assert torch.equal(a, b), i
    # y alone (the final norm of a stack: nothing flows back through the residual stream)
    xs, rs = x.clone().requires_grad_(), r.clone().requires_grad_()
    _, y1 = fused_add_rms_layernorm(xs, rs, w, 1e-6)
    x2 = x.clone().requires_grad_()
    y2 = fast_rms_layernorm(x2 + r, w, 1e-6)
    assert torch.equal(y1, y2)
    assert torch.equal(torch.autograd.grad(y1, xs, gy)[0], torch.autograd.grad(y2, x2, gy)[0])
    # against the oracle composition in fp32
    want, _ = oracle.rmsnorm_fwd_oracle((x.float() + r.float()).to(dtype).cpu(), w.cpu(), 1e-6)
    assert md(y1, want) <= 1e-2 * max(1.0, want.float().abs().max().item())


def test_native_rmsnorm_path_equals_custom_op_path():
    """the C++ autograd functions (csrc/torch_binding.cpp) and the Python custom-op path launch the same kernels: y, h and the
    gradients are bit-identical (N-d input, vector path)"""
    from flasht5_amd import _lib, fast_rms_layernorm, fused_add_rms_layernorm, Fast_RMS_Layernorm, FusedAddRMSLayernorm
    assert _lib.native() is not None
    g = torch.Generator().manual_seed(4)
    x = torch.randn(3, 50, 768, generator=g).bfloat16().cuda()
    r = torch.randn(3, 50, 768, generator=g).bfloat16().cuda()
    w = (1 + 0.1 * torch.randn(768, generator=g)).bfloat16().cuda()
    gy, gh = torch.randn_like(x), torch.randn_like(x)

    def run(fn, fused):
        xs, rs, ws = (t.clone().requires_grad_() for t in (x, r, w))
        if fused:
            h, y = fn(xs, rs, ws, 1e-6)
            return [h.detach(), y.detach()] + list(torch.autograd.grad([h, y], [xs, rs, ws], [gh, gy]))
        y = fn(xs, ws, 1e-6)
        return [y.detach()] + list(torch.autograd.grad(y, [xs, ws], gy))

    for a, b in zip(run(fast_rms_layernorm, False), run(Fast_RMS_Layernorm.apply, False)):
        assert a.shape == b.shape and torch.equal(a, b)
    for a, b in zip(run(fused_add_rms_layernorm, True), run(FusedAddRMSLayernorm.apply, True)):
        assert a.shape == b.shape and torch.equal(a, b)


@pytest.mark.parametrize("rows,V,dtype,smooth,zl,scale", [
    (37, 32768, torch.bfloat16, 0.1, 1e-4, 1.0),     # the row held in registers between the passes (16 chunks of 256 x 8 columns)
    (5, 32768 + 8, torch.bfloat16, 0.0, 0.0, 1.0),   # one chunk too long: the second pass re-reads the row
    (9, 50264, torch.float16, 0.1, 1e-4, 0.7),       # logit_scale != 1
    (7, 16384, torch.float32, 0.05, 2e-4, 1.0),      # fp32: 16 chunks of 256 x 4
    (11, 1001, torch.bfloat16, 0.1, 1e-4, 1.0),      # not vectorisable: the entry point runs the two launches
    (3, 264, torch.bfloat16, 0.0, 1e-4, 1.3),        # fewer columns than threads
])
def test_ce_forward_and_backward_in_one_launch(rows, V, dtype, smooth, zl, scale):
    """fat5_ce_fwd_bwd (round 4): losses, z-losses, lse and the in-place dlogits are bit-identical to fat5_ce_fwd followed by fat5_ce_bwd --
    same arithmetic, the row read once.  Rows with ignore_index and with out-of-range labels included."""
    from flasht5_amd.cross_entropy_loss import cross_entropy_fwd, cross_entropy_bwd, cross_entropy_fwd_bwd_
    g = torch.Generator().manual_seed(rows * 7 + V)
    logits = (torch.randn(rows, V, generator=g) * 3).to(dtype).cuda()
    labels = torch.randint(0, V, (rows,), generator=g).cuda()
    labels[0] = -100
    if rows > 4:
        labels[3] = V + 5   # out of range (not ignored): cross_entropy_loss.py:100-103
    dl = torch.randn(rows, generator=g).cuda()
    a = logits.clone()
    l0, z0, lse0 = cross_entropy_fwd(a, labels, None, smooth, scale, zl, -100)
    cross_entropy_bwd(dl, a, lse0, labels, True, smooth, scale, zl, -100)
    b = logits.clone()
    l1, z1, lse1 = (torch.empty(rows, device="cuda") for _ in range(3))
    cross_entropy_fwd_bwd_(b, labels, dl, l1, z1, lse1, smooth, scale, zl, -100)
    assert torch.equal(l0, l1) and torch.equal(z0, z1) and torch.equal(lse0, lse1)
    assert torch.equal(a, b)
    # one broadcast upstream gradient (stride 0): what the mean loss passes
    c = logits.clone()
    g1 = torch.full((1,), 1.0 / rows, device="cuda")
    cross_entropy_fwd_bwd_(c, labels, g1.expand(rows), l1, z1, lse1, smooth, scale, zl, -100)
    d = logits.clone()
    cross_entropy_bwd(g1.expand(rows).contiguous(), d, lse0, labels, True, smooth, scale, zl, -100)
    assert torch.equal(c, d)
