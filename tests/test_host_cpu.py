"""CPU tests: the C-ABI library loads and exports every declared symbol, rejects bad descriptors without touching a
GPU, the host-side bias producers match the oracle / goldens, and the unit sharding logic is sound."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

import oracle
from golden_io import load

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    from flasht5_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        import importlib.util
        spec = importlib.util.spec_from_file_location("fat5_build", os.path.join(ROOT, "flasht5_amd", "build.py"))
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
        mod.build_lib()
    return _lib.load()


def test_exports_match_header(lib):
    from flasht5_amd import _lib
    header = open(os.path.join(ROOT, "include", "fat5.h")).read()
    declared = set(re.findall(r"\b(fat5_[a-z0-9_]+)\s*\(", header))
    declared -= {"fat5_attn_params"}
    assert declared == set(_lib.EXPORTS), declared ^ set(_lib.EXPORTS)
    for name in declared:
        assert hasattr(lib, name), name
    assert lib.fat5_version() == 114
    assert lib.fat5_sizeof_attn_params() == ctypes.sizeof(_lib.AttnParams)


def test_bad_descriptors_are_rejected_without_gpu(lib):
    from flasht5_amd import _lib
    p = _lib.AttnParams()
    p.B, p.H, p.M, p.N, p.D = 1, 1, 16, 16, 48
    p.dtype = _lib.FAT5_BF16
    assert lib.fat5_attn_fwd(ctypes.byref(p), None) == -1
    assert b"head_dim 48" in lib.fat5_last_error()
    p.D, p.dtype = 64, _lib.FAT5_F32
    assert lib.fat5_attn_fwd(ctypes.byref(p), None) == -1
    p.dtype = _lib.FAT5_BF16
    assert lib.fat5_attn_fwd(ctypes.byref(p), None) == -1 and b"null" in lib.fat5_last_error()
    p.bias_mode = _lib.BIAS_RPE1D
    assert lib.fat5_attn_fwd(ctypes.byref(p), None) == -1 and b"rpe1d" in lib.fat5_last_error()
    assert lib.fat5_attn_bwd_workspace_bytes(ctypes.byref(p)) == 0
    p.bias_mode = _lib.BIAS_NONE
    p.B, p.H, p.M, p.N = 4, 12, 512, 512
    assert lib.fat5_attn_bwd_workspace_bytes(ctypes.byref(p)) >= 4 * 12 * 512 * 4
    # backward: every tensor's (b, h) slice must fit a 32-bit buffer descriptor, and the RPE radius the LDS of the dK/dV body
    # (both rejected before anything is launched: fake, aligned, non-null pointers are enough)
    for name in ("q", "k", "v", "o", "lse", "dout", "dq", "dk", "dv", "workspace"):
        setattr(p, name, 0x10000)
    p.workspace_bytes = 1 << 40
    ok = _lib.c_i64x3(12 * 512 * 64, 64, 12 * 64)
    for name in ("q_stride", "k_stride", "v_stride", "o_stride", "do_stride", "dq_stride", "dk_stride", "dv_stride"):
        setattr(p, name, ok)
    p.dk_stride = _lib.c_i64x3(0, 64, 1 << 22)  # 512 rows x 8 MiB: 4 GiB per slice
    assert lib.fat5_attn_bwd(ctypes.byref(p), None) == -1 and b"2 GiB" in lib.fat5_last_error()
    p.dk_stride = ok
    p.bias_mode, p.rpe1d, p.rpe_radius = _lib.BIAS_RPE1D, 0x10000, 2048
    assert lib.fat5_attn_bwd(ctypes.byref(p), None) == -1 and b"LDS" in lib.fat5_last_error()
    p.bias_mode = _lib.BIAS_NONE
    assert lib.fat5_rmsnorm_fwd(None, None, None, None, 4, 8, 8, 8, 1e-6, 0, 0, None) == -1
    assert lib.fat5_ce_fwd(None, None, None, None, None, 4, 8, 8, 0.0, 1.0, 0.0, -100, 0, 0, None) == -1


def test_host_bucket_and_bias_match_oracle():
    from flasht5_amd import positional_encoding as pe
    z = load("rpe_buckets")
    deltas = torch.from_numpy(z["deltas"])
    for key, val in z.items():
        if key.startswith("bucket_"):
            _, bidir, nb, mdist = key.split("_")
            got = pe.relative_position_bucket(deltas, bool(int(bidir)), int(nb), int(mdist))
            assert np.array_equal(got.numpy().astype(np.int32), val), key
    table = torch.from_numpy(z["table_1_256_256"])
    assert torch.equal(pe.compute_bias(table, 256, 256), oracle.compute_bias(table, 256, 256))
    # the clamped generator reproduces the dense bias everywhere
    for bidir, M, N in ((True, 256, 256), (True, 96, 300), (False, 128, 128)):
        r1 = pe.rpe1d_from_table(table, bidir, 32, 128)
        R = pe.rpe_radius(128)
        idx = torch.clamp(torch.arange(N)[None, :] - torch.arange(M)[:, None], -R, R) + R
        assert torch.equal(r1[:, idx].unsqueeze(0), oracle.compute_bias(table, M, N, bidir, 32, 128).float())



def test_rpe_module_randomized_position():
    """`randomized_position=True` (reference positional_encoding.py:79-89): the functional form reproduces the reference's
    bias from the reference's positions (fixture); the module draws sorted distinct positions rooted at 0 from the global
    generator (reproducible under torch.manual_seed) and refuses the 1-D form, which needs a Toeplitz bias."""
    from flasht5_amd import positional_encoding as pe
    z = load("rpe_buckets")
    for b in (0, 1):
        table, ctx, mem, want = (torch.from_numpy(z[f"rand_{k}_{b}"]) for k in ("table", "ctx", "mem", "bias"))
        got = pe.compute_bias(table, len(ctx), len(mem), bool(b), 32, 128, ctx, mem)
        assert torch.equal(got[0], want)
    mod = pe.RelativePositionalEncoding(32, 128, 2, max_sequence_length=512, randomized_position=True)
    torch.manual_seed(5)
    b1 = mod.compute_bias(48, 80)
    torch.manual_seed(5)
    ctx, mem = pe.randomized_positions(512, 48), pe.randomized_positions(512, 80)
    assert ctx[0] == 0 and bool((ctx[1:] > ctx[:-1]).all()) and ctx.max() < 512
    assert torch.equal(b1, oracle.compute_bias(mod.relative_attention_bias.weight.detach(), 48, 80, True, 32, 128, ctx.numpy(), mem.numpy()))
    with pytest.raises(NotImplementedError):
        mod.forward_1d()


def test_rpe_module_mirror_dense_and_1d():
    """`RelativePositionalEncoding` mirror: dense output equals the oracle's builder on the module's own table, the
    1-D generator reproduces it through the Toeplitz rule, and its gradient is the table gradient of the dense form."""
    from flasht5_amd import positional_encoding as pe
    torch.manual_seed(3)
    mod = pe.RelativePositionalEncoding(32, 128, 4, bidirectional=True)
    q = torch.zeros(2, 40, 8)  # the reference's forward reads lengths from dim 1 (:105-106)
    k = torch.zeros(2, 70, 8)
    _, _, _, bias = mod(q, k)
    w = mod.relative_attention_bias.weight.detach()
    assert bias.shape == (1, 4, 40, 70)
    assert torch.equal(bias, oracle.compute_bias(w, 40, 70, True, 32, 128))
    r1, R = mod.forward_1d()
    assert r1.shape == (4, 2 * R + 1) and R == 128
    idx = torch.clamp(torch.arange(70)[None, :] - torch.arange(40)[:, None], -R, R) + R
    assert torch.equal(r1.detach()[:, idx].unsqueeze(0), bias.detach().float())
    # gradient through the generator == gradient through the dense bias
    g = torch.randn(1, 4, 40, 70)
    (r1[:, idx].unsqueeze(0) * g).sum().backward()
    g1 = mod.relative_attention_bias.weight.grad.clone()
    mod.relative_attention_bias.weight.grad = None
    (mod.compute_bias(40, 70) * g).sum().backward()
    assert torch.allclose(g1, mod.relative_attention_bias.weight.grad, atol=1e-5)


def test_shard_units_partition():
    from flasht5_amd.sharding import shard_units, heads_needing_reduction
    for B, H in ((4, 12), (3, 5), (1, 7)):
        for world in (1, 2, 4, 8):
            seen = []
            for r in range(world):
                seen += shard_units(B, H, world, r)
            assert sorted(seen) == sorted((b, h) for b in range(B) for h in range(H))
            sizes = [len(shard_units(B, H, world, r)) for r in range(world)]
            assert max(sizes) - min(sizes) <= 1
    assert heads_needing_reduction(4, 12, 4) == []           # 3 whole heads per rank: no communication
    assert heads_needing_reduction(4, 12, 8) == [1, 4, 7, 10]   # 1.5 heads per rank: every third head is split


def test_fake_impls_trace_without_a_gpu():
    """every custom op has a fake (meta) implementation, so FakeTensor / torch.compile tracing of a model
    that calls the ops works -- shapes, dtypes and devices of the outputs, checked here on fake CUDA tensors (no GPU)."""
    from torch._subclasses.fake_tensor import FakeTensorMode
    import flasht5_amd  # noqa: F401  registers the fat5:: ops
    with FakeTensorMode():
        bf = dict(dtype=torch.bfloat16, device="cuda")
        q, k, v = torch.empty(2, 4, 128, 64, **bf), torch.empty(2, 4, 160, 64, **bf), torch.empty(2, 4, 160, 64, **bf)
        for bias in (None, torch.empty(1, 4, 128, 160, **bf)):
            o, L = torch.ops.fat5.flash_attn_v2_fwd(q, k, v, bias, True, 0.125)
            assert o.shape == q.shape and o.dtype == q.dtype and o.device.type == "cuda"
            assert L.shape == (2, 4, 128) and L.dtype == torch.float32
            dq, dk, dv, ds = torch.ops.fat5.flash_attn_v2_bwd(o, o, q, k, v, bias, L, True, 0.125)
            assert dq.shape == q.shape and dk.shape == k.shape and dv.shape == v.shape
            if bias is not None:
                assert ds.shape == bias.shape and ds.dtype == bias.dtype
        # linear-memory RPE mode and packed batches (the generator / table gradient / no gradient forms)
        r1 = torch.empty(4, 257, dtype=torch.float32, device="cuda")
        o, L = torch.ops.fat5.flash_attn_rpe1d_fwd(q, k, v, r1, 128, False, 0.125)
        assert o.shape == q.shape and L.shape == (2, 4, 128)
        idx = torch.empty(257, dtype=torch.int32, device="cuda")
        for need, bucket, want in ((True, None, (4, 257)), (True, idx, (32, 4)), (False, None, (0,))):
            dq, dk, dv, d1 = torch.ops.fat5.flash_attn_rpe1d_bwd(o, o, q, k, v, r1, L, 128, False, 0.125, need, bucket, 32)
            assert dq.shape == q.shape and dk.shape == k.shape and dv.shape == v.shape
            assert tuple(d1.shape) == want and d1.dtype == torch.float32
        qp, kp = torch.empty(300, 4, 64, **bf), torch.empty(500, 4, 64, **bf)
        cu = torch.empty(4, dtype=torch.int32, device="cuda")
        for rp in (None, r1):
            o2, lse2 = torch.ops.fat5.flash_attn_varlen_fwd(qp, kp, kp, cu, cu, 128, 256, False, 0.125, rp, 128)
            assert o2.shape == qp.shape and o2.dtype == qp.dtype and lse2.shape == (4, 300) and lse2.dtype == torch.float32
            dq, dk, dv, d1 = torch.ops.fat5.flash_attn_varlen_bwd(o2, qp, kp, kp, o2, lse2, cu, cu, 128, 256, False, 0.125, rp, 128, rp is not None)
            assert dq.shape == qp.shape and dk.shape == kp.shape and dv.shape == kp.shape
            assert tuple(d1.shape) == ((4, 257) if rp is not None else (0,))
        x, w = torch.empty(64, 768, **bf), torch.empty(768, **bf)
        y, rstd = torch.ops.fat5.rmsnorm_fwd(x, w, 1e-6)
        assert y.shape == x.shape and y.dtype == x.dtype and rstd.shape == (64,) and rstd.dtype == torch.float32
        dx, dw = torch.ops.fat5.rmsnorm_bwd(y, x, w, rstd, 1e-6)
        assert dx.shape == x.shape and dw.shape == w.shape and dw.dtype == w.dtype
        logits = torch.empty(16, 1000, **bf)
        labels = torch.empty(16, dtype=torch.long, device="cuda")
        losses, z, lse = torch.ops.fat5.cross_entropy_fwd(logits, labels, None, 0.0, 1.0, 1e-4, -100)
        assert losses.shape == (16,) and losses.dtype == torch.float32 and z.shape == (16,) and lse.shape == (16,)
        dl = torch.ops.fat5.cross_entropy_bwd(losses, logits, lse, labels, False, 0.0, 1.0, 1e-4, -100)
        assert dl.shape == logits.shape and dl.dtype == logits.dtype


def test_native_torch_binding_loads_and_exports(lib):
    """lib/_fat5_torch.so (the C++ host path) loads next to libfat5.so and exports its entry points (no GPU: nothing is launched)"""
    from flasht5_amd import _lib
    if not os.path.exists(os.path.join(os.path.dirname(_lib.LIB_PATH), "_fat5_torch.so")):  # (fresh checkout: g++ build, 1-2 min)
        import importlib.util
        spec = importlib.util.spec_from_file_location("fat5_build", os.path.join(ROOT, "flasht5_amd", "build.py"))
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
        mod.build_torch_binding()
        _lib._native = False
    nat = _lib.native()
    assert nat is not None, "lib/_fat5_torch.so missing: python flasht5_amd/build.py"
    for name in ("attn_fwd", "attn_bwd", "bias_apply", "rpe_table_apply", "rpe1d_apply"):
        assert callable(getattr(nat, name))
    import ctypes
    assert nat.sizeof_attn_params() == ctypes.sizeof(_lib.AttnParams)


def test_packed_slice_detection_and_unpack_fallback():
    """host logic of the packed q / k / v path without a GPU: which tensors count as slices of one projection output
    (flash_attention_v2_bias.packed_slices), the layout of the gradient buffer handed out for them, and unpack_heads' backward --
    the packed buffer when the gradients are its slices, one stack otherwise -- against autograd's own select path"""
    from flasht5_amd.flash_attention_v2_bias import packed_slices, empty_packed_like
    from flasht5_amd.attention_module import unpack_heads
    B, S, H, D = 2, 5, 3, 8
    x = torch.randn(B, S, 3 * H * D)
    p5 = x.view(B, S, 3, H, D)
    q, k, v = (p5[:, :, i].permute(0, 2, 1, 3) for i in range(3))
    assert packed_slices((q, k, v)) and not packed_slices((q, v, k)) and not packed_slices((k, v))  # (k, v alone: a row holds three slices)
    assert not packed_slices((q.contiguous(), k.contiguous(), v.contiguous()))
    kv = torch.randn(B, S, 2 * H * D).view(B, S, 2, H, D)
    assert packed_slices((kv[:, :, 0].permute(0, 2, 1, 3), kv[:, :, 1].permute(0, 2, 1, 3)))
    g = empty_packed_like((q, k, v))
    assert all(t.shape == q.shape and t.stride() == q.stride() for t in g) and packed_slices(g)

    # unpack_heads: same views, and both backward paths give the gradient of the select formulation
    xs = x.clone().requires_grad_()
    parts = unpack_heads(xs, 3, H)
    assert all(torch.equal(a, b) and a.stride() == b.stride() for a, b in zip(parts, (q, k, v)))
    w = [torch.randn(B, H, S, D) for _ in range(3)]
    sum((a * b).sum() for a, b in zip(parts, w)).backward()                      # unpacked gradients -> the stack fallback
    xr = x.clone().requires_grad_()
    r5 = xr.view(B, S, 3, H, D)
    sum((r5[:, :, i].permute(0, 2, 1, 3) * w[i]).sum() for i in range(3)).backward()
    assert torch.equal(xs.grad, xr.grad)

    class PackedGrad(torch.autograd.Function):  # stands in for the attention backward: gradients as slices of one buffer
        @staticmethod
        def forward(ctx, a, b, c):
            ctx.save_for_backward(a, b, c)
            return a + 0, b + 0, c + 0

        @staticmethod
        def backward(ctx, ga, gb, gc):
            out = empty_packed_like(ctx.saved_tensors)
            for o, gi in zip(out, (ga, gb, gc)):
                o.copy_(gi)
            return out
    xs2 = x.clone().requires_grad_()
    outs = PackedGrad.apply(*unpack_heads(xs2, 3, H))
    real_stack, calls = torch.stack, []
    torch.stack = lambda *a, **kw: (calls.append(1), real_stack(*a, **kw))[1]
    try:
        sum((a * b).sum() for a, b in zip(outs, w)).backward()
    finally:
        torch.stack = real_stack
    assert not calls and torch.equal(xs2.grad, xr.grad)


def test_graphed_train_step_and_capturable_optimizer_argument_checks():
    from flasht5_amd import GraphedTrainStep, AdamWScale
    lin = torch.nn.Linear(8, 8)
    with pytest.raises(TypeError):
        GraphedTrainStep(lin, torch.optim.SGD(lin.parameters(), lr=0.1))
    opt = AdamWScale(lin.parameters(), lr=1e-3)
    with pytest.raises(RuntimeError):
        opt.graph_advance()  # nothing captured
    opt.init_state()        # CPU parameters: state is created, no device arena
    assert all("exp_avg" in opt.state[p] for p in lin.parameters()) and opt._graph_arena == {}
    with pytest.raises(RuntimeError):
        opt._arena_take(torch.device("cpu"), 64)


def test_dispatch_rules_are_pinned():
    """fat5_attn_describe (host-only): the kernel body every stage of a problem runs.  Each line is a MEASURED decision
    (tools/dispatch_audit.py / tools/attn_time.py on MI355X): the test keeps a later threshold edit from silently
    moving a shape to a slower body."""
    from flasht5_amd import _lib as L
    if L.load().fat5_chip_cus() != 256:  # (the thresholds are rounds of the chip: the pinned lines hold for the 256 CUs they were measured on -- ADVICE r5)
        pytest.skip("dispatch rules are pinned for a 256-CU device (MI355X) or no device at all")
    rpe = dict(bias_mode=L.BIAS_RPE1D, radius=128, need_dbias=True)
    dense = dict(bias_mode=L.BIAS_DENSE, need_dbias=True)
    cases = [
        # the three headline shapes (T5 bias): one launch of both 64-wide backward bodies at S = 512 (round 4); the same up to 1152 workgroups (S = 3072); mixed dK/dV launch above; 64-wide everywhere at 8192
        # (round 6: where the one launch is resident at once -- 96 + 96 workgroups on 256 CUs -- the dQ workgroups form the table gradient's diagonal sums: 24.1 vs 27.3 us)
        (dict(B=4, H=12, M=512, N=512, **rpe), dict(fwd="64row-ksplit", dq="64row", dkdv="64key", fused="1", qdiag="1")),
        (dict(B=2, H=12, M=512, N=512, **rpe), dict(fused="1", qdiag="1")),                                   # (25.9 vs 29.4 us)
        (dict(B=8, H=12, M=512, N=512, **rpe), dict(fused="1", qdiag="0")),                                   # (384 workgroups, two rounds: 48.9 either way -> the dK/dV side keeps them)
        (dict(B=4, H=12, M=1024, N=1024, **rpe), dict(dq="64row", dkdv="64key", fused="1", qdiag="0")),       # (68.2 vs 68.9 us forced)
        (dict(B=2, H=12, M=2048, N=2048), dict(dq="64row", dkdv="64key", fused="1")),
        (dict(B=4, H=12, M=512, N=512, causal=True, **rpe), dict(fwd="64row-ksplit", dq="64row", dkdv="64key", fused="1")),
        (dict(B=16, H=12, M=1024, N=1024, causal=True, **rpe), dict(fwd="64row")),                        # (forward, round-6 audit: 3072 waves -- the 256-row form, 43.2 vs 45.8 us split; rounds 4-5: the split form)   # causal + T5 bias: the table carries the mask, the 64-wide one-launch form (round 4)
        (dict(B=4, H=12, M=1024, N=1024, causal=True, **rpe), dict(dq="64row", dkdv="64key", fused="1")),  # (384 workgroups: an exception until causal launches went longest-first -- 44.6 vs 49.6 us, profiles/r05c_dispatch_audit_H12.log)
        (dict(B=2, H=12, M=1024, N=1024, causal=True), dict(fwd="32row-split")),                           # forward, causal, 384 waves of 64 rows at 1024 keys: the 32-row body (12.5 vs 15.5 us)
        (dict(B=4, H=12, M=1024, N=1024, causal=True), dict(fwd="64row-ksplit")),                          # ... 768 waves: the pipelined body
        (dict(B=8, H=12, M=512, N=512, causal=True, **rpe), dict(dq="32row")),                            # ... the exception holds below 1024 keys
        (dict(B=4, H=12, M=2048, N=2048, causal=True), dict(dq="64row", dkdv="64key", fused="1")),        # plain causal from 2048 keys on: 117.4 vs 127.7 us (same audit)
        (dict(B=2, H=12, M=2048, N=2048, causal=True), dict(dq="64row", dkdv="64key", fused="1")),        # ... 63.5 vs 70.3
        (dict(B=4, H=12, M=512, N=512, causal=True), dict(dq="64row", dkdv="64key", fused="1")),          # causal without bias, up to 512 keys (round 5): the mask rides in the dK/dV half's score MFMAs -- the one-launch 64-wide form (21.7 vs 22.6 us; (16,12,512): 65.7 vs 72.1)
        (dict(B=4, H=12, M=1024, N=1024, causal=True), dict(dq="32row", dkdv="32key")),                   # ... not beyond (45.9 us either way)
        (dict(B=4, H=12, M=2048, N=2048, **rpe), dict(fwd="64row-mixed", dq="64row", dkdv="64key", fused="1")),   # 1.5 64-row waves per SIMD: 256-row and key-split workgroups in one launch
        (dict(B=4, H=12, M=2048, N=2048, variant=L.V_FWD64_MIX_OFF, **rpe), dict(fwd="64row-ksplit")),
        (dict(B=4, H=12, M=3072, N=3072, **rpe), dict(dq="64row", dkdv="64key", fused="1")),     # (1152 workgroups: the last size that takes the one-launch form)
        (dict(B=4, H=12, M=3584, N=3584, **rpe), dict(dq="32row", dkdv="64key", fused="0")),
        (dict(B=16, H=12, M=512, N=512, **rpe), dict(dq="64row", dkdv="64key", fused="1")),
        (dict(B=4, H=12, M=2048, N=512, **rpe), dict(dq="32row", dkdv="32key")),                 # (480 workgroups, not roughly square: not the 64-wide one-launch form)
        (dict(B=4, H=12, M=8192, N=8192, **rpe), dict(fwd="64row", dq="64row", dkdv="64key")),
        (dict(B=4, H=12, M=8192, N=8192), dict(fwd="64row", dq="64row", dkdv="64key")),
        (dict(B=4, H=12, M=4096, N=4096), dict(fwd="64row", dq="64row", dkdv="64key")),
        (dict(B=4, H=12, M=4096, N=4096, **rpe), dict(dq="64row", dkdv="64key")),            # T5 bias: band steps pipelined since round 4
        # causal: diagonal steps are unpipelined in the 64-wide backward bodies
        (dict(B=16, H=12, M=1024, N=1024, causal=True), dict(dq="32row", dkdv="64key")),          # (round-6 audit at B = 16: 74.4 vs 80.2 us for the 32-key body; rounds 2-5: "32key")
        (dict(B=6, H=12, M=1024, N=1024, causal=True), dict(dq="32row", dkdv="64key", fused="0")),  # ... and the PURE 256-key launch on causal problems: (8,12,1024) 41.7 vs 47.4 half-length / 52.7 mixed (profiles/r06_audit_causal_kv.log)
        (dict(B=8, H=12, M=1024, N=1024, causal=True), dict(dq="64row", dkdv="64key", fused="1")),  # plain causal, 1024 keys, whole rounds of workgroups (768; B H = 64: 512 -- 49.9 vs 54.5 us): the one-launch 64-wide form
        (dict(B=4, H=16, M=1536, N=1536, causal=True), dict(fused="0")),                            # ... not beyond 1024 keys (96.3 vs 90.6)
        (dict(B=5, H=12, M=1536, N=1536, causal=True), dict(dkdv="64key")),                         # ... 360 workgroups: 48.1 vs 53.0 (32-key) / 57.4
        (dict(B=16, H=12, M=2048, N=2048, causal=True), dict(dq="32row", dkdv="64key")),         # (round 6: pure 256-key launch since causal launches go longest-first -- profiles/r06_audit_s4096.log; round 5: diagonal steps pipelined without bias too -- 261 vs 269 us; (4,12,2048): 73.7 vs 82.7)
        (dict(B=4, H=12, M=4096, N=4096, causal=True), dict(fwd="64row", dq="32row", dkdv="64key", fused="0")),   # (round-6 audit: 1536 workgroups, separate launches 349.8 vs 368.1 us; forward 110.8 vs 114.2 split)   # (1536 workgroups: one launch 369.3 vs 387.8 us with the longest-first order, profiles/r05d_dispatch_audit_causal_bwd.log)
        (dict(B=4, H=12, M=8192, N=8192, causal=True), dict(dq="32row", dkdv="64key")),
        (dict(B=4, H=12, M=512, N=512, causal=True), dict(fwd="32row-split")),
        (dict(B=8, H=12, M=2048, N=2048, causal=True), dict(fwd="64row", fused="0")),                   # (round-6 audit: 3072 waves, 64.7 vs 66.3 us split; backward 1536 workgroups: 206.1 separate vs 214.7)
        (dict(B=16, H=12, M=1024, N=1024, causal=True), dict(fwd="64row")),                    # (round-6 audit: 43.9-44.9 us either way -- one rule for causal and full: split below 2048 waves)
        (dict(B=4, H=12, M=2048, N=2048, causal=True), dict(fwd="64row-ksplit")),                  # ... 1536 waves
        (dict(B=8, H=12, M=512, N=512, causal=True), dict(dq="32row", dkdv="32key", fused="1")),   # plain causal, 384 workgroups of the 64-wide form = one and a half rounds: the 32-wide one-launch form (28.5 vs 36.9 us, round-6 audit)
        (dict(B=16, H=12, M=512, N=512, causal=True), dict(dq="64row", dkdv="64key", fused="1")),  # ... three rounds: 65.7 vs 72.1 (round 5)
        (dict(B=16, H=12, M=1024, N=4096, causal=True), dict(fwd="64row")),                    # N >= 2M: the mask shortens nothing
        # large batch, short keys: the uneven 1.5-waves-per-SIMD range keeps the 32-row forward; whole rounds do not
        (dict(B=16, H=12, M=512, N=512), dict(fwd="64row-ksplit")),                            # (round-4 audit: the 1.5-waves-per-SIMD exception at <= 512 keys is gone)
        (dict(B=16, H=12, M=1024, N=512), dict(fwd="64row")),
        (dict(B=16, H=12, M=1024, N=1024), dict(fwd="64row", dq="32row", dkdv="64key")),
        # under-filled grids with long streams
        (dict(B=4, H=12, M=4096, N=1024), dict(dq="32row", dkdv="64key")),
        (dict(B=4, H=12, M=1024, N=8192), dict(dq="64row")),
        # dense bias: the 64-row forward (two-tile bias ring, one wave per SIMD) from 4096 keys on (round 4), the 32-wide backward bodies;
        # the batch-shared gradient is formed in-kernel once the staging tensor would be large
        # round 5: dQ + the batch-reduced dbias in one kernel (four batch elements per workgroup), the 64-key dK/dV body with the bias on the matrix pipe -- from 2^25
        # scores per call on (profiles/r05_dispatch_audit_none_dense_H12.log); fp32 slabs + ordered reduction beyond four batch elements
        (dict(B=4, H=12, M=8192, N=8192, **dense), dict(fwd="64row", dq="64row-batch4", dkdv="64key", dbias="dq-kernel")),
        (dict(B=4, H=12, M=2048, N=2048, **dense), dict(fwd="32row", dq="64row-batch4", dkdv="64key", fused="1", dbias="dq-kernel")),   # 384 + 384 workgroups: three rounds in ONE launch (dfused64: 230 vs 270 us)
        (dict(B=16, H=12, M=1024, N=1024, causal=True, **dense), dict(fwd="32row", dq="64row-batch4", dkdv="64key", fused="0", dbias="dq-kernel+partials")),  # six rounds: separate launches (272.6 vs 273.8)
        (dict(B=4, H=12, M=1024, N=1024, **dense), dict(dq="64row-batch4", dkdv="64key", fused="0")),                                          # 1.5 rounds: separate launches (81.1 vs 77.6 us in one)
        (dict(B=16, H=12, M=2048, N=2048, **dense), dict(fwd="64row")),                                  # (324 vs 360 us)
        (dict(B=4, H=12, M=512, N=512, **dense), dict(dq="64row-batch4", dkdv="64key", fused="1", dbias="dq-kernel")),   # the metric's smallest size: 96 + 96 workgroups side by side in one launch (dfused64: 28.4 us; the 32-wide launch + staged dS 45.2)
        (dict(B=2, H=8, M=128, N=128, **dense), dict(dq="64row-batch4", dkdv="64key", fused="1", dbias="dq-kernel")),   # ... down to config 1's shape (16.3 vs 18.6 us)
        (dict(B=4, H=12, M=768, N=768, **dense), dict(dq="64row-batch4", dkdv="64key", fused="0")),                      # 144 + 144 workgroups: two rounds either way -> separate launches (60.5 vs 63.8; the older bodies 78.5)
        (dict(B=4, H=12, M=1536, N=1536, **dense), dict(dq="64row-batch4", dkdv="64key", fused="1")),                    # 288 + 288: three rounds instead of four (164.6 vs 204.5)
        (dict(B=8, H=12, M=256, N=256, **dense), dict(dq="64row-batch4", dkdv="64key", fused="1", dbias="dq-kernel+partials")),
        (dict(B=1, H=12, M=2048, N=2048, **dense), dict(dq="32row", dbias="direct")),                    # nothing to reduce over
        (dict(B=4, H=12, M=1024, N=1024, sm_scale=0.0, **dense), dict(dq="32row", dkdv="32key")),        # a zero scale: 1 / scale does not exist -- the per-element bodies
        (dict(B=4, H=12, M=2048, N=2048, variant=L.V_DBIAS_STAGED, **dense), dict(dq="32row", dbias="staged")),  # the older paths stay selectable
        # off the H = 12 line (round 5: profiles/r05_dispatch_audit_H8_16_32.log -- T5-small's 8 heads, 16 and 32 heads, a (batch, head) count that is no multiple of 8; 1 miss of 36, fixed)
        (dict(B=4, H=8, M=512, N=512, **rpe), dict(fwd="32row-split", dq="64row", dkdv="64key", fused="1")),
        (dict(B=4, H=16, M=2048, N=2048, **rpe), dict(fwd="64row", dq="64row", dkdv="64key", fused="1")),
        (dict(B=3, H=5, M=2048, N=2048, causal=True), dict(dq="64row", dkdv="64key", fused="1")),         # an under-filled chip (240 workgroups), causal without bias: the one-launch form (56.5 vs 70.9 us)
        (dict(B=2, H=32, M=2048, N=2048, causal=True), dict(fwd="64row-ksplit", dkdv="64key", fused="0")),   # (B H = 64 audit of round 6: 2048 waves, causal -- split form 46.2 vs 49.0 us; 1024 workgroups -- separate launches 139.3 vs 146.1)
        (dict(B=2, H=8, M=128, N=128), dict(fwd="32row", fused="1")),                                     # config 1's shape
        # head dims other than 64: the 32-wide bodies
        (dict(B=4, H=6, M=8192, N=8192, D=128), dict(fwd="64row", dq="32row", dkdv="32key")),            # head_dim 128 (round 5): the forward on the pipelined body, one wave per SIMD
        (dict(B=2, H=12, M=1024, N=1024, D=128), dict(fwd="32row-split")),                                     # ... from 768 waves of 64 rows on
        (dict(B=4, H=12, M=1024, N=1024, D=128), dict(fwd="64row")),
        (dict(B=16, H=12, M=1024, N=1024, D=128, causal=True), dict(fwd="64row")),                        # plain causal: diagonal blocks masked inside the pipelined sweep
        (dict(B=16, H=12, M=1024, N=1024, D=128, **dense), dict(fwd="64row")),                            # the d_head 128 rows of the reference benchmark
        # (audit at d_head 128, profiles/r05_dispatch_audit_d128_fwd.log: 512 keys where the 256-row workgroups are one partial round; dense from 512 keys, causal dense from 768 waves)
        (dict(B=8, H=12, M=512, N=512, D=128), dict(fwd="64row")),
        (dict(B=16, H=12, M=512, N=512, D=128), dict(fwd="32row")),
        (dict(B=8, H=12, M=512, N=512, D=128, **dense), dict(fwd="64row")),
        (dict(B=4, H=12, M=512, N=512, D=128, **dense), dict(fwd="64row")),
        (dict(B=2, H=12, M=1024, N=1024, D=128, causal=True, **dense), dict(fwd="32row-split")),
        (dict(B=4, H=12, M=1024, N=1024, D=128, **dense), dict(dq="32row", dkdv="32key", dbias="staged")),   # head_dim 128: the staged dS + reduction up to 512 MB (274 vs 341 us for the batch-inner kernel)
        (dict(B=4, H=12, M=4096, N=4096, D=128, **dense), dict(dbias="inkernel")),
        # forced per call
        (dict(B=4, H=12, M=1024, N=1024, variant=L.V_KV64_ON | L.V_KV64_HALF_ON | L.V_Q64_ON | L.V_FWD64_OFF), dict(fwd="32row", dq="64row", dkdv="64key-half")),
    ]
    for args, want in cases:
        got = L.describe(**args)
        for k, v in want.items():
            assert got[k] == v, (args, k, got)
