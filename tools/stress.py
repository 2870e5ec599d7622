"""race screen: the same problem many times, every output compared bit for bit with the first run (developer tool).
LDS-DMA ordering mistakes show up as rare wrong tiles under load, not as a failing unit test."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from attn_helpers import make_inputs
from flasht5_amd.flash_attention_v2_bias import AttentionPlan
from flasht5_amd import positional_encoding as pe
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
bad = 0
for S, D, H in ((512, 64, 12), (2048, 64, 12), (777, 128, 3), (1024, 32, 6)):
    for mode in ("none", "rpe", "dense"):
        for causal in (False, True):
            q, k, v, _, do = make_inputs(4, H, S, S, D, torch.bfloat16, None, seed=S + D, strided=True)
            kw = {}
            table = (torch.randn(32, H, generator=torch.Generator().manual_seed(1)) * 0.5).cuda()
            if mode == "rpe":
                kw = dict(rpe1d=pe.rpe1d_from_table(table), radius=128, rpe_bucket=pe.bucket_index32(128, True, 32, 128, "cuda"), num_buckets=32)
            elif mode == "dense":
                kw = dict(bias=pe.compute_bias(table, S, S).to(torch.bfloat16).contiguous())
            plan = AttentionPlan(q, k, v, do, sm_scale=D ** -0.5, causal=causal, **kw)
            plan.forward(); plan.backward(); torch.cuda.synchronize()
            ref = [t.clone() for t in (plan.o, plan.lse, plan.dq, plan.dk, plan.dv)] + ([plan.dbias.clone()] if plan.dbias is not None else [])
            n_bad = 0
            for i in range(reps):
                plan.forward(); plan.backward()
                if i % 25 == 24 or i == reps - 1:
                    torch.cuda.synchronize()
                    cur = [plan.o, plan.lse, plan.dq, plan.dk, plan.dv] + ([plan.dbias] if plan.dbias is not None else [])
                    n_bad += sum(0 if torch.equal(a, b) else 1 for a, b in zip(ref, cur))
            print(f"S={S} D={D} {mode:5s} causal={int(causal)}: {'OK' if n_bad == 0 else 'MISMATCH x%d' % n_bad}", flush=True)
            bad += n_bad
# the 64-wide pipelined bodies (attn_fwd64.h, attn_bwd64.h): forced at a mid size (many workgroups per CU in flight, ragged
# ends), then at the size where the dispatch picks them by itself
for S, M, reps2, force in ((1536, 1536, reps, "1"), (1000, 1100, reps, "1"), (1536, 1536, reps, "split"), (900, 1300, reps, "split"),
                           (8192, 8192, max(3, reps // 30), "-1")):
    for mode in ("none", "rpe"):
        for causal in (False, True):
            from flasht5_amd import _lib
            on = _lib.V_FWD64_ON | _lib.V_KV64_ON | _lib.V_Q64_ON
            # "split": two waves per 64-row block (forward) / per 64-key block (dK/dV), halves merged through LDS
            _lib.set_variant({"1": on | _lib.V_FWD64_KSPLIT_OFF | _lib.V_KV64_HALF_OFF,
                              "split": on | _lib.V_FWD64_KSPLIT_ON | _lib.V_KV64_HALF_ON, "-1": 0}[force])
            q, k, v, _, do = make_inputs(4, 12, M, S, 64, torch.bfloat16, None, seed=S + 7, strided=True)
            kw = {}
            table = (torch.randn(32, 12, generator=torch.Generator().manual_seed(1)) * 0.5).cuda()
            if mode == "rpe":
                kw = dict(rpe1d=pe.rpe1d_from_table(table), radius=128, rpe_bucket=pe.bucket_index32(128, True, 32, 128, "cuda"), num_buckets=32)
            plan = AttentionPlan(q, k, v, do, sm_scale=0.125, causal=causal, **kw)
            plan.forward(); plan.backward(); torch.cuda.synchronize()
            ref = [t.clone() for t in (plan.o, plan.lse, plan.dq, plan.dk, plan.dv)] + ([plan.dbias.clone()] if plan.dbias is not None else [])
            n_bad = 0
            for i in range(reps2):
                plan.forward(); plan.backward()
                if i % 25 == 24 or i == reps2 - 1:
                    torch.cuda.synchronize()
                    cur = [plan.o, plan.lse, plan.dq, plan.dk, plan.dv] + ([plan.dbias] if plan.dbias is not None else [])
                    n_bad += sum(0 if torch.equal(a, b) else 1 for a, b in zip(ref, cur))
            print(f"64-wide[{force}] M={M} N={S} {mode:5s} causal={int(causal)} x{reps2}: {'OK' if n_bad == 0 else 'MISMATCH x%d' % n_bad}", flush=True)
            bad += n_bad
print("TOTAL MISMATCHES", bad)
# round 5: the dense 64-wide bodies in one launch (attn_bwd_dfused64_kernel: dK/dV and dQ + dBias workgroups side by side behind bwd_stat2_kernel), forced on ragged / multi-group
# shapes; the head_dim-128 pipelined forward (default dispatch) with its dense bias ring
from flasht5_amd import _lib
for B, M, S, D, force in ((6, 1000, 1096, 64, True), (16, 512, 512, 64, True), (4, 1536, 1536, 64, False), (4, 2048, 2048, 128, False), (16, 1024, 1024, 128, False)):
    for causal in (False, True):
        _lib.set_variant((_lib.V_QDB64_ON | _lib.V_KV64_ON | _lib.V_FUSED64_ON) if force else 0)
        q, k, v, _, do = make_inputs(B, 12, M, S, D, torch.bfloat16, None, seed=S + B, strided=True)
        bias = torch.randn(1, 12, M, S, generator=torch.Generator().manual_seed(3)).bfloat16().cuda()
        plan = AttentionPlan(q, k, v, do, sm_scale=1.3 if B == 16 else D ** -0.5, causal=causal, bias=bias)
        plan.forward(); plan.backward(); torch.cuda.synchronize()
        ref = [t.clone() for t in (plan.o, plan.lse, plan.dq, plan.dk, plan.dv, plan.dbias)]
        n_bad = 0
        for i in range(reps):
            plan.forward(); plan.backward()
            if i % 25 == 24 or i == reps - 1:
                torch.cuda.synchronize()
                n_bad += sum(0 if torch.equal(a, b) else 1 for a, b in zip(ref, (plan.o, plan.lse, plan.dq, plan.dk, plan.dv, plan.dbias)))
        print(f"dense[{'forced one-launch' if force else 'default'}] B={B} M={M} N={S} D={D} causal={int(causal)} {plan.describe()} x{reps}: {'OK' if n_bad == 0 else 'MISMATCH x%d' % n_bad}", flush=True)
        bad += n_bad
        del plan
_lib.set_variant(0)
# ... the head_dim-128 pipelined forward without / with the T5 table: one partial round (ring requests spread over the MFMA gaps), several rounds, ragged, causal; each behind
# its backward (cold caches: what exposed the three-slot race above)
for B, M, S in ((4, 1024, 1024), (4, 2048, 2048), (16, 1024, 1024), (8, 1000, 1100)):
    for mode in ("none", "rpe"):
        for causal in (False, True):
            q, k, v, _, do = make_inputs(B, 12, M, S, 128, torch.bfloat16, None, seed=S + B + 1, strided=True)
            kw = {}
            if mode == "rpe":
                table = (torch.randn(32, 12, generator=torch.Generator().manual_seed(1)) * 0.5).cuda()
                kw = dict(rpe1d=pe.rpe1d_from_table(table), radius=128, rpe_bucket=pe.bucket_index32(128, True, 32, 128, "cuda"), num_buckets=32)
            plan = AttentionPlan(q, k, v, do, sm_scale=128 ** -0.5, causal=causal, **kw)
            plan.forward(); plan.backward(); torch.cuda.synchronize()
            ref = [t.clone() for t in (plan.o, plan.lse, plan.dq, plan.dk, plan.dv)]
            n_bad = 0
            for i in range(reps):
                plan.forward(); plan.backward()
                if i % 25 == 24 or i == reps - 1:
                    torch.cuda.synchronize()
                    n_bad += sum(0 if torch.equal(a, b) else 1 for a, b in zip(ref, (plan.o, plan.lse, plan.dq, plan.dk, plan.dv)))
            print(f"d128 B={B} M={M} N={S} {mode:5s} causal={int(causal)} fwd={plan.describe()['fwd']} x{reps}: {'OK' if n_bad == 0 else 'MISMATCH x%d' % n_bad}", flush=True)
            bad += n_bad
            del plan
print("stress (round-5 block) done, mismatches:", bad)
