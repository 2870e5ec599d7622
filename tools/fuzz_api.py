"""developer fuzz of the PUBLIC operators at the library's own dispatch (no variant bits): `flash_attention_v2_bias(q, k, v, bias, causal, sm_scale)` -- the reference's
operator, every bias broadcast form it accepts ((1|B, 1|H, M, N), reference flash_attention_v2_bias.py:45-52) or none -- and `flash_attention_v2_rpe` (T5 table in-kernel),
through autograd, on random problems: head_dim 16 / 32 / 64 / 128, 1 .. 700 rows and keys (any remainder, M != N, single rows), batch 1 .. 5, 1 .. 4 heads, bf16 / fp16,
(B,S,H,D)-strided or contiguous operands, causal or not, several scales, and -- dense mode -- masks the way the model builds them (finfo.min or -inf added into the bias for
padded keys: modeling_flash_t5.py:267-277).  o, dq, dk, dv, dbias / dtable against the fp32 oracle with the bounds of tests/test_attention_gpu.py; a tensor past the bound
must still satisfy the reference's own rule (at most twice the error of eager attention in the input dtype, tests/fa2_triton/test_fa2_bias.py:64-67).
usage: [FUZZ_LARGE=1 | FUZZ_VARLEN=1 | FUZZ_ROWWISE=1] python tools/fuzz_api.py [n_cases] [seed]      (FUZZ_LARGE=1: production-sized problems, see tests/api_fuzz.py)"""
import os, sys, random
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from api_fuzz import run_case, run_varlen_case, run_rowwise_case

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
bad = 0
for i in range(n):
    if os.environ.get("FUZZ_VARLEN", "0") == "1":  # packed batches (flash_attn_varlen_func)
        desc, msgs = run_varlen_case(i, rng)
    elif os.environ.get("FUZZ_ROWWISE", "0") == "1":  # fast_rms_layernorm / cross_entropy_loss
        desc, msgs = run_rowwise_case(i, rng)
    else:
        desc, msgs = run_case(i, rng, large=os.environ.get("FUZZ_LARGE", "0") == "1")
    bad += bool(msgs)
    print(f"{'OK ' if not msgs else 'BAD'} case {i}: {desc} {'; '.join(msgs)}", flush=True)
print(f"FUZZ_API: {bad} bad of {n}")
