#!/bin/bash
# sample sclk / power while the forward kernel runs in a loop (developer tool)
cd "$GRAFT_REPO_ROOT" || exit 1
python - <<'PY' &
import os, sys, time
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import torch
from attn_helpers import make_inputs
from flasht5_amd.flash_attention_v2_bias import AttentionPlan
q, k, v, _, do = make_inputs(4, 12, 8192, 8192, 64, torch.bfloat16, None, seed=1, strided=True)
plan = AttentionPlan(q, k, v, do, sm_scale=0.125)
plan.forward(); torch.cuda.synchronize()
t0 = time.time()
while time.time() - t0 < 12:
    for _ in range(50): plan.forward()
    torch.cuda.synchronize()
PY
pid=$!
sleep 6
for i in 1 2 3; do
  rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power|fclk|mclk" | head -8
  sleep 1.5
done
wait $pid
echo "--- idle"
rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power" | head -4
rocm-smi --showmaxpower --showclkfrq 2>/dev/null | grep -iE "max|sclk" | head -20
