#!/bin/bash
# quick iteration: correctness (quick) + timing, optional env passthrough:  bash tools/quick_gpu.sh "ENV=1 ..." [devcheck args]
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
envs="$1"; shift
env $envs timeout 900 python tools/devcheck.py "$@" 2>&1 | grep -v amdgpu.ids | tee gpurun_out/quick.log | tail -70
