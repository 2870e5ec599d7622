#!/bin/bash
# one gpurun call: probe layouts, correctness sweep, timings
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
{
hipcc --offload-arch=gfx950 -O2 tools/probe_layout.hip -o /tmp/probe_layout 2>/dev/null && timeout 60 /tmp/probe_layout
echo "=== devcheck ==="
timeout 900 python tools/devcheck.py "$@"
} > gpurun_out/first_gpu.log 2>&1
tail -c 6000 gpurun_out/first_gpu.log
