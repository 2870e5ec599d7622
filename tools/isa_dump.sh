#!/bin/bash
# device ISA (.s) of the attention translation units into a directory: `bash tools/isa_dump.sh /tmp/isa_x` -- used to prove that a
# source clean-up (dead ablation branches removed) leaves the generated code untouched: diff the two dumps (comments stripped)
out=$1; mkdir -p "$out"; cd "$(dirname "$0")/.."
for u in "attn_fwd_inst 64" "attn_bwd_inst 64" "attn_bwd64_inst 64" "attn_fwd_inst 128" "attn_bwd_inst 32"; do
  set -- $u
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++20 -fPIC -mllvm -amdgpu-mfma-vgpr-form=1 -Wno-unused-value -I include -DFAT5_INST_D=$2 \
    --cuda-device-only -S flasht5_amd/csrc/$1.hip -o "$out/$1_$2.s" 2>/dev/null &
done
wait
for f in "$out"/*.s; do grep -v "^\s*;" "$f" | sed 's/;.*$//' | grep -v "^\s*\.\(file\|ident\|loc\|section\.debug\)" | grep -v "__hip_cuid_" > "$f.clean"; done
ls -la "$out"/*.clean
