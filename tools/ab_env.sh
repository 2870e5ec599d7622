#!/bin/bash
# bash tools/ab_env.sh "<python script + args>" "ENV1=a ENV2=b" "ENV1=c" ...   (each config = one env string; use _ for none)
cd "$GRAFT_REPO_ROOT" || exit 1
cmd="$1"; shift
for e in "$@"; do
  echo "######## env: $e"
  if [ "$e" == "_" ]; then timeout 600 python $cmd 2>&1 | grep -v amdgpu.ids; else env $e timeout 600 python $cmd 2>&1 | grep -v amdgpu.ids; fi
done
