#!/bin/bash
# A/B over library variants: bash tools/ab.sh "script args" var1 var2 ...   ("" = product lib)
cd "$GRAFT_REPO_ROOT" || exit 1
cmd="$1"; shift
for v in "$@"; do
  echo "######## variant: ${v:-<default>}"
  FAT5_LIB_VARIANT="$v" timeout 600 python $cmd 2>&1 | grep -v amdgpu.ids
done
