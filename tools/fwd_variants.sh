#!/bin/bash
# forward S = 8192 ablation table (run ON the GPU box): every FAT5_FWD_ABL variant library present (tools/build_variant.py fablN
# attn_fwd64_d64.o "-DFAT5_FWD_ABL=N"), timed by graph replay and counted by PMC.  Output: gpurun_out/fwd_variants/.
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
OUT="$GRAFT_REPO_ROOT/gpurun_out/fwd_variants"; mkdir -p "$OUT"
C1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU"
for v in "" ${VARIANTS:-fabl1 fabl2 fabl4 fabl8 fabl5}; do
  [ -n "$v" ] && [ ! -f flasht5_amd/lib/libfat5_$v.so ] && continue
  export FAT5_LIB_VARIANT=$v
  n=${v:-product}
  python tools/attn_time.py --S 8192 --modes none,rpe --what fwd --iters 20 --reps 5 > "$OUT/time_$n.txt" 2>&1
  bash tools/pmc.sh "--S 8192 --mode rpe --what fwd --iters 3 --seconds 0.3" "$C1" > "$OUT/pmc_$n.txt" 2>&1
  echo "== $n"; cat "$OUT/time_$n.txt" | tail -3; grep -E "MFMA_BUSY|GUI_ACTIVE|WAVE_CYCLES|duration|INSTS_VALU|ACTIVE_INST_VALU" "$OUT/pmc_$n.txt"
done
