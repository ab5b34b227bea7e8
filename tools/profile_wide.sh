#!/bin/bash
# Counters of one case of tools/wide_time.py (run through gpurun from the repo root):
#   tools/profile_wide.sh <tag> <case number> [sq]  -> gpurun_out/prof_wide_<tag>/{FETCH_SIZE,WRITE_SIZE[,SQ_INSTS_VALU,SQ_LDS_BANK_CONFLICT]}
# One pass per counter set, kernel trace only (no other trace domain next to --pmc).
TAG=$1; CASE=$2; SQ=$3
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_wide_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
SETS=("FETCH_SIZE" "WRITE_SIZE")
if [ "$SQ" = "sq" ]; then
  SETS+=("SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES GRBM_GUI_ACTIVE" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES")
fi
for set in "${SETS[@]}"; do
  name=$(echo $set | cut -d' ' -f1)
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d "$OUT/$name" -- python3 "$ROOT/tools/wide_time.py" --only $CASE --reps 2 > "$OUT/$name.out" 2>"$OUT/$name.err" || echo "FAILED $set"
  echo "$name done"
done
