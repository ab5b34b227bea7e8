#!/bin/bash
# counters of a cluster-engine launch shape (tools/group_ab.py on a 400 Mb record): tools/profile_group.sh <tag> K W1,W2,.. [ENV=V] -> gpurun_out/prof_grp_<tag>; summary: python tools/summarize_group_pmc.py gpurun_out/prof_grp_<tag>
TAG=$1; K=$2; WS=$3; SET=$4
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_grp_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for kv in $SET; do export $kv; done
for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES GRBM_GUI_ACTIVE" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES"; do
  name=$(echo $set | cut -d' ' -f1)
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d "$OUT/$name" -- python3 "$ROOT/tools/group_ab.py" $K $WS > "$OUT/$name.out" 2>"$OUT/$name.err" || echo "FAILED $set"
  echo "$name done"
done
