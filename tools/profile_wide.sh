#!/bin/bash
# HBM traffic counters of one case of tools/wide_time.py (run through gpurun from the repo root):
#   tools/profile_wide.sh <tag> <case number>  -> gpurun_out/prof_wide_<tag>/{FETCH_SIZE,WRITE_SIZE}
TAG=$1; CASE=$2
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_wide_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for set in "FETCH_SIZE" "WRITE_SIZE"; do
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d "$OUT/$set" -- python3 "$ROOT/tools/wide_time.py" --only $CASE --reps 2 > "$OUT/$set.out" 2>"$OUT/$set.err" || echo "FAILED $set"
  echo "$set done"
done
