#!/bin/bash
# rocprofv3 evidence for configs 3 and 4 (GRCh38-size, tools/run_config34.py); run from the repo root through gpurun:
#   tools/profile_config34.sh <tag>  -> gpurun_out/prof_<tag>/{trace,FETCH_SIZE}
set -e
TAG=${1:-r01e_c34}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 "$ROOT/tools/run_config34.py" --reps 5 --out "$OUT/config34.json" > /dev/null
echo "trace done"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$OUT/FETCH_SIZE" -- python3 "$ROOT/tools/run_config34.py" --reps 2 > /dev/null
echo "fetch done"
