#!/bin/bash
# Collects the rocprofv3 evidence for bench.py on the GPU box (run from the repo root through gpurun):
#   tools/profile_bench.sh <tag>      -> gpurun_out/prof_<tag>/{trace,FETCH_SIZE,WRITE_SIZE,SQ_INSTS_VALU,SQ_INSTS_LDS}
# One pass per counter set, kernel trace only (no other trace domain next to --pmc).
set -e
TAG=${1:-r01f}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 "$ROOT/bench.py" --no-cpu-baseline > "$OUT/trace_bench.json"
echo "trace done"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$OUT/FETCH_SIZE" -- python3 "$ROOT/bench.py" --steps 3 --warmup 1 --no-cpu-baseline > /dev/null
echo "fetch done"
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$OUT/WRITE_SIZE" -- python3 "$ROOT/bench.py" --steps 3 --warmup 1 --no-cpu-baseline > /dev/null
echo "write done"
rocprofv3 --pmc SQ_INSTS_VALU GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d "$OUT/SQ_INSTS_VALU" -- python3 "$ROOT/bench.py" --steps 3 --warmup 1 --no-cpu-baseline > /dev/null
echo "valu done"
rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace --output-format csv -d "$OUT/SQ_INSTS_LDS" -- python3 "$ROOT/bench.py" --steps 3 --warmup 1 --no-cpu-baseline > /dev/null
echo "lds done"
