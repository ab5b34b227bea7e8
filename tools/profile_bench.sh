#!/bin/bash
# Collects the rocprofv3 evidence for bench.py on the GPU box (run from the repo root through gpurun):
#   tools/profile_bench.sh <tag> [bench args]   -> gpurun_out/prof_<tag>/{trace,FETCH_SIZE,WRITE_SIZE,SQ_A,SQ_B,SQ_C}
# One pass per counter set, kernel trace only (no other trace domain next to --pmc).  The bench runs with
# --no-secondary --no-cpu-baseline so that every stream_kernel launch in the trace is a launch of the headline
# workload (the chr22-size secondary loop would otherwise be averaged into the same kernel name).
set -e
TAG=${1:-r02}
shift || true
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
ARGS="--no-secondary --no-cpu-baseline $*"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 "$ROOT/bench.py" $ARGS > "$OUT/trace_bench.json"
echo "trace done"
for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES GRBM_GUI_ACTIVE" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES"; do
  name=$(echo $set | cut -d' ' -f1)
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d "$OUT/$name" -- python3 "$ROOT/bench.py" --steps 2 --warmup 1 $ARGS > /dev/null 2>"$OUT/$name.err" || echo "FAILED $set"
  echo "$name done"
done
