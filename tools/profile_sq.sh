#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_pmc2
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for set in "SQ_INSTS_SALU SQ_INSTS_VALU SQ_WAVE_CYCLES" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS" "SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAIT_ANY" "SQ_INST_CYCLES_VMEM SQ_INSTS_SMEM SQ_INSTS_VMEM" "SQ_BUSY_CYCLES SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU"; do
  name=$(echo $set | tr ' ' '_')
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d "$OUT/$name" -- python3 "$ROOT/bench.py" --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2>"$OUT/$name.err" || echo "FAILED $set"
  echo "$set done"
done
