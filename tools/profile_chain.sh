#!/bin/bash
# rocprofv3 counters of the chain kernel alone (tools/chain_time.py): gpurun_out/prof_chain_<tag>/
TAG=${1:-r03}
shift || true
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_chain_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 "$ROOT/tools/chain_time.py" $* > "$OUT/trace.json" || echo "FAILED trace"
for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES GRBM_GUI_ACTIVE" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES" \
           "SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM"; do
  name=$(echo $set | cut -d' ' -f1)
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d "$OUT/$name" -- python3 "$ROOT/tools/chain_time.py" --reps 2 $* > /dev/null 2>"$OUT/$name.err" || echo "FAILED $set"
  echo "$name done"
done
