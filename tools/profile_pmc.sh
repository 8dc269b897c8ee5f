#!/bin/bash
# PMC passes for the pair kernel (run on the GPU box through gpurun; counters in their own
# runs, never combined with trace domains other than --kernel-trace).
# usage: [KERNEL=regex] [STEPS=n] tools/profile_pmc.sh <outdir> [extra bench args]
set -u
OUT=${1:-gpurun_out/pmc}
shift || true
mkdir -p "$OUT"
export TMPDIR=/tmp
KERNEL=${KERNEL:-k_pair_gather}
STEPS=${STEPS:-10}
BENCH="python3 bench.py --steps $STEPS --warmup 5 --no-cpu-baseline $*"
pass() {
  name=$1; shift
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc "$@" --kernel-include-regex "$KERNEL" \
     --output-format csv -d "$OUT/$name" -- $BENCH > "$OUT/$name.log" 2>&1
  echo "$name rc=$?" >> "$OUT/passes.log"
}
pass sq1 SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_INSTS_LDS
pass sq2 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_SALU
pass fetch FETCH_SIZE
pass write WRITE_SIZE
pass l2 TCC_HIT_sum TCC_MISS_sum
cat "$OUT/passes.log"
