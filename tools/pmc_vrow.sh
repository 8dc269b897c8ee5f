#!/bin/bash
# A/B of the plain gather kernel and the virtual-row kernel (option pair_vrow) under the same SQ counter sets, one box.
# usage: tools/pmc_vrow.sh <outdir> [extra bench args]
export TMPDIR=/tmp
OUT=${1:-gpurun_out/pmc_vrow}
shift || true
mkdir -p "$OUT"
C1="SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"
C2="SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_BUSY_CYCLES SQ_INSTS_SMEM"
for v in 0 1; do
  export UCG_PAIR_VROW=$v
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $C1 --kernel-include-regex "k_pair_" --output-format csv -d "$OUT/c1_vrow$v" -- python3 bench.py --steps 10 --warmup 5 --no-cpu-baseline --no-nve-leg --dropin-steps 0 "$@" > "$OUT/c1_vrow$v.log" 2>&1
  echo "c1 vrow$v rc=$?"
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $C2 --kernel-include-regex "k_pair_" --output-format csv -d "$OUT/c2_vrow$v" -- python3 bench.py --steps 10 --warmup 5 --no-cpu-baseline --no-nve-leg --dropin-steps 0 "$@" > "$OUT/c2_vrow$v.log" 2>&1
  echo "c2 vrow$v rc=$?"
done
python3 tools/pmc_summary.py "$OUT" > "$OUT/summary.txt"
