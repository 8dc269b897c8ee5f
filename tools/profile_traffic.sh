#!/bin/bash
# HBM traffic (FETCH_SIZE / WRITE_SIZE, separate --pmc passes with --kernel-trace only) of the kernels matching
# KERNEL, for one bench.py configuration.  usage: KERNEL=regex tools/profile_traffic.sh <outdir> [bench args]
set -u
OUT=${1:-gpurun_out/traffic}
shift || true
mkdir -p "$OUT"
export TMPDIR=/tmp
KERNEL=${KERNEL:-k_pair_gather}
BENCH="python3 bench.py --steps ${STEPS:-10} --warmup 5 --no-cpu-baseline $*"
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $c --kernel-include-regex "$KERNEL" --output-format csv \
      -d "$OUT/$c" -- $BENCH > "$OUT/$c.log" 2>&1
  echo "$c rc=$?" >> "$OUT/passes.log"
done
cat "$OUT/passes.log"
python3 tools/pmc_summary.py "$OUT"
