#!/bin/bash
# L1 (TCP) counters of the pair kernel only -- the TA_/TD_ counter passes hang on this pool, do not add them
set -u
OUT=${1:-gpurun_out/tcp}
shift || true
mkdir -p "$OUT"
export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum \
   --kernel-include-regex 'k_pair_gather' --output-format csv -d "$OUT/tcp1" -- python3 bench.py --steps 10 --warmup 5 --no-cpu-baseline "$@" > "$OUT/tcp1.log" 2>&1
echo "rc=$?"
