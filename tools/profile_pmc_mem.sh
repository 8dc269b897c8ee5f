#!/bin/bash
# vector-memory-path counters (TA / TCP / TD) for the pair kernel; same conventions as profile_pmc.sh
set -u
OUT=${1:-gpurun_out/pmcmem}
shift || true
mkdir -p "$OUT"
export TMPDIR=/tmp
BENCH="python3 bench.py --steps 10 --warmup 5 --no-cpu-baseline $*"
pass() {
  name=$1; shift
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc "$@" --kernel-include-regex 'k_pair_gather' \
     --output-format csv -d "$OUT/$name" -- $BENCH > "$OUT/$name.log" 2>&1
  echo "$name rc=$?" >> "$OUT/passes.log"
}
# TA / TD: at most TWO counters of the block per pass.  Round 1 asked for four TA_* (three TD_*) in one pass;
# rocprofiler_create_counter_config then fails with "error code 38: Request exceeds the capabilities of the hardware to
# collect" and the process aborts (SIGABRT under the first kernel launch) -- an over-subscribed counter block, not a hang.
pass ta1 TA_TA_BUSY_sum TA_FLAT_READ_WAVEFRONTS_sum
pass ta2 TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum
pass tcp1 TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum
pass tcp2 TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_GATE_EN1_sum TCP_GATE_EN2_sum
pass td1 TD_TD_BUSY_sum TD_TC_STALL_sum
pass td2 TD_LOAD_WAVEFRONT_sum
pass grbm GRBM_GUI_ACTIVE GRBM_COUNT
cat "$OUT/passes.log"
