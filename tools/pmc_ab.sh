#!/bin/bash
# A/B of two variants of the pair kernel under ONE set of SQ counters (same box, same pass): the regular kernel with two
# lanes per bead against the pair_once kernel.  rocprofv3 starts the program itself; the environment is exported here.
export TMPDIR=/tmp
mkdir -p gpurun_out/r2_o
CNT="SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"
export UCG_GATHER_SLOTS=2 UCG_PAIR_ONCE=0
timeout -k 10 300 rocprofv3 --kernel-trace --pmc $CNT --kernel-include-regex k_pair_gather --output-format csv -d gpurun_out/r2_o/slots2 -- python3 bench.py --steps 10 --warmup 5 --no-cpu-baseline --no-nve-leg > gpurun_out/r2_o/slots2.log 2>&1
export UCG_GATHER_SLOTS=0 UCG_PAIR_ONCE=1
timeout -k 10 300 rocprofv3 --kernel-trace --pmc $CNT --kernel-include-regex k_pair_gather --output-format csv -d gpurun_out/r2_o/once -- python3 bench.py --steps 10 --warmup 5 --no-cpu-baseline --no-nve-leg > gpurun_out/r2_o/once.log 2>&1
python3 tools/pmc_summary.py gpurun_out/r2_o | grep -v "true, t"
