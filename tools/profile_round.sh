#!/bin/bash
# The measurements of one round in ONE gpurun call: bench lines of every configuration, the rocprofv3 kernel statistics of the
# headline command and the PMC passes the roofline.traffic / roofline.valu figures come from.  Everything lands under
# gpurun_out/<tag>/; copy what is to be judged into profiles/ (tools/collect_round.py does).
# usage: tools/profile_round.sh r03
set -u
TAG=${1:-r03}
OUT=gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
run() { name=$1; shift; timeout -k 10 600 "$@" > "$OUT/$name.json" 2> "$OUT/$name.err"; echo "$name rc=$?" | tee -a "$OUT/steps.log"; }
run bench_default        python3 bench.py
run bench_driver_style   python3 bench.py --gpus 1 --steps 20 --warmup 5
run config2              python3 bench.py --config 2 --steps 2000 --no-cpu-baseline --no-nve-leg
run config3              python3 bench.py --config 3 --steps 1000 --no-cpu-baseline
run density_1M           python3 bench.py --style table_ucg_bethe_density --steps 500 --no-cpu-baseline
run config5              python3 bench.py --config 5 --steps 200 --warmup 50 --no-cpu-baseline
UCG_FORCE_MULTI=1 MASTER_PORT=29577 run decomposed_1rank_125k python3 bench.py --ncell 50 --steps 2000 --no-cpu-baseline
# kernel statistics of the headline command
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/kstats" -- python3 bench.py --steps 500 --warmup 100 --no-cpu-baseline --no-nve-leg --dropin-steps 0 > "$OUT/kstats.json" 2> "$OUT/kstats.err"
echo "kstats rc=$?" | tee -a "$OUT/steps.log"
UCG_FORCE_MULTI=1 MASTER_PORT=29578 timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/kstats_decomposed" -- python3 bench.py --ncell 50 --steps 400 --warmup 100 --no-cpu-baseline > "$OUT/kstats_decomposed.json" 2> "$OUT/kstats_decomposed.err"
echo "kstats_decomposed rc=$?" | tee -a "$OUT/steps.log"
# counters (separate --pmc passes, --kernel-trace only)
KERNEL=k_pair_gather STEPS=10 tools/profile_pmc.sh "$OUT/pmc_ucgld" --no-nve-leg --dropin-steps 0 > /dev/null
KERNEL=k_pair_gather STEPS=10 tools/profile_pmc.sh "$OUT/pmc_bethe" --config 3 --dropin-steps 0 > /dev/null
KERNEL=k_density STEPS=10 tools/profile_pmc.sh "$OUT/pmc_density" --style table_ucg_bethe_density --dropin-steps 0 > /dev/null
cat "$OUT"/pmc_*/passes.log | tee -a "$OUT/steps.log"
