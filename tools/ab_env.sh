#!/bin/bash
# Same-box comparison of one library under different environment settings (bench.py's UCG_* option hooks): each setting twice,
# interleaved.  usage: tools/ab_env.sh <outdir> "<bench args>" "NAME=VAL ..." "NAME=VAL ..." ...   ("-" = no extra setting)
set -u
OUT=$1; ARGS=$2; shift 2
mkdir -p "$OUT"
for rep in 1 2; do
  i=0
  for setting in "$@"; do
    i=$((i+1))
    if [ "$setting" = "-" ]; then envs=""; else envs="$setting"; fi
    env $envs timeout -k 10 300 python3 bench.py --no-cpu-baseline --no-nve-leg --dropin-steps 0 $ARGS > "$OUT/s${i}_$rep.json" 2> "$OUT/s${i}_$rep.err" || { echo "setting $i run $rep failed"; tail -3 "$OUT/s${i}_$rep.err"; exit 1; }
    python3 - "$OUT/s${i}_$rep.json" "$setting" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r=d["roofline"]
print(f"{sys.argv[2]:>24s}  {d['value']:9.1f} steps/s  {d['ms_per_step']*1000:7.1f} us/step  kernel {r['avg_launch_us']:7.1f} us  frac {r['frac']:.4f}")
PY
  done
done
