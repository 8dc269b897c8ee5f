#!/bin/bash
# Same-box comparison of library variants (tools/build_variant.sh): every variant's bench twice, interleaved.
# usage: tools/ab_libs.sh <outdir> "<bench args>" name1 name2 ...   (name "main" = the in-tree library)
set -u
OUT=$1; ARGS=$2; shift 2
mkdir -p "$OUT"
for rep in 1 2; do
  for n in "$@"; do
    if [ "$n" = main ]; then unset UCG_HIP_LIBRARY; else export UCG_HIP_LIBRARY=$PWD/_ab/libs/libucg_$n.so; fi
    timeout -k 10 240 python3 bench.py --no-cpu-baseline --no-nve-leg --dropin-steps 0 $ARGS > "$OUT/${n}_$rep.json" 2> "$OUT/${n}_$rep.err" || { echo "$n run $rep failed"; tail -3 "$OUT/${n}_$rep.err"; exit 1; }
    python3 - "$OUT/${n}_$rep.json" "$n" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r=d["roofline"]
print(f"{sys.argv[2]:>12s}  {d['value']:9.1f} steps/s  {d['ms_per_step']*1000:7.1f} us/step  kernel {r['avg_launch_us']:7.1f} us  frac {r['frac']:.4f}")
PY
  done
done
