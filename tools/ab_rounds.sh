#!/bin/bash
# Same-box A/B of an older tree (extracted + built under _ab/<name>) against the working tree: the two benches are run
# alternately (A B A B) so that a box's clock drift shows up as spread inside one build, not as a difference between them.
# usage: tools/ab_rounds.sh <outdir> <_ab/name> [bench args common to both trees]
set -u
OUT=${1:-gpurun_out/ab}
OLD=${2:-_ab/r02}
shift 2 || true
mkdir -p "$OUT"
ROOT=$(pwd)
for rep in 1 2; do
  (cd "$OLD" && timeout -k 10 240 python3 bench.py --no-cpu-baseline --no-nve-leg "$@" > "$ROOT/$OUT/old_$rep.json" 2> "$ROOT/$OUT/old_$rep.err") || { echo "old run $rep failed"; exit 1; }
  echo "old $rep: $(tail -1 $OUT/old_$rep.json | cut -c1-200)"
  timeout -k 10 240 python3 bench.py --no-cpu-baseline --no-nve-leg --dropin-steps 0 "$@" > "$OUT/new_$rep.json" 2> "$OUT/new_$rep.err" || { echo "new run $rep failed"; exit 1; }
  echo "new $rep: $(tail -1 $OUT/new_$rep.json | cut -c1-200)"
done
