#!/bin/bash
# effective shader clock under the pair kernel: GRBM_GUI_ACTIVE (summed over the 8 XCDs) / 8 / kernel time
# (MI355X_MICROARCH.md, DVFS give-back).  usage: tools/profile_clock.sh <outdir> [bench args]
set -u
OUT=${1:-gpurun_out/clock}
shift || true
mkdir -p "$OUT"
export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE --kernel-include-regex "${KERNEL:-k_pair_gather}" \
   --output-format csv -d "$OUT/grbm" -- python3 bench.py --steps ${STEPS:-20} --warmup 5 --no-cpu-baseline --no-nve-leg "$@" > "$OUT/grbm.log" 2>&1
echo "rc=$?"
python3 - "$OUT" <<'PY'
import csv, glob, sys, os
root = sys.argv[1]
cc = glob.glob(os.path.join(root, "grbm", "*", "*counter_collection.csv"))[0]
kt = glob.glob(os.path.join(root, "grbm", "*", "*kernel_trace.csv"))[0]
dur = {}
for r in csv.DictReader(open(kt)):
    dur[r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"]), r["Kernel_Name"])
rows = []
for r in csv.DictReader(open(cc)):
    if r["Counter_Name"] != "GRBM_GUI_ACTIVE":
        continue
    d = dur.get(r["Dispatch_Id"])
    if d and d[0] > 100000:
        rows.append((float(r["Counter_Value"]) / 8.0 / d[0], d[0], d[1][:70]))
rows.sort(key=lambda t: -t[1])
for ghz, ns, name in rows[:8]:
    print(f"{ghz:.3f} GHz  {ns / 1e3:9.1f} us  {name}")
PY
