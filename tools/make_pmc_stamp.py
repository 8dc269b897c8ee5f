#!/usr/bin/env python3
"""make_pmc_stamp.py <pmc dir of tools/profile_pmc.sh> <style> <out json> -- the per-launch HBM traffic and VALU
instruction count of the pair kernel, stamped with the commit and kernel they were measured on (bench.py prints them as
roofline.traffic / roofline.valu only for the workload named in the stamp)."""
import csv
import glob
import json
import os
import subprocess
import sys
from collections import defaultdict

root, style, out = sys.argv[1], sys.argv[2], sys.argv[3]
kernel_key = {"table_ucgld": "k_pair_gather<0", "table_ucg_bethe": "k_pair_gather<1", "table_ucg_bethe_density": "k_density_pass"}[style]
acc = defaultdict(lambda: [0.0, 0])
names = set()
for f in glob.glob(os.path.join(root, "*", "*", "*counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        if kernel_key not in r["Kernel_Name"] or ", true, true, true" in r["Kernel_Name"].split("(")[0][-60:] and False:
            continue
        # the energy / virial variant (EV = true: the third template argument) runs once at setup: not the timed kernel
        targs = r["Kernel_Name"].split("<", 1)[1].split(">")[0].split(",") if "<" in r["Kernel_Name"] else []
        if style != "table_ucg_bethe_density" and len(targs) > 2 and targs[2].strip() == "true":
            continue
        if style == "table_ucg_bethe_density":  # k_density_pass2<TS, EV, ..>, k_density_pass3<EV>
            if "k_density_pass2" in r["Kernel_Name"] and len(targs) > 1 and targs[1].strip() == "true":
                continue
            if "k_density_pass3" in r["Kernel_Name"] and targs and targs[0].strip() == "true":
                continue
        acc[r["Counter_Name"]][0] += float(r["Counter_Value"])
        acc[r["Counter_Name"]][1] += 1
        kn = r["Kernel_Name"]
        names.add(kn[:kn.find("(ucg::PairDev")] if "(ucg::PairDev" in kn else kn[:120])
mean = {k: v[0] / v[1] for k, v in acc.items() if v[1]}
passes = 3 if style == "table_ucg_bethe_density" else 1  # the density style's three kernels per evaluation
stamp = {
    "commit": subprocess.run(["git", "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip() or "unknown",
    "kernel": " + ".join(n.replace("void ", "").replace("ucg::(anonymous namespace)::", "") for n in sorted(names))[:200] if names else kernel_key,
    "workload": f"{style} spline 1024 sc 100",
    # FETCH_SIZE is in KB and reads half the bytes of a wide stream on gfx950 (MI355X_MICROARCH.md, HBM): x2
    "traffic_bytes_per_launch": passes * (2.0 * mean.get("FETCH_SIZE", 0.0) + mean.get("WRITE_SIZE", 0.0)) * 1024.0,
    "valu_insts_per_launch": passes * mean.get("SQ_INSTS_VALU", 0.0),
    "counters_per_kernel_launch": {k: mean[k] for k in sorted(mean)},
}
json.dump(stamp, open(out, "w"), indent=1)
print(json.dumps(stamp)[:600])
