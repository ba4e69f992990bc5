#!/usr/bin/env python3
"""Summarise the rocprofv3 --pmc passes of tools/profile_round.sh: mean counter value per launch for every kernel.
usage: python tools/pmc_summary.py gpurun_out/prof > profiles/<round>_pmc_summary.json"""
import csv, glob, json, os, sys
root = sys.argv[1]
out = {}
for d in sorted(glob.glob(os.path.join(root, "pmc_*"))):
    f = os.path.join(d, "c_counter_collection.csv")
    if not os.path.isfile(f):
        continue
    acc = {}
    for row in csv.DictReader(open(f)):
        k = (row["Counter_Name"], row["Kernel_Name"].split("(")[0])
        a = acc.setdefault(k, [0, 0.0])
        a[0] += 1; a[1] += float(row["Counter_Value"])
    for (c, kern), (n, tot) in acc.items():
        out.setdefault(c, {})[kern] = {"launches": n, "mean_per_launch": tot / n}
json.dump(out, sys.stdout, indent=1, sort_keys=True)
