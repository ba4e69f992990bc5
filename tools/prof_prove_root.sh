#!/bin/bash
# rocprofv3 kernel stats of `bench.py --workload prove_root`: total kernel time and the union of kernel intervals against the wall time
set -u
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof_pr
rm -rf $OUT && mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o pr -- python3 $R/bench.py --workload prove_root --steps 6 ${PR_ARGS:-} > $OUT/bench.json 2> $OUT/bench.err || { tail -3 $OUT/bench.err; exit 1; }
cd $R
python3 - <<'PY'
import csv, json, glob
d = json.load(open("gpurun_out/prof_pr/bench.json")); print("tx/s", d["value"], "ms_per_step", d["ms_per_step"])
f = glob.glob("gpurun_out/prof_pr/**/pr_kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("total kernel ms", tot / 1e6, "calls", sum(int(r["Calls"]) for r in rows))
for r in rows[:22]:
    print(r["Name"][:72], r["Calls"], round(float(r["TotalDurationNs"]) / 1e6, 1), r["Percentage"])
t = glob.glob("gpurun_out/prof_pr/**/pr_kernel_trace.csv", recursive=True)
if t:
    iv = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in csv.DictReader(open(t[0])))
    busy, (cs, ce) = 0, iv[0]
    for s, e in iv[1:]:
        if s > ce:
            busy += ce - cs; cs, ce = s, e
        else:
            ce = max(ce, e)
    busy += ce - cs
    print("span ms", (iv[-1][1] - iv[0][0]) / 1e6, "union busy ms", busy / 1e6)
PY
find $OUT -name "*kernel_trace.csv" -size +3M -delete
