#!/bin/bash
# rocprofv3 kernel trace of one Plonk proof size: ordered kernel list with the idle gap before each kernel (where the launch latency goes)
# usage: bash tools/plonk_gaps.sh [degree_bits=12]
set -u
K=${1:-12}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/plonk_gaps_$K
rm -rf $OUT && mkdir -p $OUT
NO_ORACLE=1 rocprofv3 --kernel-trace --output-format csv -d $OUT -o pg -- python3 $R/tools/plonk_bench.py $K 3 > $OUT/bench.txt 2> $OUT/bench.err || { tail -3 $OUT/bench.err; exit 1; }
cd $R
python3 - $OUT <<'PY'
import csv, glob, sys
out = sys.argv[1]
f = glob.glob(out + "/**/pg_kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
# last proof = the kernels after the last but one "plonk_hash_no_pad" ... simpler: take the last third by count of a marker kernel
marks = [i for i, r in enumerate(rows) if r["Kernel_Name"].startswith("plonk_perm_rows")]
lo = marks[-1]
# a proof starts a few kernels before perm_rows: walk back to the previous proof's last kernel (largest gap)
start = lo
while start > 0 and int(rows[start]["Start_Timestamp"]) - int(rows[start - 1]["End_Timestamp"]) < 200000 and lo - start < 60:
    start -= 1
seg = rows[start:]
tot_k = tot_g = 0
lines = []
for a, b in zip([None] + seg[:-1], seg):
    d = (int(b["End_Timestamp"]) - int(b["Start_Timestamp"])) / 1e3
    g = (int(b["Start_Timestamp"]) - int(a["End_Timestamp"])) / 1e3 if a else 0.0
    tot_k += d; tot_g += max(g, 0)
    lines.append("%8.1f %8.1f  %s" % (g, d, b["Kernel_Name"][:70]))
open(out + "/gaps.txt", "w").write("gap_us  dur_us  kernel\n" + "\n".join(lines) + "\nkernels %d  kernel time %.1f us  gaps %.1f us\n" % (len(seg), tot_k, tot_g))
print("kernels", len(seg), "kernel us", round(tot_k, 1), "gap us", round(tot_g, 1))
PY
find $OUT -name "*kernel_trace.csv" -size +3M -delete
