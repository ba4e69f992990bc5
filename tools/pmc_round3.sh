#!/bin/bash
# Round-3 counter passes (each --pmc run alone, as the pool requires):
#  A. the TILE-pass FETCH question (VERDICT r2 weak #3b): the same 93-column commit at 2^18 rows (two passes of W = 9: TILE runs of 16 elements
#     = 128 B) and at 2^20 rows (W = 10: TILE runs of 8 elements = 64 B), FETCH_SIZE / WRITE_SIZE per NTT kernel
#  B. one whole 2^20-row unit proof: FETCH_SIZE, WRITE_SIZE, SQ_INSTS_VALU, SQ_WAVE_CYCLES + SQ_BUSY_CYCLES for quotient_eval and the rest
set -u
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/pmc3
rm -rf $OUT && mkdir -p $OUT
cd $R
for LG in 18 20; do
  for C in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --kernel-trace --output-format csv --pmc $C -d $OUT/commit${LG}_$C -o c -- python3 tools/commit_bench.py $LG 93 1 > $OUT/commit${LG}_$C.log 2>&1 || echo "pmc pass commit $LG $C failed" >> $OUT/errors.txt
  done
done
for C in FETCH_SIZE WRITE_SIZE SQ_INSTS_VALU "SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY"; do
    tag=$(echo $C | tr ' ' '+')
    rocprofv3 --kernel-trace --output-format csv --pmc $C -d $OUT/proof_$tag -o c -- python3 tools/prove_twice.py 20 1 > $OUT/proof_$tag.log 2>&1 || echo "pmc pass proof $tag failed" >> $OUT/errors.txt
done
python3 - <<'PY' > $OUT/pmc_round3_summary.json
import csv, glob, json, os
root = os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/pmc3"
out = {}
for d in sorted(glob.glob(root + "/*/")):
    f = glob.glob(d + "**/c_counter_collection.csv", recursive=True)
    if not f:
        continue
    acc = {}
    for row in csv.DictReader(open(f[0])):
        k = (row["Counter_Name"], row["Kernel_Name"].split("(")[0])
        a = acc.setdefault(k, [0, 0.0, 0.0])
        v = float(row["Counter_Value"])
        a[0] += 1; a[1] += v; a[2] = max(a[2], v)
    tag = os.path.basename(d.rstrip("/"))
    for (c, kern), (n, tot, mx) in acc.items():
        out.setdefault(tag, {}).setdefault(c, {})[kern] = {"launches": n, "mean_per_launch": tot / n, "max_per_launch": mx}
json.dump(out, open("/dev/stdout", "w"), indent=1, sort_keys=True)
PY
find $OUT -name "*kernel_trace.csv" -size +3M -delete 2>/dev/null
find $OUT -name "*.csv" -size +8M -delete 2>/dev/null
du -sh $OUT; cat $OUT/errors.txt 2>/dev/null
