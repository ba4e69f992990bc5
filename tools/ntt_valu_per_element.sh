#!/bin/bash
# SQ_INSTS_VALU per element and pass of the four W = 10 passes (tools/ntt_floor.hip, mode 0), for the shipped kernel and for the same source
# built with -DNTT_SH64=0 (every inter-phase twiddle a general product: round 4's arithmetic in round 5's kernel body).
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for v in ntt_floor0 ntt_floor0_old; do
  rm -rf gpurun_out/sq_$v
  rocprofv3 --kernel-trace --output-format csv --pmc SQ_INSTS_VALU -d gpurun_out/sq_$v -o c -- tools/bin/$v 93 4 > gpurun_out/sq_$v.log 2>&1
  echo "== $v"
  python3 - $v <<PY
import csv,glob,collections,sys
f=glob.glob("gpurun_out/sq_%s/**/*counter_collection.csv" % sys.argv[1],recursive=True)[0]
acc=collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    if r["Counter_Name"]=="SQ_INSTS_VALU" and "ntt_fast" in r["Kernel_Name"]: acc[r["Kernel_Name"][:64]].append(float(r["Counter_Value"]))
for k,v in acc.items(): print("%-66s %d launches, %.1f VALU instructions per element" % (k, len(v), sum(v)/len(v)*64/(93*2**20)))
PY
  find gpurun_out/sq_$v -name "*.csv" -size +1M -delete
done
