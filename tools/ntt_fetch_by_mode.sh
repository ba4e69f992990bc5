#!/bin/bash
# HBM bytes fetched / written per launch of the four W = 10 passes (tools/ntt_floor.hip) as shipped (mode 0), with loads only (3) and with
# stores only (4): does the write stream push the half-used input lines of the TILE passes out of L2 (FETCH_SIZE above the data)?
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for m in 0 3 4; do for C in FETCH_SIZE WRITE_SIZE; do
  rm -rf gpurun_out/fb_$m_$C
  rocprofv3 --kernel-trace --output-format csv --pmc $C -d gpurun_out/fb_${m}_$C -o c -- tools/bin/ntt_floor$m 93 4 > /dev/null 2>&1
  python3 - $m $C <<PY
import csv,glob,collections,sys
m,C=sys.argv[1],sys.argv[2]
f=glob.glob("gpurun_out/fb_%s_%s/**/*counter_collection.csv" % (m,C),recursive=True)[0]
acc=collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    if r["Counter_Name"]==C and "ntt_fast" in r["Kernel_Name"]: acc[r["Kernel_Name"][16:52]].append(float(r["Counter_Value"]))
data=93*2**20*8/1e6
for k,v in acc.items():
    kb=sum(v)/len(v); mb=(2*kb if C=="FETCH_SIZE" else kb)*1024/1e6     # gfx950: FETCH_SIZE counts half the bytes of wide coalesced reads (guide)
    print("mode %s %-10s %-38s %8.0f MB per launch (data %.0f MB): %.2f x" % (m, C, k, mb, data, mb/data))
PY
  find gpurun_out/fb_${m}_$C -name "*.csv" -size +1M -delete
done; done
