#!/bin/bash
# The round's closing profile: `python3 bench.py` (the default command, exactly) under rocprofv3 --kernel-trace --stats; the kernel-stats CSV and
# the bench line it produced go to gpurun_out/prof_final/ (copied to profiles/ by hand).
set -u
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof_final
rm -rf $OUT && mkdir -p $OUT
cd $R
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o c -- python3 bench.py > $OUT/bench_default_under_rocprofv3.json 2> $OUT/bench.err || { tail -5 $OUT/bench.err; exit 1; }
cp $(find $OUT/trace -name "*kernel_stats.csv" | head -1) $OUT/rocprofv3_kernel_stats_default.csv
find $OUT/trace -name "*.csv" -size +2M -delete 2>/dev/null
head -5 $OUT/rocprofv3_kernel_stats_default.csv | cut -c1-160
python3 -c "
import json;d=json.load(open('$OUT/bench_default_under_rocprofv3.json'));print(d['value'],d['roofline']['avg_launch_ms'],d['roofline']['avg_launch_ms_all_leaf_hash_shapes'])"
