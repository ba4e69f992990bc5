#!/bin/bash
# Round-4 counter passes: where the cycles of BASELINE's unit go.  Each --pmc run alone, program directly after `--`.
#  A. bench.py's own regions (one unit alone, then four units in flight) under counters.  rocprofv3 serialises dispatches while it collects
#     counters, so these are PER-KERNEL facts (instructions, busy and stalled cycles, the clock the chip actually ran at), not concurrency facts.
#  B. the same command under --kernel-trace alone: the timeline of the real four-stream region (overlap of kernels = sum of durations / wall time).
#  C. the four W = 10 NTT passes of tools/ntt_floor.hip (mode 0, as shipped) under SQ_INSTS_VALU: vector instructions per element and pass.
set -u
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/pmc4
rm -rf $OUT && mkdir -p $OUT
cd $R
BENCH="python3 bench.py --no-cpu-baseline --no-pcie --no-live-pmc --no-prove-root --no-log23 --steps 4"
for C in "GRBM_GUI_ACTIVE" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU" "SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_ACTIVE_INST_ANY SQ_INSTS_SALU"; do
    tag=$(echo $C | tr ' ' '+')
    echo "[pmc4] bench under $C" >&2
    rocprofv3 --kernel-trace --output-format csv --pmc $C -d $OUT/bench_$tag -o c -- $BENCH > $OUT/bench_$tag.log 2>&1 || echo "pmc pass bench $tag failed" >> $OUT/errors.txt
done
echo "[pmc4] bench under --kernel-trace --stats" >&2
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/bench_trace -o c -- $BENCH > $OUT/bench_trace.log 2>&1 || echo "trace pass failed" >> $OUT/errors.txt
echo "[pmc4] ntt_floor0 under SQ_INSTS_VALU" >&2
rocprofv3 --kernel-trace --output-format csv --pmc SQ_INSTS_VALU -d $OUT/ntt_SQ_INSTS_VALU -o c -- tools/bin/ntt_floor0 93 4 > $OUT/ntt_SQ_INSTS_VALU.log 2>&1 || echo "pmc pass ntt failed" >> $OUT/errors.txt
python3 tools/pmc_round4_summary.py $OUT > $OUT/pmc_round4_summary.json 2> $OUT/summary.err
cp $(find $OUT/bench_trace -name "*kernel_stats.csv" | head -1) $OUT/rocprofv3_kernel_stats_default.csv 2>/dev/null
find $OUT -name "*kernel_trace.csv" -size +6M -delete 2>/dev/null
find $OUT -name "*.csv" -size +10M -delete 2>/dev/null
du -sh $OUT; cat $OUT/errors.txt 2>/dev/null; head -c 1500 $OUT/summary.err
