#!/bin/bash
# Profiles of one round on the GPU box (run from the repo root through gpurun); summaries are copied to profiles/ afterwards.
#   1. rocprofv3 kernel-trace stats of the default bench.py command
#   2. PMC passes (each alone, as the pool requires) over one 93-column 2^20-row commit: HBM bytes, instruction mix
set -u
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof
rm -rf $OUT && mkdir -p $OUT
cd $R
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o bench -- python3 bench.py --no-cpu-baseline --no-live-pmc > $OUT/bench_under_rocprof.json 2> $OUT/bench_under_rocprof.err
for C in FETCH_SIZE WRITE_SIZE "SQ_INSTS_VALU SQ_INSTS_SALU" "SQ_INSTS_VALU_MFMA_I8 SQ_VALU_MFMA_BUSY_CYCLES" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY"; do
    tag=$(echo $C | tr ' ' '+')
    rocprofv3 --kernel-trace --output-format csv --pmc $C -d $OUT/pmc_$tag -o c -- python3 tools/commit_bench.py 20 93 1 > $OUT/pmc_$tag.log 2>&1 || echo "pmc pass $tag failed" >> $OUT/errors.txt
done
# what bench.py reads for roofline.traffic / roofline.alu, and the per-kernel counter means: copy these two into profiles/ as rNN_*
python3 tools/pmc_leaf_hash.py $OUT > $OUT/pmc_leaf_hash.json || echo "pmc_leaf_hash failed" >> $OUT/errors.txt
python3 tools/pmc_summary.py $OUT > $OUT/pmc_summary_commit_bench.json
# the integer-ALU micro-benchmark the issue bound is taken from
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 tools/ubench_int.hip -o $OUT/ubench_int 2> $OUT/ubench_build.err && $OUT/ubench_int > $OUT/ubench_int_gfx950.txt 2>&1; rm -f $OUT/ubench_int
# keep only what is needed (the per-dispatch CSVs are small; drop traces)
find $OUT -name "*kernel_trace.csv" -size +3M -delete 2>/dev/null
du -sh $OUT
ls -R $OUT | head -60
