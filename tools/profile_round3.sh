#!/bin/bash
# Round-3 measurements on the GPU box (run from the repo root through gpurun); summaries are copied to profiles/ afterwards.
set -u
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof3
rm -rf $OUT && mkdir -p $OUT
cd $R
python3 bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err || { echo "default bench failed"; tail -5 $OUT/bench_default.err; exit 1; }
echo "default bench done"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o bench -- python3 bench.py --no-cpu-baseline --no-pcie --no-live-pmc > $OUT/bench_under_rocprof.json 2> $OUT/bench_under_rocprof.err || { echo "rocprof bench failed"; exit 1; }
echo "rocprof stats done"
python3 tools/plonk_bench.py 13 5 > $OUT/plonk_bench_2p13.txt 2>&1 || { echo "plonk bench failed"; tail -5 $OUT/plonk_bench_2p13.txt; exit 1; }
NO_ORACLE=1 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_plonk -o plonk -- python3 tools/plonk_bench.py 13 20 > $OUT/plonk_under_rocprof.txt 2>&1 || { echo "rocprof plonk failed"; exit 1; }
python3 tools/plonk_bench.py 12 5 > $OUT/plonk_bench_2p12.txt 2>&1 || { echo "plonk bench 2^12 failed"; exit 1; }
python3 tools/cfg4_bench.py > $OUT/cfg4_bench.txt 2>&1 || { echo "cfg4 bench failed"; tail -5 $OUT/cfg4_bench.txt; exit 1; }
[ -x tools/bin/ubench_lat ] && tools/bin/ubench_lat > $OUT/ubench_lat_gfx950.txt 2>&1
echo "plonk done"
python3 bench.py --workload prove_root --steps 4 > $OUT/bench_prove_root.json 2> $OUT/bench_prove_root.err || { echo "prove_root bench failed"; tail -5 $OUT/bench_prove_root.err; exit 1; }
echo "prove_root done"
ZK_BENCH_BACKEND=gloo python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29701 bench.py --gpus 2 --steps 8 --warmup 1 --inflight 2 --no-pcie --no-cpu-baseline --no-profile > $OUT/bench_2ranks_gloo.json 2> $OUT/bench_2ranks_gloo.err || { echo "2-rank rehearsal failed"; tail -5 $OUT/bench_2ranks_gloo.err; exit 1; }
echo "2-rank rehearsal done"
find $OUT -name "*kernel_trace.csv" -size +3M -delete 2>/dev/null
du -sh $OUT
