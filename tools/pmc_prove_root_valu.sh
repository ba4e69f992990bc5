#!/bin/bash
# vector instructions of ONE transaction, per kernel (rocprofv3 --pmc SQ_INSTS_VALU over bench.py --workload prove_root --inflight 1): the
# instruction-stream time of a transaction = sum / the chip's issue rate, against the 42 ms per transaction at saturation
set -u
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/pmc6
rm -rf $OUT && mkdir -p $OUT
cd $R
rocprofv3 --kernel-trace --output-format csv --pmc SQ_INSTS_VALU -d $OUT/valu -o c -- python3 bench.py --workload prove_root --inflight 1 --steps 2 --warmup 1 > $OUT/bench.json 2> $OUT/bench.err || { tail -3 $OUT/bench.err; exit 1; }
python3 - <<'PY' > $OUT/prove_root_valu_per_kernel.txt
import csv, glob, json, os, collections
root = os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/pmc6"
f = glob.glob(root + "/valu/**/c_counter_collection.csv", recursive=True)[0]
acc = collections.OrderedDict()
n_root = 0
for row in csv.DictReader(open(f)):
    k = row["Kernel_Name"].split("(")[0].replace("void ", "")[:70]
    a = acc.setdefault(k, [0, 0.0]); a[0] += 1; a[1] += float(row["Counter_Value"])
# the run proves the same transaction 1 (setup) + 3 (split) + 1 (profiled) + 1 (warm-up) + 2 (timed) = 8 times, plus circuit setup commits
N_TX = 8.0
tot = sum(t for _, t in acc.values())
print("SQ_INSTS_VALU (wave-instructions) per transaction, by kernel; %d transactions in the run" % N_TX)
for k, (n, t) in sorted(acc.items(), key=lambda kv: -kv[1][1])[:30]:
    print("%-72s %8.0f launches/tx %10.1f M instr/tx  %5.1f %%" % (k, n / N_TX, t / N_TX / 1e6, 100 * t / tot))
print("total %.1f M wave-instructions per transaction" % (tot / N_TX / 1e6))
print("at 1024 SIMDs x 2.1 GHz / 4 cycles = 537 G wave-instr/s: %.1f ms; at the 515 G/s of tools/ubench_int: %.1f ms" % (tot / N_TX / 537e9 * 1e3, tot / N_TX / 515e9 * 1e3))
PY
cat $OUT/prove_root_valu_per_kernel.txt | head -40
rm -rf $OUT/valu
