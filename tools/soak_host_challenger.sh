#!/bin/bash
# VERDICT r4 #4a: the configuration that once produced nothing for six minutes (round 4, before the SCC fix) -- the Fiat-Shamir sponge forced
# onto the host for EVERY transaction, 16 hardware queues, 12 transactions in flight -- run N times in fresh processes.  A run that is not
# done after 150 s gets SIGUSR1 (bench.py dumps every Python thread's stack), then SIGKILL, and the loop STOPS there (no retry after a hang).
# usage: tools/soak_host_challenger.sh [runs=20] [out=gpurun_out/soak_host_challenger.txt]
N=${1:-20}; OUT=${2:-gpurun_out/soak_host_challenger.txt}
: > "$OUT"
for i in $(seq 1 "$N"); do
  S=$(date +%s.%N)
  ZK_BENCH_DUMP_ON_USR1=1 ZK_TRANSCRIPT_HOST=1 GPU_MAX_HW_QUEUES=16 ZK_WITNESS_STANDIN=0 timeout -s USR1 -k 20 150 \
    python bench.py --workload prove_root --inflight 12 --steps 2 --warmup 1 > /tmp/soak_line.json 2> /tmp/soak_err.txt
  rc=$?
  E=$(date +%s.%N)
  v=$(python -c "import json; d=json.loads(open('/tmp/soak_line.json').read().strip().splitlines()[-1]); print('%.2f tx/s identical_to_oracle=%s' % (d['value'], d['timed_proof_check']['identical_to_oracle']))" 2>/dev/null)
  printf "run %2d: rc=%d  %s s  %s\n" "$i" "$rc" "$(python3 -c "print(round($E - $S, 1))")" "$v" | tee -a "$OUT"
  if [ "$rc" -ne 0 ]; then echo "STOPPED at run $i (rc $rc); stderr tail:" | tee -a "$OUT"; tail -60 /tmp/soak_err.txt | tee -a "$OUT"; exit 1; fi
done
echo "all $N runs finished" | tee -a "$OUT"
