#!/bin/bash
# VERDICT r4 #4a follow-up: transaction throughput by challenger placement, hardware-queue count and transactions in flight (witness stand-in off).
# Round 5: A 25.1, B 28.2, C 26.3, D 27.5, E 23.1, F 17.0 tx/s (profiles/r05i_prove_root_variants.txt): no setting beats the default by more than a region scatters.
run() { echo "== $1"; shift; env "$@" ZK_WITNESS_STANDIN=0 python bench.py --workload prove_root --steps 4 --warmup 1 --inflight $INF 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value'],2),'tx/s', d['timed_proof_check']['identical_to_oracle'], round(d['one_tx_alone_ms']['total'],1),'ms alone')"; }
INF=8 run "A default 8" X=1
INF=8 run "B host 8" ZK_TRANSCRIPT_HOST=1
INF=12 run "C host q16 12" ZK_TRANSCRIPT_HOST=1 GPU_MAX_HW_QUEUES=16
INF=12 run "D dev q16 12" GPU_MAX_HW_QUEUES=16
INF=12 run "E host 12" ZK_TRANSCRIPT_HOST=1
INF=16 run "F host q16 16" ZK_TRANSCRIPT_HOST=1 GPU_MAX_HW_QUEUES=16
