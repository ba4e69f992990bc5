#!/bin/bash
# A/B of the Plonk gate kernels' piece length: the shipped library against tools/bin/libzkgl_plk320.so (PLK_SEG 320 / slack 80: 21 kernels of
# 68-186 registers, one of 348, instead of 9 of 68-474).  Single proofs (plonk_quotient per proof) and whole transactions, alternating.
for L in "" "$PWD/tools/bin/libzkgl_plk320.so" "" "$PWD/tools/bin/libzkgl_plk320.so"; do
  echo "== library: ${L:-shipped}"
  for k in 12 15; do ZK_LIB_PATH=$L NO_ORACLE=1 python tools/plonk_bench.py $k 5 2>/dev/null | grep -E "profiler off|plonk_quotient|plonk_gates" | tr '\n' ' '; echo; done
  ZK_LIB_PATH=$L ZK_WITNESS_STANDIN=0 python bench.py --workload prove_root --inflight 8 --steps 4 --warmup 1 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('  prove_root', round(d['value'],2),'tx/s', d['timed_proof_check']['identical_to_oracle'], round(d['one_tx_alone_ms']['total'],1),'ms alone')"
done
