#!/bin/bash
# the bench AIR's constraint kernel whole (ZK_QJIT_SEG_MIN above its 844 instructions) against sequences of call-free
# kernels of at most N instructions (hiprtc builds them in seconds), each inside bench.py's four-stream region: proofs/s, the kernel's
# time beside the other streams and alone, and the digest check.  Result of round 5: profiles/r05j_quotient_split.txt; 300 became the
# default for programs of 640..1535 instructions (prover_quotient_jit.hip seg_len).
run() { echo "== $1"; shift; env "$@" python bench.py --steps 12 --warmup 3 --no-prove-root --no-log23 --no-cpu-baseline --no-pcie --no-live-pmc 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernels_profiled_region']; a=d['kernels_one_unit_alone']
print(round(d['value'],2),'proofs/s  quotient_eval region avg_ms',k.get('quotient_eval'),' alone',a.get('quotient_eval'), d['timed_proof_check']['identical_to_oracle'])"; }
run "whole kernel" ZK_QUOTIENT_AOT=0 ZK_QUOTIENT_JIT=2 ZK_QJIT_SEG_MIN=100000
for s in ${@:-250 300 350 448}; do run "pieces of <= $s" ZK_QUOTIENT_AOT=0 ZK_QUOTIENT_JIT=2 ZK_QJIT_SEG_MIN=800 ZK_QJIT_SEG=$s; done
run "as shipped" X=1
