run() { echo "== $1"; shift; env "$@" python bench.py --steps 12 --warmup 3 --no-prove-root --no-log23 --no-cpu-baseline --no-pcie --no-live-pmc 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernels_profiled_region']; a=d['kernels_one_unit_alone']
print(round(d['value'],2),'proofs/s  quotient_eval region avg_ms',k.get('quotient_eval'),' alone',a.get('quotient_eval'), d['timed_proof_check']['identical_to_oracle'])"; }
run "whole kernel (AOT)" X=1
run "3 kernels of <= 300 instructions (hiprtc)" ZK_QUOTIENT_AOT=0 ZK_QJIT_SEG_MIN=800 ZK_QJIT_SEG=300
run "2-3 kernels of <= 448" ZK_QUOTIENT_AOT=0 ZK_QJIT_SEG_MIN=800 ZK_QJIT_SEG=448
run "whole kernel (AOT) again" X=1
