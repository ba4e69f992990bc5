import os, sys
os.environ["ZK_QUOTIENT_JIT"]="2"; os.environ["ZK_QUOTIENT_JIT_DUMP"]="gpurun_out/quotient_jit_93.hip"
sys.path.insert(0, ".")
import numpy as np
import eth_tx_proof_amd as E
from eth_tx_proof_amd import lib as zk, air
ctx = zk.Context(0)
cfg = E.standard_fast_config()
prog = air.cpu_like_air(93, 19)
d, pub = zk.synth_trace_dev(ctx, 93, 12, 7)
proof = zk.prove_tables(ctx, cfg, [(prog, 12, d, pub)])
print("ok", proof.size)
