import sys, os, cProfile, pstats, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
import eth_tx_proof_amd as E
from eth_tx_proof_amd import air, lib as zl, ops, plonk
import bench
rec = plonk.SyntheticRecursion()
st = ops.set_prover_state_from_config(0, recursion=rec, chain_overlap=False)
zl.prover_state_cache_dir(bench.PROVER_STATE_DIR)
ctx0 = st.ctx()
tabs = []
for t, (w, lg) in enumerate(zip(air.ZKEVM_TABLE_WIDTHS, air.ZKEVM_CFG4_LOG_SIZES)):
    d, pub = zl.synth_trace_dev(ctx0, w, lg, 0x5EED0000 + 1000 * t)
    tabs.append(ops.TableInput(air.cpu_like_air(w, min(19, w // 4)), lg, d, pub))
u0 = ops.GenerationInputs(tabs)
st.prepare(u0.tables)
st.chain_streams = 1
ops.prove_root(u0); ops.prove_root(u0)
pr = cProfile.Profile()
t0 = time.perf_counter()
pr.enable()
for _ in range(5):
    ops.prove_root(u0)
pr.disable()
print("5 calls: %.1f ms each" % ((time.perf_counter() - t0) / 5 * 1e3))
ps = pstats.Stats(pr); ps.sort_stats("tottime"); ps.print_stats(18)
