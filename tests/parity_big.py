#!/usr/bin/env python3
"""One-off parity run beyond what the test suite does: the unit proof at 2^LOG_N rows x 93 (+40) columns (default 23, north_star's
target shape) on the GPU against the oracle's proof of the same trace, word for word.  The oracle needs about 5 KB of host memory per
row and minutes of CPU time, so it runs outside the default suite (tests/test_gpu_big.py, ZK_TEST_BIG=1): it refuses to start without
enough free memory and prints a heartbeat while the oracle runs.  usage: python tests/parity_big.py [log_n] > result.json"""
import json, os, sys, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))      # tests/ -> repo root
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
import eth_tx_proof_amd as E
from eth_tx_proof_amd import air, lib as zl
import orc

log_n = int(sys.argv[1]) if len(sys.argv) > 1 else 23
need_gb = 6.0 * (1 << log_n) / (1 << 20)          # measured: ~5 GB per 2^20 rows, with headroom
avail_gb = next(int(l.split()[1]) for l in open("/proc/meminfo") if l.startswith("MemAvailable")) / (1 << 20)
if avail_gb < need_gb * 1.5:
    print(json.dumps({"log_n": log_n, "skipped": "needs %.0f GB of host memory, %.0f available" % (need_gb, avail_gb)}))
    sys.exit(0)
cfg = E.standard_fast_config()
prog = air.cpu_like_air(93, 19)
seed = 0x5EED0000 + log_n
ctx = zl.Context(0)
t0 = time.time()
d, pub = zl.synth_trace_dev(ctx, 93, log_n, seed)
proof = zl.prove_tables(ctx, cfg, [(prog, log_n, d, pub)])
t_gpu = time.time() - t0
ctx.dev_free(d)
stop = False


def heartbeat():
    t = time.time()
    while not stop:
        time.sleep(20)
        print("[parity_big] oracle running, %.0f s" % (time.time() - t), file=sys.stderr, flush=True)
threading.Thread(target=heartbeat, daemon=True).start()
orc.lib().orc_set_threads(min(64, os.cpu_count() or 1))
t0 = time.time()
tr, opub = orc.synth_trace(93, log_n, seed)
rc, ref = orc.prove(cfg, [(prog, log_n, tr, opub)])
t_cpu = time.time() - t0
stop = True
same = bool(rc == 0 and proof.size == ref.size and (proof == ref).all())
print(json.dumps({"log_n": log_n, "cols": 93, "aux_cols": 40, "proof_words": int(proof.size), "gpu_seconds_incl_trace_generation": round(t_gpu, 3),
                  "oracle_seconds_incl_trace_generation": round(t_cpu, 1), "oracle_threads": min(64, os.cpu_count() or 1),
                  "identical_word_for_word": same, "host_mem_available_gb": round(avail_gb)}))
sys.exit(0 if same else 1)
