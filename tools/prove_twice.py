#!/usr/bin/env python3
"""Full-size unit proof N times on one context; every proof must equal the first.  usage: prove_twice.py [log_n] [reps] [use_torch]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import eth_tx_proof_amd as E
from eth_tx_proof_amd import air, lib as zl
log_n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
SEED = int(os.environ.get("SEED", "5"), 0)
use_torch = int(sys.argv[3]) if len(sys.argv) > 3 else 0
cfg = E.standard_fast_config()
prog = air.cpu_like_air(93, 19)
if use_torch:
    import torch
    torch.cuda.set_device(0)
ctx = zl.Context(0)
if use_torch:
    t = torch.empty((93, 1 << log_n), dtype=torch.int64, device="cuda:0")
    _, pub = zl.synth_trace_dev(ctx, 93, log_n, SEED, dptr=t.data_ptr()); d = t.data_ptr()
else:
    d, pub = zl.synth_trace_dev(ctx, 93, log_n, SEED)
mode = sys.argv[4] if len(sys.argv) > 4 else ""
first = None
out = np.empty(zl.proof_size_words(cfg, [(prog, log_n, d, pub)]), dtype=np.uint64) if "o" in mode else None
import threading
for r in range(reps):
    if "p" in mode and r == 1:
        ctx.profile(True); ctx.profile_reset()
    res = {}
    def work():
        res["p"] = zl.prove_tables(ctx, cfg, [(prog, log_n, d, pub)], out=out) if out is not None else zl.prove_tables(ctx, cfg, [(prog, log_n, d, pub)])
    if "t" in mode:
        th = threading.Thread(target=work); th.start(); th.join()
    else:
        work()
    p = res["p"] if res["p"] is not None else out
    if first is None: first = p.copy()
    print("mode %r proof %d: %s" % (mode, r, "same" if (p == first).all() else "DIFFERENT"), flush=True)
if "p" in mode: print(sorted(ctx.profile_report().items(), key=lambda kv: -kv[1][1])[:3])
